#!/bin/bash
# round 5, session q: what the data-parallel code path costs per step on ONE GPU (RCCL group of size 1) against the plain step, same box, interleaved
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5q; mkdir -p $O; cd $R
BOX="$(hostname) gpu-uid $(cat /sys/class/drm/card*/device/unique_id 2>/dev/null | head -1)"; echo "# box: $BOX   commit: $(cat tools/.evidence_commit 2>/dev/null)" | tee $O/dp_path_cost.txt
for i in 1 2; do
  for mode in plain group; do
    X=""; [ $mode = group ] && X="--rccl-group-of-one"
    timeout 200 python bench.py $X --no-b32 --no-extras --no-cpu-baseline --no-kernel-timing --steps 30 2>$O/$mode$i.err | tail -1 > $O/$mode$i.json
    python - $O/$mode$i.json $mode $i <<'P' | tee -a $O/dp_path_cost.txt
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read())
    print(f"{sys.argv[2]:5s} [{sys.argv[3]}] {j['value']:.1f} img/s  {j['ms_per_step']:.3f} ms/step  host enqueue {j['host_enqueue_ms_per_step']:.2f} ms  hip_graphs {j['hip_graphs']}  aux/side {j.get('aux_stream')}/{j.get('side_stream')}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "no line:", e)
P
  done
done
tail -2 $O/group1.err | cut -c1-300
