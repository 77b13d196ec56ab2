#!/bin/bash
# round 5, session r: split replay with the update as eager launches vs as a second graph: the data-parallel graph test both ways, then the
# path's cost on one GPU (RCCL group of one rank), interleaved
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5r; mkdir -p $O; cd $R
BOX="$(hostname) gpu-uid $(cat /sys/class/drm/card*/device/unique_id 2>/dev/null | head -1)"; echo "# box: $BOX   commit: $(cat tools/.evidence_commit 2>/dev/null)" | tee $O/dp_eager_update.txt
timeout 200 python -m pytest tests/test_gpu_graphs.py -q -m gpu -k "data_parallel or failed_capture or replay_matches" > $O/pytest.log 2>&1; echo "pytest rc=$? $(grep -aE 'passed|failed' $O/pytest.log | tail -1)" | tee -a $O/dp_eager_update.txt
for i in 1 2; do
  for mode in plain eager graph; do
    X="--rccl-group-of-one"; [ $mode = plain ] && X=""
    E=1; [ $mode = graph ] && E=0
    SGX_DP_EAGER_UPDATE=$E timeout 200 python bench.py $X --no-b32 --no-extras --no-cpu-baseline --no-kernel-timing --steps 30 2>$O/$mode$i.err | tail -1 > $O/$mode$i.json
    python - $O/$mode$i.json $mode $i <<'P' | tee -a $O/dp_eager_update.txt
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read())
    print(f"{sys.argv[2]:5s} [{sys.argv[3]}] {j['value']:.1f} img/s  {j['ms_per_step']:.3f} ms/step  host enqueue {j['host_enqueue_ms_per_step']:.2f} ms  hip_graphs {j['hip_graphs']}  aux/side {j.get('aux_stream')}/{j.get('side_stream')}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "no line:", e)
P
  done
done
grep -a "Error\|error" $O/pytest.log | head -5 | cut -c1-300
