#!/bin/bash
# round 5, session l: the default bench line with its two extra blocks (configs[4] top depths replayed, configs[1] in a child)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5l; mkdir -p $O; cd $R
BOX="$(cat /sys/class/kfd/kfd/topology/nodes/*/gpu_id 2>/dev/null | tr '\n' ' ')"
t0=$(date +%s)
timeout 600 python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json; echo "rc=$? wall $(( $(date +%s) - t0 )) s"
python - <<'P'
import json,os
j=json.loads(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r5l/bench_default.json").read())
print("b4", j["value"], j["ms_per_step"], "b32", j["b32"]["value"], j["b32"]["ms_per_step"])
print("sweep", json.dumps(j.get("sweep_top_depths"))[:900])
print("f128", json.dumps(j.get("ffhq128_fp32_b64"))[:600])
print("cpu", j.get("cpu_baseline"))
P
tail -5 $O/bench_default.err
