#!/bin/bash
# bench under several environment settings (eager, no graphs, per-layer table each).  usage: tools/gpu_env_sweep.sh <tag> "VAR=a VAR2=b" "VAR=c" ...
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
i=0
for cfg in "$@"; do
  i=$((i+1))
  echo "== [$i] $cfg"
  env $cfg timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --layer-table $O/layers_$i.txt 2>&1 | tail -1 > $O/bench_$i.json
  python -c "
import json; d=json.load(open('$O/bench_$i.json')); print('img/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],2), 'lib ms', d['roofline']['library_kernels_ms_per_step'])"
  python tools/layer_summary.py $O/layers_$i.txt 40 | grep -E "wgrad" | head -8
done
