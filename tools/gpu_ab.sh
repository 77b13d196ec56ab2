#!/bin/bash
# A/B the bench under environment / flag variants (no per-kernel survey).  usage: tools/gpu_ab.sh <tag> "ENV=.. -- --flags" ...
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
i=0
for cfg in "$@"; do
  i=$((i+1))
  envs="${cfg%%--*}"; flags="${cfg#*--}"; [ "$flags" = "$cfg" ] && flags=""
  env $envs timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing $flags 2>&1 | tail -1 > $O/bench_$i.json
  python -c "
import json; d=json.load(open('$O/bench_$i.json')); print('[$cfg]', 'img/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],2), 'graphs', d['hip_graphs'])"
done
