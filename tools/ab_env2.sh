#!/bin/bash
# Interleaved A/B of two ENVIRONMENTS on the default bench line (headline batch 4 + the batch-32 block) on one box.
#   usage (GPU box): tools/ab_env2.sh "<env assignments of arm OFF>" [rounds]     e.g. tools/ab_env2.sh "SGX_GEPI_APPLY1=0 SGX_GRID_ALL_CAP=8192" 2
off=$1; rounds=${2:-2}
for i in $(seq $rounds); do
  for v in off on; do
    if [ $v = off ]; then e="$off"; else e="SGX_AB_DUMMY=1"; fi
    env $e timeout 400 python bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('[$v] b4', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 3), 'ms graphs', d.get('hip_graphs'), 'launches', d.get('library_launches_per_step'), '| b32', round(d['b32']['value'], 1), 'img/s', round(d['b32']['ms_per_step'], 2), 'ms')"
  done
done
