#!/bin/bash
# A/B of one environment switch on the default bench line (headline batch 4 + the batch-32 block), interleaved, on one box.
#   usage (GPU box): tools/ab_bench.sh VAR OFFVALUE [rounds]        e.g. tools/ab_bench.sh SGX_BLUR_SHFL 0 2
var=$1; off=$2; rounds=${3:-2}
for i in $(seq $rounds); do
  for v in off on; do
    if [ $v = off ]; then export $var=$off; else unset $var; fi
    timeout 400 python bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$var $v: b4', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 3), 'ms graphs', d.get('hip_graphs'), '| b32', round(d['b32']['value'], 1), 'img/s', round(d['b32']['ms_per_step'], 2), 'ms')"
  done
done
