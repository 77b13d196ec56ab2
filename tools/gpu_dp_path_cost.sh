#!/bin/bash
# the data-parallel code path over an RCCL group of ONE rank against the plain single-process step, interleaved on one box (what the path costs before a byte crosses a link)
cd $GRAFT_REPO_ROOT
for i in 1 2; do for m in plain group; do
  if [ $m = group ]; then X="--rccl-group-of-one"; else X=""; fi
  timeout 300 python bench.py $X --no-b32 --no-extras --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$m [$i]', round(d['value'], 1), 'img/s ', round(d['ms_per_step'], 3), 'ms/step  host enqueue', round(d['host_enqueue_ms_per_step'], 2), 'ms  hip_graphs', d['hip_graphs'])"
done; done
