#!/bin/bash
# stream-structure A/B at batch 32 (and 4): side stream for weight gradients / auxiliary stream for the fake branch
tag=${1:-r2m}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
for B in 32 4; do
for cfg in "1 1" "0 1" "1 0" "0 0"; do
  set -- $cfg
  st=3; [ $B = 4 ] && st=10
  r=$(SGX_AUX_STREAM=$1 SGX_PARAM_STREAM=$2 timeout 300 python bench.py --batch-per-gpu $B --steps $st --warmup 2 --no-cpu-baseline --no-kernel-timing --graphs off 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f ms/step %.1f img/s host %.1f'%(d['ms_per_step'],d['value'],d['host_enqueue_ms_per_step']))")
  echo "B=$B aux=$1 param=$2 eager: $r"
done
done
