#!/bin/bash
# round 5, session h: batch-4 A/B of the statistics-from-the-producer policies (launch count vs epilogue cost)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5h; mkdir -p $O; cd $R
run() { env "$@" timeout 300 python bench.py --no-b32 --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$*: b4', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 3), 'ms graphs', d.get('hip_graphs'), 'launches', d['roofline']['library_launches_per_step'], 'kernel ms', d['roofline']['library_kernels_ms_per_step'])"; }
for i in 1 2; do
  run X=0
  run SGX_FUSE_EPI_STATS_MIN=0
  run SGX_FUSE_EPI_STATS=3 SGX_FUSE_EPI_STATS_MIN=0
done 2>&1 | tee $O/ab_stats_b4.txt
