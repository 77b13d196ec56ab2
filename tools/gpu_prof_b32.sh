#!/bin/bash
# rocprofv3 --kernel-trace --stats of the single-stream batch-32 step (the `b32` block's roofline leg) + a kernel-level sanity run.
tag=${1:-prof_b32}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -1 $O/pytest_kernels.log
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o st -- python $R/bench.py --batch-per-gpu 32 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --graphs off --streams 00 > $O/prof_bench.log 2>&1
grep '^{"metric"' $O/prof_bench.log | tail -1 | cut -c1-200
rm -f $O/prof/*kernel_trace.csv $O/prof/*agent_info.csv
head -6 $O/prof/st_kernel_stats.csv | cut -c1-160
