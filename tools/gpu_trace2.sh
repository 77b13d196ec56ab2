#!/bin/bash
# kernel traces of the default bench in both launch modes + gap / exclusive-time analysis.   usage: tools/gpu_trace2.sh <tag>
tag=${1:-tX}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in off on; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$mode -o t -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --graphs $mode > $O/bench_$mode.log 2>&1
  grep '"metric"' $O/bench_$mode.log | cut -c1-200
  python $R/tools/trace_gaps.py $O/prof_$mode/t_kernel_trace.csv 0.4 > $O/gaps_$mode.txt 2>&1
  rm -rf $O/prof_$mode
done
