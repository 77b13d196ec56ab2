#!/bin/bash
# rocprofv3 kernel trace of the default bench (no library profiler events) + gap analysis.   usage: tools/gpu_trace.sh <tag>
tag=${1:-tX}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $tag -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/bench.log 2>&1
grep '"metric"' $O/bench.log | cut -c1-260
python $R/tools/trace_gaps.py $O/prof/${tag}_kernel_trace.csv 0.5
