#!/bin/bash
tag=${1:-tX}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o $tag -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --graphs off > $O/bench.log 2>&1
grep '"metric"' $O/bench.log | cut -c1-200
