#!/bin/bash
# usage: tools/gpu_one.sh <tag> <pytest args...>   -- one pytest invocation on the GPU box, log merged back
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 2000 python -m pytest "$@" > $O/pytest.log 2>&1; echo "rc=$?"; grep -E "^\[|passed|failed" $O/pytest.log | tail -40
