#!/bin/bash
# run one pytest selection on the GPU box.   usage: tools/gpu_one.sh <tag> <pytest args...>
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 1200 python -m pytest "$@" -q -m gpu -s > $O/pytest.log 2>&1; echo "rc=$?"; grep -E "^\[|passed|failed|Error|FAILED|assert" $O/pytest.log | tail -80
