#!/bin/bash
# round-2 session I: two ranks on one GPU (gloo), flags re-run, graph/DP regression
tag=${1:-r2i}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_dp2.py -q -m gpu -s > $O/pytest_dp2.log 2>&1; echo "dp2 rc=$?"; grep -E "passed|failed|Error|assert|FAILED" $O/pytest_dp2.log | tail -30
timeout 600 python -m pytest tests/test_gpu_flags.py -q -m gpu > $O/pytest_flags.log 2>&1; echo "flags rc=$?"; tail -3 $O/pytest_flags.log
timeout 900 python -m pytest tests/test_gpu_graphs.py tests/test_gpu_train.py -q -m gpu -x > $O/pytest_graphs.log 2>&1; echo "graphs rc=$?"; tail -8 $O/pytest_graphs.log
