#!/bin/bash
# round-2 session B: real-config parity, train driver, capture-failure fallback; then the whole suite
tag=${1:-r2b}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_realconfigs.py tests/test_gpu_train.py -q -m gpu -x -s --durations=12 > $O/pytest_new.log 2>&1; echo "new tests rc=$?"; grep -E "^\[real|passed|failed|Error|error|assert" $O/pytest_new.log | tail -40; tail -25 $O/pytest_new.log
timeout 600 python -m pytest tests/test_gpu_graphs.py -q -m gpu -x > $O/pytest_graphs.log 2>&1; echo "graphs rc=$?"; tail -8 $O/pytest_graphs.log
