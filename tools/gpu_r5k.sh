#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5k; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_graphs.py -q -m gpu -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|launches \(host" $O/pytest.log | tail -12
