#!/bin/bash
# round-2 session H: non-default options / conditional parity tests + regression of the kernel and network tests
tag=${1:-r2h}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_flags.py -q -m gpu -s > $O/pytest_flags.log 2>&1; echo "flags rc=$?"; grep -E "passed|failed|Error|assert|FAILED" $O/pytest_flags.log | tail -30
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_networks.py tests/test_cabi.py -q -m gpu -x > $O/pytest_reg.log 2>&1; echo "regression rc=$?"; tail -6 $O/pytest_reg.log
