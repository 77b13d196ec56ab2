#!/bin/bash
tag=${1:-r2f}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k conv2 > $O/pytest_conv2.log 2>&1; echo "conv2 tests rc=$?"; tail -4 $O/pytest_conv2.log; grep "^FAILED" $O/pytest_conv2.log | head -20
timeout 600 python tools/conv2_probe.py --reps 10 --geo D U > $O/probe.log 2>&1; echo "probe rc=$?"; grep -v amdgpu.ids $O/probe.log
