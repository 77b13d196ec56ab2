#!/bin/bash
tag=${1:-r2t}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dp2.py -q -m gpu > $O/pytest.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FAILED" $O/pytest.log | tail -5
timeout 300 python tools/wgrad_probe.py > $O/probe.txt 2>&1; grep "^wgrad" $O/probe.txt | grep "16->16\|32->32\|16->32\|32->16\|32x32 512\|64x64 256->256" | cut -c1-100
