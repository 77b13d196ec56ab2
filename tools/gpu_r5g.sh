#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5g; mkdir -p $O; cd $R
{ timeout 300 python tools/repro_check.py kernel
  SGX_CONV_UPBLUR=1 SGX_FUSE_FADE_RGB=1 timeout 300 python tools/repro_check.py; } 2>&1 | grep -v amdgpu.ids | tee $O/repro.txt
timeout 600 python -m pytest tests/test_gpu_fusions.py tests/test_gpu_fullsize.py -q -m gpu -k "upblur or up_blur or fused_blur or fused_up or reproducible" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED" $O/pytest.log | tail -5
timeout 300 python tools/upblur_probe.py 32 4 2>&1 | grep -v "amdgpu.ids\|no fused" | tee $O/upblur_probe.txt
