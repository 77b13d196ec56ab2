#!/bin/bash
# round 5, session i: sign-word stores + pooled-image prefetch in the stride-2 lerp store
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5i; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_rgbconv.py tests/test_gpu_fusions.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -q -m gpu -k "fade or sign_bits or signbits or conv2 or conv3 or reproducible or discriminator_block or composed" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED" $O/pytest.log | tail -6
B32="--batch-per-gpu 32 --no-b32 --steps 6 --warmup 2 --graphs off --streams 00 --no-cpu-baseline"
timeout 600 python bench.py $B32 --layer-table $O/layers_b32.tsv > $O/bench_b32.json 2> $O/err1.txt; tail -1 $O/bench_b32.json | cut -c1-190
grep -E "convD\+fade|convS\+bits|convD B32 1024" $O/layers_b32.tsv | cut -f1,3,4,7
SGX_HIP_LIB= timeout 400 python bench.py --no-b32 --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | cut -c1-200
