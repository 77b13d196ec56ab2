#!/bin/bash
# round-2 session L: regression (kernels, networks, real configs) + bench lines with wgrad2 on
tag=${1:-r2l}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_networks.py tests/test_gpu_fullsize.py -q -m gpu -x > $O/pytest_a.log 2>&1; echo "kernels+networks+fullsize rc=$?"; tail -4 $O/pytest_a.log
echo "== bench default"; timeout 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --layer-table $O/layers_b4.tsv 2>$O/bench_b4.err | tail -1 | tee $O/bench_b4.json | cut -c1-420
echo "== bench b32"; timeout 400 python bench.py --batch-per-gpu 32 --steps 3 --warmup 1 --no-cpu-baseline --layer-table $O/layers_b32.tsv 2>$O/bench_b32.err | tail -1 | tee $O/bench_b32.json | cut -c1-420
