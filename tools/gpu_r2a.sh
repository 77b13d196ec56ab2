#!/bin/bash
# round-2 session A: new conv kernel probe + parity + regression of the whole suite + quick bench
tag=${1:-r2a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k conv2 > $O/pytest_conv2.log 2>&1; echo "conv2 tests rc=$?"; tail -5 $O/pytest_conv2.log
timeout 300 python tools/conv2_probe.py --reps 10 > $O/probe.log 2>&1; echo "probe rc=$?"; cat $O/probe.log
timeout 600 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
echo "== bench default"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --layer-table $O/layers_b4.txt 2>&1 | tail -1 | tee $O/bench_b4.json | cut -c1-600
echo "== bench b32"; timeout 400 python bench.py --batch-per-gpu 32 --steps 3 --warmup 1 --no-cpu-baseline --layer-table $O/layers_b32.txt 2>&1 | tail -1 | tee $O/bench_b32.json | cut -c1-600
