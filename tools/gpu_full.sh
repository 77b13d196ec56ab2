#!/bin/bash
# whole GPU suite + the two bench lines (+ layer tables)
tag=${1:-full}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
echo "== bench default"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --layer-table $O/layers_b4.txt 2>&1 | tail -1 | tee $O/bench_b4.json | cut -c1-700
echo "== bench b32"; timeout 400 python bench.py --batch-per-gpu 32 --steps 3 --warmup 1 --no-cpu-baseline --layer-table $O/layers_b32.txt 2>&1 | tail -1 | tee $O/bench_b32.json | cut -c1-400
