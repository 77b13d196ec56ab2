#!/bin/bash
# session 1 of round 3: HEAD-of-round-2 baseline -- layer tables on ONE stream (true per-kernel times) and with the step's streams
tag=${1:-r3s1}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
echo "== b32 single stream"; timeout 400 python bench.py --batch-per-gpu 32 --steps 3 --warmup 1 --no-cpu-baseline --graphs off --streams 00 --layer-table $O/layers_b32_single.tsv 2>$O/b32s.err | tail -1 | tee $O/bench_b32_single.json | cut -c1-250
echo "== b4 single stream"; timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --graphs off --streams 00 --layer-table $O/layers_b4_single.tsv 2>$O/b4s.err | tail -1 | tee $O/bench_b4_single.json | cut -c1-250
echo "== b32 default"; timeout 400 python bench.py --batch-per-gpu 32 --steps 4 --warmup 1 --no-cpu-baseline --layer-table $O/layers_b32.tsv 2>$O/b32.err | tail -1 | tee $O/bench_b32.json | cut -c1-250
echo "== b4 default"; timeout 400 python bench.py --no-cpu-baseline --layer-table $O/layers_b4.tsv 2>$O/b4.err | tail -1 | tee $O/bench_b4.json | cut -c1-400
