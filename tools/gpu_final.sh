#!/bin/bash
# Round-end evidence in one GPU-box session: full parity suite, smoke, the three bench lines, rocprofv3 kernel statistics.
# usage: tools/gpu_final.sh <tag>
tag=${1:-final}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
echo "== bench default"; timeout 600 python bench.py --layer-table $O/layers.txt 2>&1 | tail -1 | tee $O/bench_default.json | cut -c1-400
echo "== bench ffhq128 fp32 b64"; timeout 300 python bench.py --config ffhq128 --dtype fp32 --batch-per-gpu 64 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee $O/bench_ffhq128_fp32_b64.json | cut -c1-200
echo "== bench b32"; timeout 400 python bench.py --batch-per-gpu 32 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee $O/bench_b32.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $tag -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/prof_bench.log 2>&1
tail -1 $O/prof_bench.log | cut -c1-200
ls $O/prof | head
rm -f $O/prof/*kernel_trace.csv                                   # tens of MB; the statistics are what is kept
