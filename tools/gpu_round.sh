#!/bin/bash
# One GPU-box session: parity tests, conv probes (default / experimental lib), benches, rocprof summary.
# usage: tools/gpu_round.sh <tag>
tag=${1:-rX}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
echo "== probe default"; timeout 200 python tools/conv_probe.py --reps 10 2>&1 | tee $O/probe_default.log
if [ -f stylegan/pytorch_amd/csrc/build/exp/libsgx_occ4.so ]; then
  echo "== probe occ4"; SGX_HIP_LIB=$R/stylegan/pytorch_amd/csrc/build/exp/libsgx_occ4.so timeout 200 python tools/conv_probe.py --reps 10 2>&1 | tee $O/probe_occ4.log
fi
echo "== bench default"; timeout 600 python bench.py --steps 6 --warmup 2 --layer-table $O/layers.txt 2>&1 | tail -1 | tee $O/bench_default.json
echo "== bench ffhq128 fp32 b64"; timeout 300 python bench.py --config ffhq128 --dtype fp32 --batch-per-gpu 64 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee $O/bench_ffhq128_fp32_b64.json
echo "== bench b32"; timeout 400 python bench.py --batch-per-gpu 32 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee $O/bench_b32.json
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $tag -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/prof_bench.log 2>&1
tail -1 $O/prof_bench.log
ls $O/prof | head
