#!/bin/bash
# Round-2 evidence in one GPU-box session: full parity suite, smoke, bench lines (headline, B=32, ffhq128 fp32), rocprofv3
# kernel statistics, SQ counter pass, HBM traffic passes.   usage: tools/gpu_final2.sh <tag> [notest]
tag=${1:-final2}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
if [ "$2" != "notest" ]; then
  timeout 1500 python -m pytest tests -q -m gpu --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
  tail -14 $O/pytest.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
fi
echo "== bench default"; timeout 600 python bench.py --layer-table $O/layers_b4.tsv 2>$O/bench_b4.err | tail -1 | tee $O/bench_b4.json | cut -c1-300
echo "== bench b32"; timeout 400 python bench.py --batch-per-gpu 32 --steps 3 --warmup 1 --no-cpu-baseline --layer-table $O/layers_b32.tsv 2>$O/bench_b32.err | tail -1 | tee $O/bench_b32.json | cut -c1-300
echo "== bench b32, single stream (per-layer times without the side stream's weight gradients sharing the GPU)"; timeout 400 python bench.py --batch-per-gpu 32 --steps 3 --warmup 1 --no-cpu-baseline --graphs off --streams 00 --layer-table $O/layers_b32_single_stream.tsv 2>/dev/null | tail -1 | tee $O/bench_b32_single_stream.json | cut -c1-200
echo "== bench ffhq128 fp32 b64"; timeout 300 python bench.py --config ffhq128 --dtype fp32 --batch-per-gpu 64 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | tee $O/bench_ffhq128_fp32_b64.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
echo "== rocprofv3 stats"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o st -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing --graphs off --streams 00 > $O/prof_bench.log 2>&1
tail -1 $O/prof_bench.log | cut -c1-200
rm -f $O/prof/*kernel_trace.csv $O/prof/*agent_info.csv
echo "== SQ counter pass"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
timeout 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/sq -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --graphs off --streams 00 > $O/sq.log 2>&1
python $R/tools/pmc_mfma.py $O/sq/p_counter_collection.csv $O/pmc_mfma_b4.json | head -30
rm -rf $O/sq
echo "== HBM traffic passes"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --graphs off --streams 00 > $O/$c.log 2>&1
done
python $R/tools/pmc_traffic.py $O/FETCH_SIZE/p_counter_collection.csv $O/WRITE_SIZE/p_counter_collection.csv $O/pmc_traffic_b4.json 0.4
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
