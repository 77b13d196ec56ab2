#!/bin/bash
# round-2 session G: bench lines (B=4, B=32) with all three conv2 geometries on, rocprofv3 kernel statistics, one SQ counter
# pass (MFMA busy / MFMA ops / LDS conflicts / LDS issue stalls) and the two HBM traffic passes.   usage: tools/gpu_r2g.sh <tag>
tag=${1:-r2g}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
echo "== bench default"; timeout 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --layer-table $O/layers_b4.tsv 2>$O/bench_b4.err | tail -1 | tee $O/bench_b4.json | cut -c1-700
echo "== bench b32"; timeout 400 python bench.py --batch-per-gpu 32 --steps 3 --warmup 1 --no-cpu-baseline --layer-table $O/layers_b32.tsv 2>$O/bench_b32.err | tail -1 | tee $O/bench_b32.json | cut -c1-700
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_avail.txt 2>&1; grep -c . $O/counters_avail.txt
echo "== rocprofv3 stats"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o st -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing --graphs off > $O/prof_bench.log 2>&1
tail -1 $O/prof_bench.log | cut -c1-200
ls $O/prof | head; rm -f $O/prof/*kernel_trace.csv
echo "== SQ counter pass"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
timeout 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/sq -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --graphs off > $O/sq.log 2>&1
tail -2 $O/sq.log | cut -c1-200
if [ -f $O/sq/p_counter_collection.csv ]; then
  python $R/tools/pmc_mfma.py $O/sq/p_counter_collection.csv $O/pmc_mfma_b4.json | head -40
else
  echo "one 8-counter pass refused; two passes of 4"
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/sq$i -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --graphs off > $O/sq$i.log 2>&1
    tail -1 $O/sq$i.log | cut -c1-160
  done
  python $R/tools/pmc_mfma.py $O/sq1/p_counter_collection.csv $O/pmc_mfma_b4.json $O/sq2/p_counter_collection.csv | head -40
fi
rm -rf $O/sq $O/sq1 $O/sq2
echo "== HBM traffic passes"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --graphs off > $O/$c.log 2>&1
done
python $R/tools/pmc_traffic.py $O/FETCH_SIZE/p_counter_collection.csv $O/WRITE_SIZE/p_counter_collection.csv $O/pmc_traffic_b4.json 0.4
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
