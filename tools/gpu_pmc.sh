#!/bin/bash
# HBM traffic per kernel launch of the default bench: two PMC passes (FETCH_SIZE, WRITE_SIZE), kernel trace only.
tag=${1:-pmcX}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --graphs off > $O/$c.log 2>&1
  ls $O/$c | head -5
done
python $R/tools/pmc_traffic.py $O/FETCH_SIZE/p_counter_collection.csv $O/WRITE_SIZE/p_counter_collection.csv $O/pmc_traffic.json 0.4
rm -f $O/FETCH_SIZE/p_counter_collection.csv.keep
# keep the merged output small: drop the raw per-dispatch tables (tens of MB), keep the summary
du -sh $O/FETCH_SIZE $O/WRITE_SIZE
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
