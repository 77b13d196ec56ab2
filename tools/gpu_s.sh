#!/bin/bash
# One GPU-box session: a pytest selection, then (optionally) bench lines.
#   usage: tools/gpu_s.sh <tag> "<pytest args>" ["<bench args>" ...]      (each bench arg string = one bench.py run)
tag=$1; sel=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
if [ -n "$sel" ]; then
  timeout 1500 python -m pytest $sel -q -m gpu $PYTEST_EXTRA --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?"
  grep -E "passed|failed|Error|FAILED|assert|rel-L2" $O/pytest.log | tail -30
fi
i=0
for b in "$@"; do
  i=$((i+1))
  echo "== bench $b"
  timeout 900 python bench.py $b --layer-table $O/layers_$i.tsv > $O/bench_$i.json 2> $O/bench_$i.err; echo "rc=$?"
  tail -1 $O/bench_$i.json | cut -c1-330; tail -3 $O/bench_$i.err | cut -c1-300
done
