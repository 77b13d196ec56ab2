#!/bin/bash
# A/B of one pytest selection under two environments.   usage: tools/gpu_ab_env.sh <tag> "<pytest args>" "<env A>" "<env B>"
tag=$1; sel=$2; ea=$3; eb=$4
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
for v in A B; do
  if [ $v = A ]; then e="$ea"; else e="$eb"; fi
  echo "== $v: $e"
  env $e timeout 900 python -m pytest $sel -q -m gpu --durations=3 > $O/pytest_$v.log 2>&1; echo "rc=$?"
  grep -E "passed|failed|^E  .*assert|FAILED" $O/pytest_$v.log | tail -8
done
