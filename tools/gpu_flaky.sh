#!/bin/bash
# Repeat a test selection N times (chaos-bounded comparisons must hold on every draw).  usage: tools/gpu_flaky.sh <tag> <N> <pytest args...>
tag=$1; n=$2; shift; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
for i in $(seq 1 $n); do timeout 600 python -m pytest -q -m gpu "$@" > $O/run_$i.log 2>&1; echo "run $i rc=$?: $(tail -1 $O/run_$i.log)"; done
