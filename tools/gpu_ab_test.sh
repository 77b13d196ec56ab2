#!/bin/bash
# Kernel parity subset, then an A/B of the bench.   usage: tools/gpu_ab_test.sh <tag> "<pytest -k expr>" "ENV=.. -- --flags" ...
tag=$1; kexpr=$2; shift; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -q -m gpu -x -k "$kexpr" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
exec bash tools/gpu_ab.sh $tag "$@"
