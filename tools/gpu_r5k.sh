#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5k; mkdir -p $O; cd $R
for i in 1 2 3; do
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_graphs.py -q -m gpu -s > $O/pytest$i.log 2>&1; echo "pytest[$i] rc=$?"; grep -E "passed|failed|FAILED|launches \(host|capture of" $O/pytest$i.log | tail -8
done
