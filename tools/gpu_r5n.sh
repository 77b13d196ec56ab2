#!/bin/bash
# round 5, session n: how often does a capture next to a live RCCL group get invalidated -- with and without the garbage-collector guard
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5n; mkdir -p $O; cd $R
for i in 1 2 3 4 5; do
  SGX_CAPTURE_GC_GUARD=0 timeout 300 python -m pytest tests/test_gpu_graphs.py -q -m gpu > $O/noguard$i.log 2>&1; echo "no guard [$i] rc=$? $(grep -c 'hipErrorStreamCaptureInvalidated' $O/noguard$i.log) invalidated; $(grep -aE 'passed|failed' $O/noguard$i.log | tail -1)"
done
for i in 1 2 3 4 5 6 7; do
  timeout 300 python -m pytest tests/test_gpu_graphs.py -q -m gpu > $O/guard$i.log 2>&1; echo "guard    [$i] rc=$? $(grep -c 'hipErrorStreamCaptureInvalidated' $O/guard$i.log) invalidated; $(grep -aE 'passed|failed' $O/guard$i.log | tail -1)"
done
