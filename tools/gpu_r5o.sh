#!/bin/bash
# round 5, session o: guard OFF (captures get invalidated by collections) with the retry on fresh streams: does the eager retry survive?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5o; mkdir -p $O; cd $R
for i in 1 2 3 4 5 6; do
  SGX_CAPTURE_GC_GUARD=0 timeout 300 python -m pytest tests/test_gpu_graphs.py -q -m gpu > $O/noguard$i.log 2>&1; echo "no guard, fresh-stream retry [$i] rc=$? $(grep -ac 'StreamCaptureInvalidated' $O/noguard$i.log) invalidated; $(grep -aE 'passed|failed' $O/noguard$i.log | tail -1)"
done
grep -ah "Error\|assert" $O/noguard*.log | sort | uniq -c | sort -rn | head -8 | cut -c1-300
