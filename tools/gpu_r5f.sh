#!/bin/bash
# round 5, session f: upblur mask-word loads, then the WHOLE gpu suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5f; mkdir -p $O; cd $R
timeout 300 python tools/upblur_probe.py 32 4 2>&1 | grep -v "amdgpu.ids\|no fused" | tee $O/upblur_probe.txt
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu --durations=15 > $O/pytest_all.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - t0 )) s"; grep -E "passed|failed|Error|FAILED" $O/pytest_all.log | tail -8; grep -A18 "slowest" $O/pytest_all.log | tail -17
