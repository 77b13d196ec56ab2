#!/bin/bash
# round-2 session K: second-generation weight-gradient kernels -- parity, then the A/B probe
tag=${1:-r2k}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wgrad2 or bias_gradient" > $O/pytest_wgrad2.log 2>&1; echo "wgrad2 rc=$?"; grep -E "passed|failed|FAILED|rel-L2|Error" $O/pytest_wgrad2.log | tail -30
SGX_WGRAD2=0 timeout 300 python tools/wgrad_probe.py > $O/probe_v1.txt 2>&1; SGX_WGRAD2=3 timeout 300 python tools/wgrad_probe.py > $O/probe_v2.txt 2>&1
paste -d'|' <(grep "^wgrad" $O/probe_v1.txt | cut -c1-75) <(grep "^wgrad" $O/probe_v2.txt | cut -c30-75)
