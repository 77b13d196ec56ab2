#!/bin/bash
# round 5, session a: conv3 bit-equality + probe, the rgbconv / bf16 parity tests of this round
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5a; mkdir -p $O; cd $R
rocminfo 2>/dev/null | grep -m1 "Uuid.*GPU" > $O/box.txt; git rev-parse HEAD 2>/dev/null >> $O/box.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "conv3" -x > $O/pytest_conv3.log 2>&1; echo "conv3 rc=$?"; tail -3 $O/pytest_conv3.log
timeout 600 python tools/conv2_probe.py --batch 32 --geo S D --variants 8 30 31 32 33 34 35 36 --reps 10 --check 0 > $O/probe_b32.txt 2>&1; echo "probe rc=$?"; cat $O/probe_b32.txt
timeout 300 python tools/conv2_probe.py --batch 4 --geo S D --variants 4 8 30 31 33 34 36 --reps 20 --check 0 > $O/probe_b4.txt 2>&1; cat $O/probe_b4.txt
timeout 300 python -m pytest tests/test_gpu_rgbconv.py -q -m gpu -s > $O/pytest_rgbconv.log 2>&1; echo "rgbconv rc=$?"; grep -E "passed|failed|magnitude" $O/pytest_rgbconv.log | tail -12
timeout 1500 python -m pytest tests/test_gpu_bf16.py -q -m gpu -s --durations=10 > $O/pytest_bf16.log 2>&1; echo "bf16 rc=$?"; grep -E "^\[bf16|passed|failed|Error|assert|slowest|s call" $O/pytest_bf16.log | cut -c1-600 | tail -40
