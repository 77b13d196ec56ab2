#!/bin/bash
# ablation of conv_kernel phases on the probe shapes (SGX_CONV_DBG bits: 1 no MFMA, 2 no global loads, 4 no LDS stores, 8 no output stores)
cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 4 8 6 7 15; do
  SGX_CONV_DBG=$d rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abl_$d -o a -- python $GRAFT_REPO_ROOT/tools/conv_probe.py --reps 6 > /dev/null 2>&1
  echo "== SGX_CONV_DBG=$d"
  python $GRAFT_REPO_ROOT/tools/trace_summary.py $GRAFT_REPO_ROOT/gpurun_out/abl_$d/a_kernel_trace.csv conv_kernel | awk '{print $1,$2,$3,$4,$5,$6, $(NF-5), $(NF-4)}' | head -4
done
