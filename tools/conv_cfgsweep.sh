#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for cfg in "4 256" "2 256" "1 256" "4 128" "2 128" "2 64"; do
  set -- $cfg
  SGX_CONV_MAXCT=$1 SGX_CONV_MAXBP=$2 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cfg_$1_$2 -o a -- python $GRAFT_REPO_ROOT/tools/conv_probe.py --reps 6 > /dev/null 2>&1
  echo "== MAXCT=$1 MAXBP=$2"
  python $GRAFT_REPO_ROOT/tools/trace_summary.py $GRAFT_REPO_ROOT/gpurun_out/cfg_$1_$2/a_kernel_trace.csv conv_kernel | awk '{printf "%s %s %s %s %s %s %s | grid %s | %s us\n", $1,$2,$3,$4,$5,$6,$7,$9,$(NF-5)}'
done
