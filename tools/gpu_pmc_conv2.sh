#!/bin/bash
# SQ counter passes over ONE 3x3 layer shape of the conv2 probe (8-wave variant, batch 32): where the wave cycles of the MFMA-bound
# layers go.   usage: tools/gpu_pmc_conv2.sh <tag> <H> [ENV=..]
tag=$1; H=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU"
B="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_SALU SQ_INSTS_LDS"
C="SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_IFETCH SQ_WAVES"
for p in A B C; do
  timeout 300 rocprofv3 --kernel-trace --pmc ${!p} --output-format csv -d $O/$p -o p -- python $R/tools/conv2_probe.py --geo S --batch 32 --variants 8 --check 0 --reps 3 --only-h $H > $O/$p.log 2>&1
  echo "== pass $p rc=$?"; python $R/tools/pmc_summary.py $O/$p/p_counter_collection.csv conv2
done
