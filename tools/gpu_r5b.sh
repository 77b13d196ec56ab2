#!/bin/bash
# round 5, session b: what bounds conv3 / conv2 -- DMA ablations of configuration 31 (8 waves x 2 rows, fragments two sub-steps ahead)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O; cd $R
for dbg in 0 2 1 3 16 8 24 4; do
  echo "== SGX_CONV3_DBG=$dbg (1 patch contiguous, 2 weights contiguous, 4 no DMA, 8 no patch DMA on odd steps, 16 no weight DMA)"
  SGX_CONV3_DBG=$dbg timeout 300 python tools/conv2_probe.py --batch 32 --geo S D --variants 31 33 --reps 10 --check 0 2>&1 | grep -v amdgpu.ids
done > $O/ablate.txt 2>&1
cat $O/ablate.txt
