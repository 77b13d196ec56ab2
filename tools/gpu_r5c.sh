#!/bin/bash
# round 5, session c: the composite up+blur kernel -- parity, probe, step bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_fusions.py -q -m gpu -s -k "upblur or up_blur or fused_blur or fused_up" > $O/pytest_fusions.log 2>&1; echo "fusions rc=$?"; grep -E "^\[upblur3|passed|failed|Error|assert" $O/pytest_fusions.log | cut -c1-300 | tail -50
timeout 300 python tools/upblur_probe.py 32 4 2>&1 | grep -v amdgpu.ids | tee $O/upblur_probe.txt
for b in "--batch-per-gpu 32 --no-b32 --steps 6 --warmup 2 --graphs off --streams 00" ; do
  SGX_CONV_UPBLUR=1 timeout 600 python bench.py $b --layer-table $O/layers_b32_new.tsv > $O/bench_b32_new.json 2> $O/bench_b32_new.err; tail -1 $O/bench_b32_new.json | cut -c1-200
  SGX_CONV_UPBLUR=0 timeout 600 python bench.py $b --layer-table $O/layers_b32_old.tsv > $O/bench_b32_old.json 2> $O/bench_b32_old.err; tail -1 $O/bench_b32_old.json | cut -c1-200
done
