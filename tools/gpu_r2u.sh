#!/bin/bash
# XCD-band tile order of the convolutions: parity, then same-box A/B (SGX_TILE_BANDS=0/1)
tag=${1:-r2u}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -q -m gpu -k "conv or adjoint or full or deterministic or linear" > $O/pytest.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FAILED" $O/pytest.log | tail -5
for rep in 1 2; do for f in 0 1; do
  r=$(SGX_TILE_BANDS=$f timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --graphs on --streams 11 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step'%d['ms_per_step'])")
  echo "B=4 graph bands=$f: $r"
  r=$(SGX_TILE_BANDS=$f timeout 300 python bench.py --batch-per-gpu 32 --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing --graphs off --streams 11 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f ms/step'%d['ms_per_step'])")
  echo "B=32 eager bands=$f: $r"
done; done
