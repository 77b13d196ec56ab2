#!/bin/bash
# Quick GPU-box session: parity tests + default bench with the per-layer table.   usage: tools/gpu_quick.sh <tag> [pytest -k expr]
tag=${1:-qX}; kexpr=${2:-}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
if [ -n "$kexpr" ]; then timeout 900 python -m pytest tests -q -m gpu -x -k "$kexpr" > $O/pytest.log 2>&1; else timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; fi
echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --layer-table $O/layers.txt 2>&1 | tail -1 > $O/bench.json
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print("img/s", round(d["value"],2), "ms/step", round(d["ms_per_step"],2), "dominant", d["roofline"]["kernel"][:60], "frac", d["roofline"]["frac"], "lib ms", d["roofline"]["library_kernels_ms_per_step"], "launches", d["roofline"]["library_launches_per_step"], "host enqueue ms", round(d["host_enqueue_ms_per_step"],2))
PY
