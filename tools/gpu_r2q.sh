#!/bin/bash
tag=${1:-r2q}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_networks.py tests/test_gpu_kernels.py tests/test_gpu_graphs.py tests/test_gpu_flags.py -q -m gpu -x > $O/pytest.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FAILED|Error|assert" $O/pytest.log | tail -12
for i in 1; do
echo "== bench default ($i)"; timeout 600 python bench.py --no-cpu-baseline --layer-table $O/layers_b4.tsv 2>$O/bench_b4.err | tail -1 | tee $O/bench_b4_$i.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','hip_graphs','aux_stream','side_stream','launch_mode_calibration','host_enqueue_ms_per_step')}); print(d['roofline']['library_launches_per_step'], d['roofline']['library_kernels_ms_per_step'])"
done
echo "== bench b32"; timeout 400 python bench.py --batch-per-gpu 32 --steps 3 --warmup 1 --no-cpu-baseline --layer-table $O/layers_b32.tsv 2>$O/bench_b32.err | tail -1 | tee $O/bench_b32.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','hip_graphs','aux_stream','side_stream')})"
