#!/bin/bash
tag=${1:-r2n}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
echo "== bench default"; timeout 500 python bench.py --no-cpu-baseline 2>$O/bench_b4.err | tail -1 | tee $O/bench_b4.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','hip_graphs','aux_stream','side_stream','launch_mode_calibration','host_enqueue_ms_per_step')}); print(d['roofline']['kernel'][:60], d['roofline']['frac'])"
tail -3 $O/bench_b4.err
timeout 900 python -m pytest tests/test_gpu_graphs.py -q -m gpu -x > $O/pytest_graphs.log 2>&1; echo "graphs rc=$?"; tail -3 $O/pytest_graphs.log
