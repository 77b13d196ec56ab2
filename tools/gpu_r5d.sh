#!/bin/bash
# round 5, session d: residual-in-the-store fusion, train-with-replay, regression subset, A/B benches, sweep with replay
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5d; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_rgbconv.py tests/test_gpu_networks.py tests/test_gpu_graphs.py tests/test_gpu_train.py tests/test_gpu_fusions.py -q -m gpu --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|FAILED" $O/pytest.log | tail -20
B32="--batch-per-gpu 32 --no-b32 --steps 6 --warmup 2 --graphs off --streams 00 --no-cpu-baseline"
SGX_FUSE_FADE_RGB=1 timeout 600 python bench.py $B32 --layer-table $O/layers_b32_fadergb1.tsv > $O/bench_b32_fadergb1.json 2> $O/err1.txt; tail -1 $O/bench_b32_fadergb1.json | cut -c1-190
SGX_FUSE_FADE_RGB=0 timeout 600 python bench.py $B32 --layer-table $O/layers_b32_fadergb0.tsv > $O/bench_b32_fadergb0.json 2> $O/err0.txt; tail -1 $O/bench_b32_fadergb0.json | cut -c1-190
SGX_FUSE_EPI_STATS_MIN=33554432 timeout 600 python bench.py $B32 > $O/bench_b32_stats25.json 2> $O/err2.txt; tail -1 $O/bench_b32_stats25.json | cut -c1-190
SGX_FUSE_EPI_STATS_MIN=8388608 timeout 600 python bench.py $B32 > $O/bench_b32_stats23.json 2> $O/err3.txt; tail -1 $O/bench_b32_stats23.json | cut -c1-190
timeout 900 python bench.py --layer-table $O/layers_default.tsv > $O/bench_default.json 2> $O/err_default.txt; tail -1 $O/bench_default.json | cut -c1-400
timeout 600 python bench.py --sweep --sweep-depths 6,7,8 --graphs off > $O/sweep_eager.json 2> $O/err_s0.txt; python - <<'P'
import json
for f in ("sweep_eager","sweep_replay"):
    try:
        d=json.loads(open(f"/root/repo/gpurun_out/r5d/{f}.json").read().strip().split("\n")[-1])
        print(f, [(r["depth"], r["batch"], r["img_per_s"], r["ms_per_step"], r["host_enqueue_ms_per_step"]) for r in d["sweep"]])
    except Exception as e: print(f, "n/a", e)
P
timeout 600 python bench.py --sweep --sweep-depths 6,7,8 --graphs on > $O/sweep_replay.json 2> $O/err_s1.txt; python - <<'P'
import json
for f in ("sweep_replay",):
    try:
        d=json.loads(open(f"/root/repo/gpurun_out/r5d/{f}.json").read().strip().split("\n")[-1])
        print(f, [(r["depth"], r["batch"], r["img_per_s"], r["ms_per_step"], r["host_enqueue_ms_per_step"]) for r in d["sweep"]])
    except Exception as e: print(f, "n/a", e)
P
