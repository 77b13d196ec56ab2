#!/bin/bash
# round 5, session e: residual-in-the-store with the LDS weight table, sweep under replay
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5e; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_rgbconv.py tests/test_gpu_fusions.py -q -m gpu -k "fade or statistics" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|FAILED" $O/pytest.log | tail -8
B32="--batch-per-gpu 32 --no-b32 --steps 6 --warmup 2 --graphs off --streams 00 --no-cpu-baseline"
SGX_FUSE_FADE_RGB=1 timeout 600 python bench.py $B32 --layer-table $O/layers_b32_fadergb1.tsv > $O/bench_b32_fadergb1.json 2> $O/err1.txt; tail -1 $O/bench_b32_fadergb1.json | cut -c1-190
SGX_FUSE_FADE_RGB=0 timeout 600 python bench.py $B32 --layer-table $O/layers_b32_fadergb0.tsv > $O/bench_b32_fadergb0.json 2> $O/err0.txt; tail -1 $O/bench_b32_fadergb0.json | cut -c1-190
grep -E "convD\+fade" $O/layers_b32_fadergb*.tsv | cut -c1-200
timeout 600 python bench.py --sweep --sweep-depths 6,7,8 --graphs on > $O/sweep_replay.json 2> $O/err_s1.txt; tail -3 $O/err_s1.txt
timeout 600 python bench.py --sweep --sweep-depths 6,7,8 --graphs off > $O/sweep_eager.json 2> $O/err_s0.txt
python - <<'P'
import json
for f in ("sweep_eager","sweep_replay"):
    try:
        d=json.loads(open(f"/root/repo/gpurun_out/r5e/{f}.json").read().strip().split("\n")[-1])
        print(f, [(r["depth"], r["batch"], r["img_per_s"], r["ms_per_step"], r["host_enqueue_ms_per_step"]) for r in d["sweep"]])
    except Exception as e: print(f, "n/a", e)
P
