#!/bin/bash
# round 5, session j: BASELINE configs[1] (ffhq128 fp32 batch 64) and configs[4] (the whole progressive sweep) at HEAD, eager and replayed
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5j; mkdir -p $O; cd $R
BOX="$(hostname) gpu-uid $(cat /sys/class/drm/card*/device/unique_id 2>/dev/null | head -1)"; COMMIT="$(cat tools/.evidence_commit 2>/dev/null || echo unknown)"
stamp() { python - "$1" "$BOX" "$COMMIT" <<'EOF'
import json, sys
p, box, commit = sys.argv[1:4]
try:
    d = json.load(open(p))
except Exception:
    sys.exit(0)
d["evidence"] = {"box": box, "commit": commit}; json.dump(d, open(p, "w"))
EOF
}
timeout 300 python bench.py --config ffhq128 --dtype fp32 --batch-per-gpu 64 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_ffhq128_fp32_b64.json; stamp $O/bench_ffhq128_fp32_b64.json; cut -c1-200 $O/bench_ffhq128_fp32_b64.json
timeout 500 python bench.py --sweep --graphs on 2>/dev/null | tail -1 > $O/bench_sweep_replay.json; stamp $O/bench_sweep_replay.json
timeout 500 python bench.py --sweep --graphs off 2>/dev/null | tail -1 > $O/bench_sweep_eager.json; stamp $O/bench_sweep_eager.json
python - <<'P'
import json
for f in ("bench_sweep_eager","bench_sweep_replay"):
    d=json.load(open(f"/root/repo/gpurun_out/r5j/{f}.json"))
    print(f, round(d["value"],1), [(r["depth"], r["batch"], r["img_per_s"], r["ms_per_step"], r["frac_of_mfma_peak"]) for r in d["sweep"]])
P
