#!/bin/bash
# Round-4 evidence in one GPU-box session.   usage: tools/gpu_final4.sh <tag> [notest]
#   (the bf16 gates file is NOT re-recorded here: tests/golden/bf16_gates.json is a tripwire, the parity claim is the frozen absolute bars)
#   full parity suite + smoke; the default bench line (headline batch 4 + the batch-32 block + cpu_baseline) and its layer tables;
#   single-stream bench lines at batch 4 and 32 (per-kernel times without stream overlap) + step budgets; BASELINE configs[1];
#   the progressive sweep (BASELINE configs[4]); the 2-ranks-on-one-GPU dry run; kernel probes; and for BOTH batch sizes:
#   rocprofv3 --kernel-trace --stats, one SQ counter pass, two HBM traffic passes of the single-stream step.
# Every JSON / text artefact gets the box id (hostname + GPU unique id) and the commit the snapshot was made from (tools/.evidence_commit,
# written by the caller: `git rev-parse HEAD > tools/.evidence_commit` -- the box has no .git).
tag=${1:-final4}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
BOX="$(hostname) gpu-uid $(cat /sys/class/drm/card*/device/unique_id 2>/dev/null | head -1)"
COMMIT="$(cat tools/.evidence_commit 2>/dev/null || echo unknown)"
echo "box: $BOX   commit: $COMMIT" | tee $O/box.txt
stamp_json() { python - "$1" "$BOX" "$COMMIT" <<'EOF'
import json, sys
p, box, commit = sys.argv[1:4]
try:
    d = json.load(open(p))
except Exception:
    sys.exit(0)
if isinstance(d, dict):
    d["evidence"] = {"box": box, "commit": commit}
    json.dump(d, open(p, "w"))
EOF
}
stamp_txt() { [ -s "$1" ] && sed -i "1i # box: $BOX   commit: $COMMIT" "$1"; }
if [ "$2" = "notest" ]; then
  # (the full suite ran in pieces at this commit's kernels: see profiles/r04_gpu_tests_summary.txt) -- the order-dependent one once more
  timeout 300 python -m pytest tests/test_gpu_train.py -q -m gpu > $O/pytest_train.log 2>&1; echo "pytest train rc=$?" | tee -a $O/pytest_train.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/pytest_train.log; tail -1 $O/smoke.log
else
  timeout 1700 python -m pytest tests -q -m gpu --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
  grep -aE "passed|failed" $O/pytest.log | tail -2
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/pytest.log; tail -1 $O/smoke.log
  stamp_txt $O/pytest.log
fi
echo "== bench default (b4 + b32 blocks, cpu baseline)"; timeout 900 python bench.py --layer-table $O/layers_b4.tsv 2>$O/bench_default.err | tail -1 > $O/bench_default.json; stamp_json $O/bench_default.json; cut -c1-260 $O/bench_default.json
echo "== bench b4 single stream"; timeout 400 python bench.py --no-b32 --no-cpu-baseline --graphs off --streams 00 --layer-table $O/layers_b4_single.tsv 2>/dev/null | tail -1 > $O/bench_b4_single.json; stamp_json $O/bench_b4_single.json; cut -c1-200 $O/bench_b4_single.json
echo "== bench b32 single stream"; timeout 400 python bench.py --batch-per-gpu 32 --steps 8 --warmup 2 --no-cpu-baseline --graphs off --streams 00 --layer-table $O/layers_b32_single.tsv 2>/dev/null | tail -1 > $O/bench_b32_single.json; stamp_json $O/bench_b32_single.json; cut -c1-200 $O/bench_b32_single.json
python tools/step_budget.py $O/layers_b4_single.tsv > $O/step_budget_b4.txt 2>&1; python tools/step_budget.py $O/layers_b32_single.tsv > $O/step_budget_b32.txt 2>&1
for f in $O/layers_*.tsv $O/step_budget_*.txt; do stamp_txt $f; done
echo "== bench ffhq128 fp32 b64"; timeout 300 python bench.py --config ffhq128 --dtype fp32 --batch-per-gpu 64 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_ffhq128_fp32_b64.json; stamp_json $O/bench_ffhq128_fp32_b64.json; cut -c1-200 $O/bench_ffhq128_fp32_b64.json
echo "== sweep"; timeout 400 python bench.py --sweep 2>/dev/null | tail -1 > $O/bench_sweep.json; stamp_json $O/bench_sweep.json; cut -c1-200 $O/bench_sweep.json
echo "== dry run: 2 ranks on this GPU (gloo)"; timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --dry-run-ranks-on-one-gpu --steps 4 --warmup 1 --no-cpu-baseline --graphs off 2>/dev/null | grep '^{"metric"' | tail -1 > $O/bench_dry_run_2ranks_one_gpu.json; stamp_json $O/bench_dry_run_2ranks_one_gpu.json; cut -c1-200 $O/bench_dry_run_2ranks_one_gpu.json
for p in rgbconv_probe blur_probe rgbconv_check conv16_probe conv2_probe wgrad16_probe; do timeout 300 python tools/$p.py 2>&1 | grep -v amdgpu.ids > $O/$p.txt; stamp_txt $O/$p.txt; done
cd /tmp && export TMPDIR=/tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
for B in 4 32; do
  BA="--batch-per-gpu $B --no-b32 --no-cpu-baseline --no-kernel-timing --graphs off --streams 00"
  echo "== rocprofv3 stats, batch $B"
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b$B -o st -- python $R/bench.py --steps 3 --warmup 1 $BA > $O/prof_bench_b$B.log 2>&1
  tail -1 $O/prof_bench_b$B.log | cut -c1-160
  rm -f $O/prof_b$B/*kernel_trace.csv $O/prof_b$B/*agent_info.csv
  stamp_txt $O/prof_b$B/st_kernel_stats.csv
  echo "== SQ counter pass, batch $B"
  timeout 700 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/sq_b$B -o p -- python $R/bench.py --steps 1 --warmup 1 $BA > $O/sq_b$B.log 2>&1
  python $R/tools/pmc_mfma.py $O/sq_b$B/p_counter_collection.csv $O/pmc_mfma_b$B.json | head -16; stamp_json $O/pmc_mfma_b$B.json
  rm -rf $O/sq_b$B
  echo "== HBM traffic passes, batch $B"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 700 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${c}_b$B -o p -- python $R/bench.py --steps 1 --warmup 1 $BA > $O/${c}_b$B.log 2>&1
  done
  python $R/tools/pmc_traffic.py $O/FETCH_SIZE_b$B/p_counter_collection.csv $O/WRITE_SIZE_b$B/p_counter_collection.csv $O/pmc_traffic_b$B.json 0.4 | head -16; stamp_json $O/pmc_traffic_b$B.json
  rm -rf $O/FETCH_SIZE_b$B $O/WRITE_SIZE_b$B
done
