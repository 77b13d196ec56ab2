#!/bin/bash
# Round-6 evidence in one GPU-box session.   usage: tools/gpu_final6.sh <tag> [test|notest] [ab]
#   smoke; the default bench line (headline batch 4 + the batch-32 block + cpu_baseline) and its layer tables; single-stream bench lines at
#   batch 4 and 32 (per-kernel times without stream overlap) + step budgets; and for BOTH batch sizes: rocprofv3 --kernel-trace --stats,
#   one SQ counter pass, two HBM traffic passes of the single-stream step.  `test`: the whole -m gpu suite first.
# Every JSON / text artefact gets the box id and the commit the snapshot was made from (tools/.evidence_commit, written by the caller).
tag=${1:-final6}
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
if [ "$2" = "test" ]; then
  t0=$(date +%s); timeout 1500 python -m pytest tests -q -m gpu --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - t0 )) s" | tee -a $O/pytest.log
  grep -aE "passed|failed" $O/pytest.log | tail -2; stamp_txt $O/pytest.log
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
echo "== bench default (b4 + b32 blocks, cpu baseline)"; timeout 900 python bench.py --layer-table $O/layers_b4.tsv 2>$O/bench_default.err | tail -1 > $O/bench_default.json; stamp_json $O/bench_default.json; cp gpurun_out/bench_detail.json $O/bench_default_detail.json 2>/dev/null; wc -c < $O/bench_default.json; cut -c1-260 $O/bench_default.json
echo "== bench b4 single stream"; timeout 400 python bench.py --no-b32 --no-extras --no-cpu-baseline --graphs off --streams 00 --layer-table $O/layers_b4_single.tsv 2>/dev/null | tail -1 > $O/bench_b4_single.json; stamp_json $O/bench_b4_single.json; cut -c1-200 $O/bench_b4_single.json
echo "== bench b32 single stream"; timeout 400 python bench.py --batch-per-gpu 32 --steps 8 --warmup 2 --no-cpu-baseline --graphs off --streams 00 --layer-table $O/layers_b32_single.tsv 2>/dev/null | tail -1 > $O/bench_b32_single.json; stamp_json $O/bench_b32_single.json; cut -c1-200 $O/bench_b32_single.json
python tools/step_budget.py $O/layers_b4_single.tsv > $O/step_budget_b4.txt 2>&1; python tools/step_budget.py $O/layers_b32_single.tsv > $O/step_budget_b32.txt 2>&1
for f in $O/layers_*.tsv $O/step_budget_*.txt; do stamp_txt $f; done
if [ "$3" = "ab" ]; then
  OFF1="SGX_GEPI_SMALL=0 SGX_CONV2_SMALL=0"
  OFF2="SGX_GEPI_APPLY1=0 SGX_GEPI_BWD2S=0 SGX_GRID_ALL_CAP=8192 SGX_CONV_SPLITK=0 SGX_FUSE_FADE_BWD2=0 SGX_POOL_FORK=0 SGX_RGB_FORK=0"
  echo "== A/B on this box, default bench line, interleaved: [all off] = the round's switches off ($OFF1 $OFF2); [session 1] = only the second session's off ($OFF2); [on] = defaults"
  for i in 1 2; do for v in alloff session1 on; do
    if [ $v = alloff ]; then e="$OFF1 $OFF2"; elif [ $v = session1 ]; then e="$OFF2"; else e="SGX_AB_DUMMY=1"; fi
    env $e timeout 400 python bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('[$v] b4', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 3), 'ms graphs', d.get('hip_graphs'), 'launches', d.get('library_launches_per_step'), '| b32', round(d['b32']['value'], 1), 'img/s', round(d['b32']['ms_per_step'], 2), 'ms')"
  done; done | tee $O/ab_round6.txt
  stamp_txt $O/ab_round6.txt
  echo "== probes of the second session"
  timeout 120 tools/stream_probe > $O/stream_probe.txt 2>&1; stamp_txt $O/stream_probe.txt; head -12 $O/stream_probe.txt
  timeout 200 python tools/overlap_probe.py 2>/dev/null > $O/overlap_probe.txt; stamp_txt $O/overlap_probe.txt; cat $O/overlap_probe.txt
  timeout 200 python tools/splitk_probe.py --batch 4 8 2>/dev/null > $O/splitk_probe.txt; stamp_txt $O/splitk_probe.txt
  ( for a in 0 1; do SGX_GEPI_APPLY1=$a SGX_GEPI_BWD2S=$a python tools/gepi_probe.py --batch 32 4 --min-h 128 --reps 10 2>&1 | grep epilogue | sed "s/^/[short-lived apply passes $a] /"; done ) > $O/gepi_probe.txt; stamp_txt $O/gepi_probe.txt
fi
cd /tmp && export TMPDIR=/tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
for B in 4 32; do
  BA="--batch-per-gpu $B --no-b32 --no-extras --no-cpu-baseline --no-kernel-timing --graphs off --streams 00"
  echo "== rocprofv3 stats, batch $B"
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b$B -o st -- python $R/bench.py --steps 3 --warmup 1 $BA > $O/prof_bench_b$B.log 2>&1
  tail -1 $O/prof_bench_b$B.log | cut -c1-160
  rm -f $O/prof_b$B/*kernel_trace.csv $O/prof_b$B/*agent_info.csv
  stamp_txt $O/prof_b$B/st_kernel_stats.csv
  echo "== SQ counter pass, batch $B"
  timeout 700 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/sq_b$B -o p -- python $R/bench.py --steps 1 --warmup 1 $BA > $O/sq_b$B.log 2>&1
  python $R/tools/pmc_mfma.py $O/sq_b$B/p_counter_collection.csv $O/pmc_mfma_b$B.json | head -12; stamp_json $O/pmc_mfma_b$B.json
  rm -rf $O/sq_b$B
  echo "== HBM traffic passes, batch $B"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 700 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${c}_b$B -o p -- python $R/bench.py --steps 1 --warmup 1 $BA > $O/${c}_b$B.log 2>&1
  done
  python $R/tools/pmc_traffic.py $O/FETCH_SIZE_b$B/p_counter_collection.csv $O/WRITE_SIZE_b$B/p_counter_collection.csv $O/pmc_traffic_b$B.json 0.4 | head -12; stamp_json $O/pmc_traffic_b$B.json
  rm -rf $O/FETCH_SIZE_b$B $O/WRITE_SIZE_b$B
done
