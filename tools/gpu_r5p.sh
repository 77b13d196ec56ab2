#!/bin/bash
# round 5, session p (last): at HEAD -- bench.py's N>1 path as two gloo ranks on the one GPU, then the whole -m gpu suite and smoke()
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5p; mkdir -p $O; cd $R
BOX="$(hostname) gpu-uid $(cat /sys/class/drm/card*/device/unique_id 2>/dev/null | head -1)"; COMMIT="$(cat tools/.evidence_commit 2>/dev/null || echo unknown)"
echo "box: $BOX   commit: $COMMIT" | tee $O/box.txt
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 --dry-run-ranks-on-one-gpu 2>$O/dry2.err | tail -1 > $O/bench_dry_run_2ranks_one_gpu.json; echo "dry-run rc=$?"; cut -c1-400 $O/bench_dry_run_2ranks_one_gpu.json; tail -3 $O/dry2.err | cut -c1-300
t0=$(date +%s); timeout 1000 python -m pytest tests -q -m gpu --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - t0 )) s" | tee -a $O/pytest.log
grep -aE "passed|failed" $O/pytest.log | tail -2; sed -i "1i # box: $BOX   commit: $COMMIT" $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
