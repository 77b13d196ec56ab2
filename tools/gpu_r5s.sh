#!/bin/bash
# round 5, session s: bench.py's N>1 path once more at the final code (two gloo ranks on the one GPU): eager update after the all-reduce, JSON line last
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5s; mkdir -p $O; cd $R
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 6 --warmup 2 --dry-run-ranks-on-one-gpu > $O/stdout.txt 2>$O/stderr.txt; echo "rc=$?"
echo "stdout lines: $(wc -l < $O/stdout.txt); last line is JSON: $(tail -1 $O/stdout.txt | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print(True, j["value"], j["ms_per_step"], j["hip_graphs"], j["n_gpus"])')"
tail -3 $O/stderr.txt | cut -c1-200
