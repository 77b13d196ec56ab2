#!/bin/bash
# run arbitrary commands on the GPU box with logs under gpurun_out/<tag>/.   usage: tools/gpu_cmd.sh <tag> "<cmd>" ["<cmd>" ...]
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
i=0
for c in "$@"; do
  i=$((i+1)); echo "== $c"
  ( eval "$c" ) > $O/cmd_$i.log 2>&1; echo "rc=$?"; tail -25 $O/cmd_$i.log | cut -c1-400
done
