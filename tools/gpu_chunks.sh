#!/bin/bash
# usage: tools/gpu_chunks.sh <tag> -- the minibatch in 1 / 2 / 4 / 8 gradient calls (does a chunk's H / dZ set stay in the Infinity Cache between the two kernels?)
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; out=$O/${tag}_chunks.txt; : > $out
for ch in 1 4 8 2 1; do
  echo "== chunks $ch" >> $out
  HGYM_CHUNKS=$ch HGYM_S=245760 timeout 200 python tools/bench_update.py 2>&1 | grep "minibatch\|mlp_fwd\|dw \|reduce" >> $out
done
cat $out
