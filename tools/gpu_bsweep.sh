#!/bin/bash
# usage: tools/gpu_bsweep.sh <tag> -- the minibatch kernels over batch sizes around 61 440 (is mlp_fb_kernel's time a staircase in tiles / CUs?)
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; out=$O/${tag}_bsweep.txt; : > $out
for b in 49152 53248 57344 59392 61440 63488 65536 61440; do
  echo "== B $b" >> $out
  HGYM_B=$b HGYM_S=245760 timeout 200 python tools/bench_update.py 2>&1 | grep "minibatch\|mlp_fwd\|dw " >> $out
done
cat $out
