#!/bin/bash
# usage: tools/gpu_variant_time.sh <tag> <variant> ...  -- update micro-benchmark only (timing experiments: results of ablated variants are wrong by design)
tag=$1; shift; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; out=$O/${tag}_time.txt; : > $out
lib() { if [ "$1" == "base" ]; then echo $R/humanoid-gym_amd/lib/libhgym_hip.so; else echo $R/humanoid-gym_amd/lib/variants/$1/libhgym_hip.so; fi; }
for rep in 1 2; do
  for v in "$@"; do
    echo "== $v (rep $rep)" >> $out
    HGYM_LIB=$(lib $v) HGYM_S=245760 timeout 200 python tools/bench_update.py 2>&1 | grep "minibatch\|mlp_fwd\|dw " >> $out
  done
done
cat $out
