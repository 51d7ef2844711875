#!/bin/bash
# usage: tools/gpu_fb2_ab.sh <tag> "<variant>[:ENV=val,...]" ...  -- update micro-benchmark per (library variant, environment), interleaved, 2 reps;
# then the net / fused tests on the base library.  Variant "base" = the in-tree library (64-row kernel); "base:HGYM_FB2=1" = the 128-row kernel.
tag=$1; shift; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; out=$O/${tag}_ab.txt; : > $out
lib() { if [ "$1" == "base" ]; then echo $R/humanoid-gym_amd/lib/libhgym_hip.so; else echo $R/humanoid-gym_amd/lib/variants/$1/libhgym_hip.so; fi; }
for rep in 1 2; do
  for spec in "$@"; do
    v=${spec%%:*}; e=""; if [[ "$spec" == *:* ]]; then e=$(echo "${spec#*:}" | tr ',' ' '); fi
    echo "== $spec (rep $rep)" >> $out
    env $e HGYM_LIB=$(lib $v) HGYM_S=245760 timeout 200 python tools/bench_update.py 2>&1 | grep "minibatch\|mlp_fwd\|dw " >> $out
  done
done
cat $out
if [ -z "$NO_TESTS" ]; then timeout 900 python -m pytest tests/test_net_gpu.py tests/test_fused_gpu.py -m gpu -q -x 2>&1 | tail -4; fi
