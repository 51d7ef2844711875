#!/bin/bash
# usage: tools/gpu_variant_update.sh <tag> <variant> ...  -- update micro-benchmark (with the shadow) per library variant, interleaved, + fb phase clock; then net / fused tests on base
tag=$1; shift; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; out=$O/${tag}_ab.txt; : > $out
lib() { if [ "$1" == "base" ]; then echo $R/humanoid-gym_amd/lib/libhgym_hip.so; else echo $R/humanoid-gym_amd/lib/variants/$1/libhgym_hip.so; fi; }
for rep in 1 2; do
  for v in "$@"; do
    echo "== $v (rep $rep)" >> $out
    HGYM_LIB=$(lib $v) HGYM_S=245760 timeout 200 python tools/bench_update.py 2>&1 | grep "calib\|minibatch\|mlp_fwd\|dw \|policy_act" >> $out
  done
done
for v in "$@"; do
  echo "== phases $v" >> $out
  HGYM_LIB=$(lib $v) HGYM_S=245760 timeout 200 python tools/probe_phases.py 2>&1 | grep "mlp_fb\|fwd<64>" >> $out
done
cat $out
timeout 600 python -m pytest tests/test_net_gpu.py tests/test_fused_gpu.py tests/test_aux_head_gpu.py -m gpu -q -x 2>&1 | tail -2
