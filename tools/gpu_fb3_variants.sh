#!/bin/bash
# usage: tools/gpu_fb3_variants.sh <tag> <variant> ...   -- per library variant (lib/variants/<v>/, `base` = the default library) with HGYM_FB3=1:
# the equality test against mlp_fb_kernel, the update micro-benchmark (B = 61 440 of 245 760 rows), the tile's phase clock; HGYM_FB3=0 first as the reference
tag=$1; shift
mkdir -p gpurun_out
out=gpurun_out/${tag}_fb3_variants.txt
: > $out
run() {   # name, env...
  name=$1; shift
  echo "== $name" >> $out
  env "$@" HGYM_S=245760 timeout 300 python tools/bench_update.py 2>&1 | grep "mlp_fwd\|dw  \|minibatch" >> $out
  env "$@" timeout 300 python tools/probe_phases.py 2>&1 | grep "mlp_fb<64>" >> $out
}
run "mlp_fb_kernel (HGYM_FB3=0)" HGYM_FB3=0
for v in "$@"; do
  if [ "$v" = base ]; then L=""; else L="HGYM_LIB=$PWD/humanoid-gym_amd/lib/variants/$v/libhgym_hip.so"; fi
  env $L HGYM_FB3=1 timeout 300 python -m pytest tests/test_fused_gpu.py -m gpu -q -x -k "role_specialised" 2>&1 | tail -1 | sed "s/^/[$v] /" >> $out
  run "fb3 $v" $L HGYM_FB3=1
done
run "mlp_fb_kernel (HGYM_FB3=0) again" HGYM_FB3=0
cut -c1-260 $out
