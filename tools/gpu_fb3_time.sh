#!/bin/bash
# usage: tools/gpu_fb3_time.sh <tag> <variant> ...  -- timing only (ablated variants, results wrong by design): update micro-benchmark + phase clock per variant
tag=$1; shift
mkdir -p gpurun_out
out=gpurun_out/${tag}_fb3_time.txt
: > $out
run() { name=$1; shift; echo "== $name" >> $out
  env "$@" HGYM_S=245760 timeout 300 python tools/bench_update.py 2>&1 | grep "mlp_fwd\|dw  " >> $out
  env "$@" timeout 300 python tools/probe_phases.py 2>&1 | grep "mlp_fb<64>" >> $out; }
run "mlp_fb_kernel (HGYM_FB3=0)" HGYM_FB3=0
for v in "$@"; do
  if [ "$v" = base ]; then L=""; else L="HGYM_LIB=$PWD/humanoid-gym_amd/lib/variants/$v/libhgym_hip.so"; fi
  run "fb3 $v" $L HGYM_FB3=1
done
cut -c1-260 $out
