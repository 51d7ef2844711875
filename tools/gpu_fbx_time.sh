#!/bin/bash
# usage: tools/gpu_fbx_time.sh <tag> "<label>|<ENV=.. ENV=..>|<variant or base>" ...
# per spec: update micro-benchmark (tools/bench_update.py, B = 61 440 of 245 760 rows) + the tile's phase clock (tools/probe_phases.py), all in ONE call
tag=$1; shift
mkdir -p gpurun_out
out=gpurun_out/${tag}_fbx_time.txt
: > $out
for spec in "$@"; do
  IFS='|' read -r label envs v <<< "$spec"
  if [ -z "$v" ] || [ "$v" = base ]; then L=""; else L="HGYM_LIB=$PWD/humanoid-gym_amd/lib/variants/$v/libhgym_hip.so"; fi
  echo "== $label" >> $out
  env $L $envs HGYM_S=245760 timeout 300 python tools/bench_update.py 2>&1 | grep "mlp_fwd\|dw  \|minibatch" >> $out
  env $L $envs timeout 300 python tools/probe_phases.py 2>&1 | grep "mlp_fb<" >> $out
done
cut -c1-260 $out
