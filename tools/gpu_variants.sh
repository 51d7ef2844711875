#!/bin/bash
# usage: tools/gpu_variants.sh <variant> ...   -- same-box A/B of library variants built by `build.py --variant <name> <flags>`
# ("base" = the default library): the update micro-benchmark for each, twice, interleaved; then the fused / net GPU tests
# under every non-base variant.  HGYM_AB_BENCH=1 adds the whole bench.py per variant.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
lib() { if [ "$1" == "base" ]; then echo $R/humanoid-gym_amd/lib/libhgym_hip.so; else echo $R/humanoid-gym_amd/lib/variants/$1/libhgym_hip.so; fi; }
out=$O/variants_$(echo "$@" | tr ' ' '_').txt
: > $out
for rep in 1 2; do
  for v in "$@"; do
    echo "== $v (rep $rep)" >> $out
    HGYM_LIB=$(lib $v) HGYM_S=245760 timeout 200 python tools/bench_update.py 2>&1 | grep "calib\|minibatch\|mlp_\|dw \|loss" >> $out
  done
done
if [ -n "$HGYM_AB_BENCH" ]; then
  for v in "$@"; do
    echo "== bench $v" >> $out
    HGYM_LIB=$(lib $v) timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-400 >> $out
  done
fi
for v in "$@"; do
  [ "$v" == "base" ] && continue
  echo "== tests $v" >> $out
  HGYM_LIB=$(lib $v) timeout 300 python -m pytest tests/test_fused_gpu.py tests/test_net_gpu.py -x -q -m gpu 2>&1 | tail -2 >> $out
done
cat $out
