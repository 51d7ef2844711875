#!/bin/bash
# usage: tools/gpu_update_ab.sh <tag> [pytest -k expr]  -- fused/net GPU tests, then the update micro-benchmark with / without the bf16 shadow (same box)
tag=$1; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_net_gpu.py tests/test_aux_head_gpu.py -m gpu -q -x ${2:+-k "$2"} > $O/${tag}_pytest.txt 2>&1
echo "pytest exit $?" >> $O/${tag}_pytest.txt; tail -3 $O/${tag}_pytest.txt
out=$O/${tag}_ab.txt; : > $out
for rep in 1 2; do
  for sh in 0 1; do
    echo "== bench_update shadow=$sh (rep $rep)" >> $out
    HGYM_BU_SHADOW=$sh HGYM_S=245760 timeout 200 python tools/bench_update.py 2>&1 | grep "calib\|minibatch\|mlp_\|dw \|reduce\|apply" >> $out
  done
done
cat $out
