#!/bin/bash
# usage: tools/gpu_variants.sh  -- fused-kernel GPU tests, then tools/bench_update.py under each tuning override
mkdir -p gpurun_out; out=gpurun_out/variants.txt; : > $out
timeout 600 python -m pytest tests/test_fused_gpu.py tests/test_net_gpu.py -m gpu -q -x 2>&1 | tail -5 >> $out
for v in "" "HGYM_RING=2" "HGYM_FWD_WAVES=8" "HGYM_FWD_TILE=32"; do
  echo "== $v" >> $out
  env $v timeout 300 python tools/bench_update.py 2>&1 | grep -E "calib|minibatch|mlp_fwd|mlp_bwd|dw |policy" >> $out
done
cat $out
