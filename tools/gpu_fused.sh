#!/bin/bash
# usage: tools/gpu_fused.sh "<VAR=val ...>" ...  -- fused-kernel GPU tests, then the update microbench under each env setting (same box)
mkdir -p gpurun_out; out=gpurun_out/fused.txt; : > $out
timeout 600 python -m pytest tests/test_fused_gpu.py tests/test_net_gpu.py -m gpu -q -x 2>&1 | tail -4 >> $out
for v in "$@"; do
  echo "== $v" >> $out
  env $v timeout 300 python tools/bench_update.py 2>&1 | grep -E "calib|minibatch|mlp_fwd|mlp_bwd|dw |loss|apply|reduce|policy_act M=4096" >> $out
done
cat $out
