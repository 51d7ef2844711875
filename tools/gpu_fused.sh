#!/bin/bash
# usage: tools/gpu_fused.sh [VAR=val ...] -- fused-kernel GPU tests, phase clock, update microbench
mkdir -p gpurun_out; out=gpurun_out/fused.txt; : > $out
timeout 600 python -m pytest tests/test_fused_gpu.py tests/test_net_gpu.py -m gpu -q -x 2>&1 | tail -4 >> $out
env "$@" timeout 300 python tools/probe_phases.py 2>&1 | grep -E "fwd|bwd|policy" >> $out
env "$@" timeout 300 python tools/bench_update.py 2>&1 | grep -E "calib|minibatch|mlp_fwd|mlp_bwd|dw |policy" >> $out
cat $out
