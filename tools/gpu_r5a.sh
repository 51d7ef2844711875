#!/bin/bash
# usage: tools/gpu_r5a.sh <tag>  -- round 5's first call: the whole GPU suite, then dw_kernel_rs on 32x32x16 MFMAs (variant dw32: net / fused tests under it,
# update micro-benchmark interleaved with base) and the prepared HGYM_RO_ASSUME_FAST rollout variant (whole bench, interleaved with base)
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
lib() { if [ "$1" == "base" ]; then echo $R/humanoid-gym_amd/lib/libhgym_hip.so; else echo $R/humanoid-gym_amd/lib/variants/$1/libhgym_hip.so; fi; }
timeout 1500 python -m pytest tests -m gpu -q -x > $O/${tag}_pytest.txt 2>&1; echo "pytest exit $?" >> $O/${tag}_pytest.txt; tail -12 $O/${tag}_pytest.txt
HGYM_LIB=$(lib dw32) timeout 600 python -m pytest tests/test_net_gpu.py tests/test_fused_gpu.py tests/test_aux_head_gpu.py -m gpu -q -x > $O/${tag}_dw32_pytest.txt 2>&1; echo "dw32 pytest exit $?" >> $O/${tag}_dw32_pytest.txt; tail -6 $O/${tag}_dw32_pytest.txt
out=$O/${tag}_update_ab.txt; : > $out
for rep in 1 2 3; do
  for v in base dw32; do
    echo "== $v (rep $rep)" >> $out
    HGYM_LIB=$(lib $v) HGYM_S=245760 timeout 200 python tools/bench_update.py 2>&1 | grep "calib\|minibatch\|mlp_fwd\|dw \|policy_act" >> $out
  done
done
cat $out
HGYM_LIB=$(lib rofast) timeout 600 python -m pytest tests/test_fused_gpu.py tests/test_synth_path.py tests/test_runner_gpu.py -m gpu -q -x > $O/${tag}_rofast_pytest.txt 2>&1; echo "rofast pytest exit $?" >> $O/${tag}_rofast_pytest.txt; tail -4 $O/${tag}_rofast_pytest.txt
bash tools/gpu_bench_ab.sh base rofast dw32
