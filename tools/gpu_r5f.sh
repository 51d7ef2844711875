#!/bin/bash
# round 5: adam_kernel with four parameters per lane (library variant adam4) -- bit identity of the training state, then timing, same call
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
V=$R/humanoid-gym_amd/lib/variants/adam4/libhgym_hip.so
: > $O/r05f_adam4.txt
for lib in base adam4; do
  for cfg in "3 1024" "2 4096"; do
    if [ $lib == base ]; then timeout 300 python tools/param_digest.py $cfg 2>&1 | grep "^digest" >> $O/r05f_adam4.txt
    else HGYM_LIB=$V timeout 300 python tools/param_digest.py $cfg 2>&1 | grep "^digest" >> $O/r05f_adam4.txt; fi
  done
done
cat $O/r05f_adam4.txt
HGYM_LIB=$V timeout 600 python -m pytest tests/test_fused_gpu.py tests/test_net_gpu.py tests/test_aux_head_gpu.py -x -q -m gpu 2>&1 | tail -2 | tee -a $O/r05f_adam4.txt
bash tools/gpu_variants.sh base adam4 2>&1 | grep -v "^==.*tests\|passed" | tee -a $O/r05f_adam4.txt | tail -30
bash tools/gpu_bench_ab.sh base adam4 | tee -a $O/r05f_adam4.txt
