#!/bin/bash
# round 5: the previous step's finaliser on the first actor workgroup to finish (library variant fintk) -- tests, then bench A/B (headline, logging, 8192 envs)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
V=$R/humanoid-gym_amd/lib/variants/fintk/libhgym_hip.so
HGYM_LIB=$V timeout 900 python -m pytest tests/test_synth_path.py tests/test_fused_gpu.py tests/test_runner_gpu.py tests/test_env_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/r05g_fintk.txt
bash tools/gpu_bench_ab.sh base fintk | tee -a $O/r05g_fintk.txt
HGYM_AB_BENCH_ARGS="--num-envs 8192" bash tools/gpu_bench_ab.sh base fintk | tee -a $O/r05g_fintk.txt
for rep in 1 2; do for v in base fintk; do
  if [ $v == base ]; then L=$R/humanoid-gym_amd/lib/libhgym_hip.so; else L=$V; fi
  HGYM_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-roofline --configs logging --steps 6 --warmup 2 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
c=[x for x in d['configs'] if x['name']=='logging_on'][0]
print('logging $v rep $rep:', {k:c.get(k) for k in ('value','value_without_checkpoints','collection_ms','ppo_update_ms')})" | tee -a $O/r05g_fintk.txt
done; done
