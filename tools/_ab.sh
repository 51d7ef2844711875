R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; out=$O/envfast_ab.txt; : > $out
timeout 300 python -m pytest tests/test_env_gpu.py -x -q -m gpu 2>&1 | tail -2 >> $out
for rep in 1 2; do for v in envslow base; do
  if [ $v == base ]; then lib=$R/humanoid-gym_amd/lib/libhgym_hip.so; else lib=$R/humanoid-gym_amd/lib/variants/$v/libhgym_hip.so; fi
  echo "== $v" >> $out
  HGYM_LIB=$lib timeout 60 python tools/probe_scale.py env4096 2>&1 | grep ablate >> $out
  HGYM_LIB=$lib HGYM_ENV_ABLATE=8 timeout 60 python tools/probe_scale.py env4096 2>&1 | grep ablate >> $out
done; done
cat $out
bash tools/gpu_bench_ab.sh envslow base
