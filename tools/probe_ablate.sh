#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/run5_ablate.txt; : > $out
for ab in 0 2 8 10 16 32 48 1 4; do
  echo "== HGYM_ENV_ABLATE=$ab" >> $out
  HGYM_ENV_ABLATE=$ab timeout 120 python tools/probe_env.py 2>&1 | grep -E "env_step_synth N=4096|env_step_synth N=32768" >> $out
done
for big in 0 1 2; do
  echo "== HGYM_GEMM_BIG=$big" >> $out
  HGYM_GEMM_BIG=$big timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ppo_update_ms'], d['collection_ms'], d['roofline']['achieved'], d['roofline']['avg_launch_us'])" >> $out
done
cat $out
