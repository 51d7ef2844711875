#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/run6_gemm.txt; : > $out
echo "== env probe" >> $out
timeout 120 python tools/probe_env.py 2>&1 | grep -E "env_step_synth" >> $out
for ab in 0 1 2 4 8 3 7 15; do
  echo "== HGYM_GEMM_ABLATE=$ab" >> $out
  HGYM_GEMM_ABLATE=$ab timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps/s %.3g update_ms %.2f coll_ms %.2f gemm_TF %.1f gemm_avg_us %.2f gemm_share %.2f' % (d['value'], d['ppo_update_ms'], d['collection_ms'], d['roofline']['achieved'] if d['roofline']['kernel']=='gemm_nt_kernel' else d['roofline']['second']['achieved'], d['roofline']['avg_launch_us'] if d['roofline']['kernel']=='gemm_nt_kernel' else d['roofline']['second']['avg_us'], d['roofline']['share_of_iteration']))" >> $out 2>&1
done
cat $out
