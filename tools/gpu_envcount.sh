#!/bin/bash
# bench.py at several env counts per GPU (BASELINE configs[1] = 4096, configs[3] = 8192, beyond: throughput regime)
mkdir -p gpurun_out; out=gpurun_out/scale_envs.txt; : > $out
for n in "$@"; do python bench.py --num-envs $n --steps 6 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; ks={k['kernel']:k for k in r['kernels']}; ks[r['kernel']]=r
e=ks['env_step_kernel']; f=ks['mlp_fwd_kernel']; p=ks['mlp_fwd_kernel<32>']
print('N=%6d  %.4g env-steps/s  %.2f ms/iter  coll %.2f  upd %.2f | env_step %.1f us %.0f GB/s (%.1f%% of HBM peak) | policy step %.1f us | fwd64 %.0f us %.0f TF/s' % (d['config']['envs_per_gpu'], d['value'], d['ms_per_step'], d['collection_ms'], d['ppo_update_ms'], e['avg_launch_us'], e['achieved'], 100*e['frac'], p['avg_launch_us'], f['avg_launch_us'], f['achieved']))
" >> $out 2>&1; done
cat $out
