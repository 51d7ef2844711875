#!/bin/bash
# stand-alone env step (hgym_env_step_synth incl. its finaliser launch, 20 steps per graph replay) at several env counts: us per
# step and the fraction of the HBM peak on the algorithmic 7898 B per env-step
mkdir -p gpurun_out
python - "$@" > gpurun_out/envscale.txt 2>&1 <<'P'
import sys, os
sys.path.insert(0, "tools"); sys.path.insert(0, "humanoid-gym_amd")
import probe_scale as ps
for n in [int(x) for x in sys.argv[1:]]:
    us = ps.env_time(n)
    print("N=%6d  env step %.1f us  %.0f GB/s  (%.3f of 8 TB/s)" % (n, us, n * 7898 / us / 1e3, n * 7898 / us / 1e3 / 8000))
P
cat gpurun_out/envscale.txt
