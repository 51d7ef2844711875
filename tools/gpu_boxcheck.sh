#!/bin/bash
# Which latency class is this box?  shader clock under 1 / 256 / 2048 busy workgroups, env / policy step latency, rocm-smi clocks
mkdir -p gpurun_out; out=gpurun_out/boxcheck.txt; : > $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probes/probe_clock.hip -o /tmp/probe_clock 2>/dev/null && /tmp/probe_clock | tail -3 >> $out
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk|socclk" | head -6 >> $out
rocm-smi --showperflevel --showpower 2>/dev/null | grep -E "Performance Level|Power" | head -4 >> $out
HGYM_ENV_ABLATE=0 python tools/probe_scale.py env4096 2>&1 | grep ablate >> $out
python tools/bench_update.py 2>&1 | grep -E "calib|policy_act M=4096" >> $out
cat $out
{ rocm-smi --showmemorypartition --showcomputepartition 2>/dev/null | grep -iE "partition" | head -4; cat /sys/module/amdgpu/parameters/vm_fragment_size 2>/dev/null; } >> $out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench: %.4g env-steps/s  ms/iter %.3f  coll %.3f  upd %.3f' % (d['value'], d['ms_per_step'], d['collection_ms'], d['ppo_update_ms']))" >> $out
tail -4 $out
