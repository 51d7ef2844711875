#!/bin/bash
./tools/probes/probe_clock
python tools/probe_env.py 2>&1 | grep env_step
python tools/bench_update.py 2>&1 | grep -E "calib|minibatch|policy"
for g in 1 0; do HGYM_GRAPH=$g python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph=$g', 'steps/s %.3g ms %.2f coll %.2f upd %.2f' % (d['value'], d['ms_per_step'], d['collection_ms'], d['ppo_update_ms']))"; done
./tools/probes/probe_clock | tail -3
