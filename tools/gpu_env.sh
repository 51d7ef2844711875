#!/bin/bash
# usage: tools/gpu_env.sh <tag> -- env tests + env probe + bench
tag=$1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_env_gpu.py tests/test_runner_gpu.py -m gpu -q -x > gpurun_out/${tag}_pytest.txt 2>&1
echo "pytest exit $?" >> gpurun_out/${tag}_pytest.txt
tail -15 gpurun_out/${tag}_pytest.txt | grep -v Warning
python tools/probe_env.py 2>&1 | grep -E "env_step|gae"
for ab in 2 8 10; do echo "== ablate $ab"; HGYM_ENV_ABLATE=$ab python tools/probe_env.py 2>&1 | grep "env_step_synth N=4096"; done
python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps/s %.3g ms %.2f coll %.2f upd %.2f' % (d['value'], d['ms_per_step'], d['collection_ms'], d['ppo_update_ms']))"
