#!/bin/bash
# env-step GPU tests, then same-box A/B of the isolated env step and the whole bench under HGYM_ENV_ABLATE settings
mkdir -p gpurun_out; out=gpurun_out/envab.txt; : > $out
timeout 900 python -m pytest tests/test_env_gpu.py tests/test_fused_gpu.py -m gpu -q -x 2>&1 | tail -3 >> $out
for ab in "$@"; do HGYM_ENV_ABLATE=$ab python tools/probe_scale.py env4096 2>&1 | grep ablate >> $out; done
for ab in "$@" "$@"; do
  HGYM_ENV_ABLATE=$ab timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate=$ab steps/s %.4g  ms/iter %.3f  coll %.3f  upd %.3f' % (d['value'], d['ms_per_step'], d['collection_ms'], d['ppo_update_ms']))" >> $out
done
cat $out
