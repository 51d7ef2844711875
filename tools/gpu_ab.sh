#!/bin/bash
# usage: tools/gpu_ab.sh "<VAR=val ...>" "<VAR=val ...>" ...  -- same-box A/B of bench.py (no cpu baseline / roofline) under each env setting
mkdir -p gpurun_out; out=gpurun_out/ab.txt; : > $out
for round in 1 2; do
for v in "$@"; do
  line=$(env $v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --configs none 2>&1 | tail -1)
  echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s steps/s %.4g  ms/iter %.3f  coll %.3f  upd %.3f' % ('$v', d['value'], d['ms_per_step'], d['collection_ms'], d['ppo_update_ms']))" >> $out 2>&1 || echo "$v FAILED: $line" | cut -c1-300 >> $out
done
done
cat $out
