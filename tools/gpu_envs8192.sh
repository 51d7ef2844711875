#!/bin/bash
# usage: tools/gpu_envs8192.sh <tag> "<ENV=..>" ...  -- bench at 8192 envs/GPU under each environment setting (same box, two interleaved repetitions)
tag=$1; shift; O=gpurun_out; mkdir -p $O
out=$O/${tag}_ab.txt; : > $out
for rep in 1 2; do
  for v in "$@"; do
    line=$(env $v timeout 300 python bench.py --num-envs 8192 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --configs none 2>&1 | tail -1)
    echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8192 envs %-28s steps/s %.4g  ms/iter %.3f  coll %.3f  upd %.3f' % ('$v', d['value'], d['ms_per_step'], d['collection_ms'], d['ppo_update_ms']))" >> $out 2>&1 || echo "$v FAILED: $line" | cut -c1-300 >> $out
  done
done
cat $out
