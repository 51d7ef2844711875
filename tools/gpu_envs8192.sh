#!/bin/bash
# usage: tools/gpu_envs8192.sh <tag>  -- fused-rollout parity tests (both forms), then bench at 8192 envs/GPU: un-fused rollout vs the SEQ form
tag=$1; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_gpu.py -m gpu -q -x -k "rollout" > $O/${tag}_pytest.txt 2>&1
echo "pytest exit $?" >> $O/${tag}_pytest.txt; tail -4 $O/${tag}_pytest.txt
out=$O/${tag}_ab.txt; : > $out
for rep in 1 2; do
  for v in "HGYM_FUSE_ROLLOUT=0" "HGYM_FUSE_ROLLOUT=1"; do
    line=$(env $v timeout 300 python bench.py --num-envs 8192 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --configs none 2>&1 | tail -1)
    echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8192 envs %-22s steps/s %.4g  ms/iter %.3f  coll %.3f  upd %.3f' % ('$v', d['value'], d['ms_per_step'], d['collection_ms'], d['ppo_update_ms']))" >> $out 2>&1 || echo "$v FAILED: $line" | cut -c1-300 >> $out
  done
done
cat $out
