#!/bin/bash
# usage: tools/gpu_shadow_ab.sh <tag>  -- shadow tests, then same-box A/B: update micro-benchmark and whole bench with / without the bf16 shadow
tag=$1
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_net_gpu.py tests/test_synth_path.py tests/test_runner_gpu.py tests/test_aux_head_gpu.py -m gpu -q -x > $O/${tag}_pytest.txt 2>&1
echo "pytest exit $?" >> $O/${tag}_pytest.txt
tail -5 $O/${tag}_pytest.txt
out=$O/${tag}_ab.txt; : > $out
for rep in 1 2; do
  for sh in 0 1; do
    echo "== bench_update shadow=$sh (rep $rep)" >> $out
    HGYM_BU_SHADOW=$sh HGYM_S=245760 timeout 200 python tools/bench_update.py 2>&1 | grep "calib\|minibatch\|mlp_\|dw \|reduce\|apply" >> $out
  done
done
for rep in 1 2; do
  for sh in 0 1; do
    line=$(HGYM_SHADOW=$sh timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --configs none 2>&1 | tail -1)
    echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench shadow=$sh  steps/s %.4g  ms/iter %.3f  coll %.3f  upd %.3f' % (d['value'], d['ms_per_step'], d['collection_ms'], d['ppo_update_ms']))" >> $out 2>&1 || echo "shadow=$sh FAILED: $line" | cut -c1-300 >> $out
  done
done
cat $out
