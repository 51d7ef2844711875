#!/bin/bash
# usage: tools/gpu_fb3.sh <tag> [reps]  -- mlp_fb3_kernel vs mlp_fb_kernel: equality test, then the update micro-benchmark (tools/bench_update.py,
# B = 61 440 of 245 760 rows) interleaved HGYM_FB3=0 / 1 in ONE call, then the whole bench both ways
tag=$1; reps=${2:-2}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_fused_gpu.py -m gpu -q -x -k "role_specialised" > gpurun_out/${tag}_pytest.txt 2>&1
echo "pytest exit $?" >> gpurun_out/${tag}_pytest.txt
tail -5 gpurun_out/${tag}_pytest.txt | cut -c1-300
for r in $(seq $reps); do
  for v in 0 1; do
    echo "== HGYM_FB3=$v (rep $r)" >> gpurun_out/${tag}_update.txt
    HGYM_FB3=$v HGYM_S=245760 timeout 300 python tools/bench_update.py >> gpurun_out/${tag}_update.txt 2>&1
  done
done
grep -n "==\|mlp_fwd\|dw  \|minibatch" gpurun_out/${tag}_update.txt | cut -c1-160
if [ -n "$HGYM_FB3_BENCH" ]; then
  for v in 0 1; do
    HGYM_FB3=$v timeout 600 python bench.py --no-cpu-baseline --no-pmc --configs none > gpurun_out/${tag}_bench_fb3_$v.txt 2>&1
    python - <<PY
import json
l=[x for x in open("gpurun_out/${tag}_bench_fb3_$v.txt") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("HGYM_FB3=$v", round(d["value"]/1e6,2), "M", d["ms_per_step"], "collection", d.get("collection_ms"), "update", d.get("ppo_update_ms"))
else:
    print("HGYM_FB3=$v: no bench line"); print(open("gpurun_out/${tag}_bench_fb3_$v.txt").read()[-1500:])
PY
  done
fi
