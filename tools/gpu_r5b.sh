#!/bin/bash
# usage: tools/gpu_r5b.sh <tag>  -- the log sink's book-keeping in one round trip (base) vs the row-by-row loop (variant logold): the runner tests under
# base, then the bench with its logging_on configuration per library, interleaved
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
lib() { if [ "$1" == "base" ]; then echo $R/humanoid-gym_amd/lib/libhgym_hip.so; else echo $R/humanoid-gym_amd/lib/variants/$1/libhgym_hip.so; fi; }
timeout 900 python -m pytest tests/test_runner_gpu.py tests/test_scripts_gpu.py tests/test_fused_gpu.py -m gpu -q -x > $O/${tag}_pytest.txt 2>&1; echo "pytest exit $?" >> $O/${tag}_pytest.txt; tail -3 $O/${tag}_pytest.txt
out=$O/${tag}_logging_ab.txt; : > $out
for rep in 1 2; do
  for v in base logold; do
    HGYM_LIB=$(lib $v) timeout 400 python bench.py --no-cpu-baseline --no-roofline --no-pmc --configs logging --steps 20 2>/dev/null | tail -1 > $O/_line.json
    python - "$v" "$rep" $O/_line.json >> $out <<'P'
import json, sys
d = json.load(open(sys.argv[3]))
c = d["configs"][0]
print("%-7s rep %s: headline %.2f M (collection %.3f update %.3f) | logging_on %.2f M, without checkpoints %.2f M (collection %.3f update %.3f, %.1f ms of checkpoints)"
      % (sys.argv[1], sys.argv[2], d["value"] / 1e6, d["collection_ms"], d["ppo_update_ms"], c["value"] / 1e6, c["value_without_checkpoints"] / 1e6,
         c["collection_ms"], c["ppo_update_ms"], c["checkpoint_ms_total"]))
P
  done
done
cat $out
