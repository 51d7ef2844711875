#!/bin/bash
# usage: tools/gpu_bench_ab.sh <variant> ...  -- same-box A/B of the WHOLE bench (bench.py, no CPU baseline) per library variant
# ("base" = the default library; others from `build.py --variant`), two interleaved repetitions; prints value, ms/iter,
# collection and update ms.  HGYM_AB_BENCH_ARGS: extra bench.py arguments (e.g. --num-envs 8192).  HGYM_AB_TESTS=1 then runs the full GPU test suite under the default library.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
lib() { if [ "$1" == "base" ]; then echo $R/humanoid-gym_amd/lib/libhgym_hip.so; else echo $R/humanoid-gym_amd/lib/variants/$1/libhgym_hip.so; fi; }
out=$O/benchab_$(echo "$@" | tr ' ' '_').txt
: > $out
for rep in 1 2; do
  for v in "$@"; do
    HGYM_LIB=$(lib $v) timeout 300 python bench.py --no-cpu-baseline --no-roofline --configs none --steps 20 $HGYM_AB_BENCH_ARGS 2>&1 | tail -1 > $O/_line.json
    python - "$v" "$rep" $O/_line.json >> $out <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[3]))
    print("%-8s rep %s: %.2f M env-steps/s  %.3f ms/iter  collection %.3f  update %.3f" % (sys.argv[1], sys.argv[2], d["value"] / 1e6, d["ms_per_step"], d["collection_ms"], d["ppo_update_ms"]))
except Exception as e:
    print(sys.argv[1], "failed:", e, open(sys.argv[3]).read()[-300:])
P
  done
done
cat $out
if [ -n "$HGYM_AB_TESTS" ]; then
  timeout 600 python -m pytest tests -m gpu -q -x > $O/benchab_pytest.txt 2>&1; tail -2 $O/benchab_pytest.txt
fi
