#!/bin/bash
# usage: tools/gpu_r5d.sh <tag>  -- deferred values: the new / changed GPU tests; the whole bench at 8192 envs with the critic
# deferred (the new default there) against the two-launch path (HGYM_ROLLOUT_CRITIC=inline), and at 4096 envs inline (default) against deferred;
# then ten more runs of the code-object-size reproducer
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gae_gpu.py tests/test_synth_path.py tests/test_fused_gpu.py tests/test_net_gpu.py tests/test_runner_gpu.py tests/test_scripts_gpu.py -m gpu -q -x > $O/${tag}_pytest.txt 2>&1
echo "pytest exit $?" >> $O/${tag}_pytest.txt; tail -12 $O/${tag}_pytest.txt
out=$O/${tag}_deferred_ab.txt; : > $out
line() { python - "$1" "$2" $O/_line.json >> $out <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[3]))
    print("%-26s rep %s: %.2f M env-steps/s  %.3f ms/iter  collection %.3f  update %.3f" % (sys.argv[1], sys.argv[2], d["value"] / 1e6, d["ms_per_step"], d["collection_ms"], d["ppo_update_ms"]))
except Exception as e:
    print(sys.argv[1], "failed:", e, open(sys.argv[3]).read()[-300:])
P
}
for rep in 1 2; do
  for mode in auto inline; do
    HGYM_ROLLOUT_CRITIC=$mode timeout 300 python bench.py --no-cpu-baseline --no-roofline --configs none --steps 20 --num-envs 8192 2>&1 | tail -1 > $O/_line.json; line "8192 envs critic=$mode" $rep
  done
done
for rep in 1 2; do
  for mode in auto deferred; do
    HGYM_ROLLOUT_CRITIC=$mode timeout 300 python bench.py --no-cpu-baseline --no-roofline --configs none --steps 20 2>&1 | tail -1 > $O/_line.json; line "4096 envs critic=$mode" $rep
  done
done
cat $out
lib() { if [ "$1" == "base" ]; then echo $R/humanoid-gym_amd/lib/libhgym_hip.so; else echo $R/humanoid-gym_amd/lib/variants/$1/libhgym_hip.so; fi; }
out=$O/${tag}_code_object_repro.txt; : > $out
for rep in 1 2 3 4 5 6 7 8 9 10; do
    HGYM_LIB=$(lib pad1m) timeout 300 python -m pytest "tests/test_dist_gpu.py::test_eight_ranks_one_gpu_stay_in_lockstep" -m gpu -q -x > $O/_repro.txt 2>&1
    rc=$?
    echo "pad1m rep $rep: pytest exit $rc; ILLEGAL_INSTRUCTION lines: $(grep -c ILLEGAL_INSTRUCTION $O/_repro.txt); $(tail -1 $O/_repro.txt)" >> $out
done
cat $out
