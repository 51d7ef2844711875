#!/bin/bash
# same-call A/B of the per-env chain on four wavefronts (default) against one (variant w1 = build.py --variant w1 -DHGYM_ENV_WAVES3=0;
# probe variants w3p / w3pe / w1p: -DHGYM_W3_PROBE=1, + -DHGYM_RO_AHEAD_LATE=0, + -DHGYM_ENV_WAVES3=0):
# whole bench twice each, the rollout launch's phase clock under both, then the GPU test suite under the default library
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
for v in w3p w3pe w1p; do echo "== $v"; HGYM_LIB=$R/humanoid-gym_amd/lib/variants/$v/libhgym_hip.so timeout 250 python tools/probe_w3.py 2>&1 | grep "^step"; done > $O/w3_probe.txt 2>&1
cat $O/w3_probe.txt
rm -f $O/w3_phase.txt
bash tools/gpu_bench_ab.sh base w1
for v in base w1; do
  if [ "$v" == "base" ]; then L=$R/humanoid-gym_amd/lib/libhgym_hip.so; else L=$R/humanoid-gym_amd/lib/variants/$v/libhgym_hip.so; fi
  echo "== phase clock, $v" >> $O/w3_phase.txt
  HGYM_LIB=$L timeout 300 python tools/probe_rollout.py 2>&1 | grep "^step 3" >> $O/w3_phase.txt
done
cat $O/w3_phase.txt
HGYM_LIB=$R/humanoid-gym_amd/lib/variants/w1/libhgym_hip.so timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-pmc --configs envs8192 --steps 10 2>&1 | tail -1 > $O/w3_cfg_w1.json
timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-pmc --configs envs8192 --steps 10 2>&1 | tail -1 > $O/w3_cfg_base.json
python - <<'P'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
for v in ("w1", "base"):
    try:
        d = json.load(open("%s/w3_cfg_%s.json" % (O, v)))
        c = [x for x in d.get("configs", []) if x.get("name") == "envs8192"][0]
        print(v, "envs8192: %.2f M env-steps/s, collection %.3f ms" % (c["value"] / 1e6, c["collection_ms"]))
    except Exception as e:
        print(v, "failed", e)
P
timeout 900 python -m pytest tests -m gpu -q -x > $O/w3_pytest.txt 2>&1; tail -3 $O/w3_pytest.txt
