#!/bin/bash
R=$GRAFT_REPO_ROOT
echo "== plain"; python tools/probe_env.py 2>&1 | grep "env_step_synth N=4096"
for ab in 2 8 10 1 4; do echo "== ablate $ab"; HGYM_ENV_ABLATE=$ab python tools/probe_env.py 2>&1 | grep "env_step_synth N=4096"; done
echo "== epb32"; HGYM_ENV_EPB=32 python tools/probe_env.py 2>&1 | grep "env_step_synth N=4096"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o pe -- python $R/tools/probe_env.py > /tmp/pe.txt 2>&1
grep "env_step_synth N=4096" /tmp/pe.txt
grep -E "env_step_kernel|env_finalize" /tmp/pe/*/pe_kernel_stats.csv 2>/dev/null || find /tmp/pe -name "*kernel_stats.csv" -exec grep -E "env_step|env_final" {} \;
