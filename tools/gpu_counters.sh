#!/bin/bash
# usage: tools/gpu_counters.sh <tag>  -- four rocprofv3 --pmc passes (counters + kernel-trace only) over tools/traffic_run.py (a calibration copy +
# 4 bench iterations at 4096 envs): FETCH_SIZE, WRITE_SIZE -> HBM bytes per launch; an SQ pass and a TCC pass -> MFMA occupancy, wave states,
# L2 hit rate.  Leaves gpurun_out/<tag>_pmc_traffic.json and gpurun_out/<tag>_sq_tcc_counters.txt.
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/ctr_${tag}_$name -o t -- python $R/tools/traffic_run.py > $O/${tag}_pass_$name.txt 2>&1; echo "exit $?" >> $O/${tag}_pass_$name.txt; tail -1 $O/${tag}_pass_$name.txt; }
pass FETCH FETCH_SIZE
pass WRITE WRITE_SIZE
pass SQ SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass TCC TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE
f() { find /tmp/ctr_${tag}_$1 -name "*counter_collection.csv" | head -1; }
python $R/tools/pmc_to_json.py $(f FETCH) $(f WRITE) $O/${tag}_pmc_traffic.json 268435456
python $R/tools/pmc_counters.py $(f SQ) $(f TCC) $O/${tag}_pmc_traffic.json $O/${tag}_sq_tcc_counters.txt
