#!/bin/bash
# usage: tools/gpu_pmc.sh <tag> <counters...>   -- one PMC pass over tools/bench_update.py (kernel-trace + counters only)
tag=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
if [ "$1" == "list" ]; then rocprofv3 -L 2>&1 | grep -o -E "\b(SQ_[A-Z_0-9]+|TCC_[A-Z_0-9]+|TCP_[A-Z_0-9]+|FETCH_SIZE|WRITE_SIZE|MfmaUtil|VALUBusy|GRBM_[A-Z_]+|LDSBankConflict|OccupancyPercent|MemUnitStalled)\b" | sort -u | tr '\n' ' ' > $R/gpurun_out/${tag}_counters.txt; wc -w $R/gpurun_out/${tag}_counters.txt; exit 0; fi
HGYM_BENCH_REPS=3 timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$tag -o $tag -- python $R/tools/bench_update.py > $R/gpurun_out/${tag}_pmc.txt 2>&1
echo "exit $?" >> $R/gpurun_out/${tag}_pmc.txt
f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
python $R/tools/pmc_stats.py "$f" | tee $R/gpurun_out/${tag}_pmc_summary.txt
