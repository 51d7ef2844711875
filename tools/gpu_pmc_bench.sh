#!/bin/bash
# usage: tools/gpu_pmc_bench.sh <tag> <counters...>  -- one PMC pass (kernel-trace + counters only) over a short bench.py run (in-situ kernels)
export HGYM_BENCH_PMC=0   # (bench.py collects PMC traffic itself by default: not under another profiler)
tag=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmcb_$tag -o $tag -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/${tag}_pmcb.txt 2>&1
echo "exit $?" >> $R/gpurun_out/${tag}_pmcb.txt
tail -2 $R/gpurun_out/${tag}_pmcb.txt | cut -c1-300
f=$(find /tmp/pmcb_$tag -name "*counter_collection.csv" | head -1)
python $R/tools/pmc_stats.py "$f" | tee $R/gpurun_out/${tag}_pmcb_summary.txt
