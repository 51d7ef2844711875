#!/bin/bash
# usage: tools/gpu_prof.sh <tag> "<bench args>"  -- rocprofv3 kernel trace + stats of one short bench run, per-(kernel, grid) summary
export HGYM_BENCH_PMC=0   # (bench.py collects PMC traffic itself by default: not under another profiler)
tag=$1; bargs=$2
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $tag -- python $R/bench.py $bargs > $R/gpurun_out/${tag}_rocprof.txt 2>&1
echo "rocprof exit $?" >> $R/gpurun_out/${tag}_rocprof.txt
cd $R
tail -2 gpurun_out/${tag}_rocprof.txt | cut -c1-600
python tools/trace_stats.py $(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1) --grid --tail-frac 0.5 > gpurun_out/${tag}_trace_stats.txt 2>&1
head -14 gpurun_out/${tag}_trace_stats.txt | cut -c1-200
find gpurun_out/prof_$tag -name "*kernel_trace.csv" -size +20M -delete
