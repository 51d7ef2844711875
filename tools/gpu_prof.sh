#!/bin/bash
# usage: tools/gpu_prof.sh <tag> [bench args]  -- rocprofv3 kernel trace of a short bench run
tag=$1; shift
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $tag -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline "$@" > $R/gpurun_out/${tag}_rocprof.txt 2>&1
echo "rocprof exit $?" >> $R/gpurun_out/${tag}_rocprof.txt
cd $R
tail -2 gpurun_out/${tag}_rocprof.txt | cut -c1-400
python tools/trace_stats.py $(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1) --grid --tail-frac 0.5 | cut -c1-175 | head -45
