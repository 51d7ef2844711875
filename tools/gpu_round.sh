#!/bin/bash
# usage: tools/gpu_round.sh <tag>  -- GPU tests + smoke + default bench + rocprofv3 kernel-trace stats of the same bench command
tag=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/${tag}_pytest.txt 2>&1
echo "pytest exit $?" >> gpurun_out/${tag}_pytest.txt
tail -6 gpurun_out/${tag}_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.txt 2>&1
echo "smoke exit $?" >> gpurun_out/${tag}_smoke.txt
tail -3 gpurun_out/${tag}_smoke.txt
timeout 600 python bench.py > gpurun_out/${tag}_bench.txt 2>&1
echo "bench exit $?" >> gpurun_out/${tag}_bench.txt
tail -2 gpurun_out/${tag}_bench.txt | cut -c1-1500
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $tag -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --configs none > $R/gpurun_out/${tag}_rocprof.txt 2>&1
echo "rocprof exit $?" >> $R/gpurun_out/${tag}_rocprof.txt
cd $R
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
echo "stats file: $f"; cut -c1-200 "$f" | head -30
python tools/trace_stats.py $(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1) --grid --tail-frac 0.5 > gpurun_out/${tag}_trace_stats.txt 2>&1
find gpurun_out/prof_$tag -name "*kernel_trace.csv" -size +20M -delete
