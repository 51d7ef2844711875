#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/run7_pytest.txt 2>&1
echo "pytest exit $?" >> gpurun_out/run7_pytest.txt
tail -25 gpurun_out/run7_pytest.txt
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/run7_bench.txt 2>&1
echo "bench exit $?" >> gpurun_out/run7_bench.txt
tail -5 gpurun_out/run7_bench.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof7 -o r7 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/run7_rocprof.txt 2>&1
echo "rocprof exit $?" >> $GRAFT_REPO_ROOT/gpurun_out/run7_rocprof.txt
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof7 | head -20
f=$(find gpurun_out/prof7 -name "*kernel_stats.csv" | head -1)
echo "stats file: $f"; head -25 "$f"
find gpurun_out/prof7 -name "*kernel_trace.csv" -size +20M -delete
