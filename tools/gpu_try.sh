#!/bin/bash
# usage: tools/gpu_try.sh <tag> "<pytest selection>" ["<bench args>"]  -- a selected set of GPU tests, then (optionally) one bench.py run
tag=$1; sel=$2; bargs=$3
mkdir -p gpurun_out
eval timeout 1500 python -m pytest $sel -m gpu -q -x > gpurun_out/${tag}_pytest.txt 2>&1
echo "pytest exit $?" >> gpurun_out/${tag}_pytest.txt
tail -15 gpurun_out/${tag}_pytest.txt
if [ -n "$bargs" ]; then
  timeout 900 python bench.py $bargs > gpurun_out/${tag}_bench.txt 2> gpurun_out/${tag}_bench.err
  echo "bench exit $?" >> gpurun_out/${tag}_bench.txt
  tail -3 gpurun_out/${tag}_bench.txt | cut -c1-6000
  tail -5 gpurun_out/${tag}_bench.err
fi
