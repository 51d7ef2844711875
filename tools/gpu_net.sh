#!/bin/bash
# usage: tools/gpu_net.sh <tag>  -- net tests + short bench
tag=$1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_net_gpu.py tests/test_runner_gpu.py -m gpu -q -x > gpurun_out/${tag}_pytest.txt 2>&1
echo "pytest exit $?" >> gpurun_out/${tag}_pytest.txt
tail -40 gpurun_out/${tag}_pytest.txt
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench.txt 2>&1
echo "bench exit $?" >> gpurun_out/${tag}_bench.txt
tail -3 gpurun_out/${tag}_bench.txt | cut -c1-1500
