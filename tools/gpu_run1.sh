#!/bin/bash
# first GPU visit: env + GAE parity tests, then a timing probe
mkdir -p gpurun_out
rocminfo | grep -E "gfx|Compute Unit" | head -4 > gpurun_out/run1_dev.txt 2>&1
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/run1_pytest.txt 2>&1
echo "pytest exit $?" >> gpurun_out/run1_pytest.txt
timeout 300 python tools/probe_env.py > gpurun_out/run1_probe.txt 2>&1
echo "probe exit $?" >> gpurun_out/run1_probe.txt
tail -5 gpurun_out/run1_pytest.txt; cat gpurun_out/run1_probe.txt
