#!/bin/bash
# usage: tools/gpu_pytest.sh <tag> [pytest args...]
tag=$1; shift
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q "$@" > gpurun_out/${tag}_pytest.txt 2>&1
echo "pytest exit $?" >> gpurun_out/${tag}_pytest.txt
tail -60 gpurun_out/${tag}_pytest.txt
