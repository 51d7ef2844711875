#!/bin/bash
mkdir -p gpurun_out; out=gpurun_out/scale.txt; : > $out
python tools/probe_scale.py all >> $out 2>&1
for ab in 1 2 4 8 64 128 130 194 195 203 207; do HGYM_ENV_ABLATE=$ab python tools/probe_scale.py env4096 2>&1 | grep ablate >> $out; done
cat $out
