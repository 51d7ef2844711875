#!/bin/bash
# round 5, closing evidence: the same 600 learning iterations (4096 envs, same seed) at the headline's bf16 and at the reference's fp32
mkdir -p gpurun_out
for p in bf16 f32; do
  HGYM_PRECISION=$p HGYM_SANITY_EVERY=50 timeout 900 python tools/train_sanity.py 600 > gpurun_out/r05_train_600_$p.txt 2>&1
  echo "exit $?" >> gpurun_out/r05_train_600_$p.txt
  tail -4 gpurun_out/r05_train_600_$p.txt | cut -c1-220
done
