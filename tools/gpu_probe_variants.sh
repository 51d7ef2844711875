#!/bin/bash
# usage: tools/gpu_probe_variants.sh <probe.py> <variant> ...   -- one probe script under each library variant ("base" = default)
R=$GRAFT_REPO_ROOT
probe=$1; shift
lib() { if [ "$1" == "base" ]; then echo $R/humanoid-gym_amd/lib/libhgym_hip.so; else echo $R/humanoid-gym_amd/lib/variants/$1/libhgym_hip.so; fi; }
out=$R/gpurun_out/probe_$(basename $probe .py)_$(echo "$@" | tr ' ' '_').txt
: > $out
for v in "$@"; do
  echo "== $v" >> $out
  HGYM_LIB=$(lib $v) timeout 300 python $probe 2>&1 | grep "^step 2\|^step 3\|rror" >> $out
done
cat $out
