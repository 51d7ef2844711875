#!/bin/bash
# usage: tools/gpu_apply_ab.sh <tag> <variant>... -- kernel durations of the minibatch's small kernels (rocprofv3 kernel-trace stats over tools/bench_update.py)
tag=$1; shift; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; out=$O/${tag}_apply.txt; : > $out
lib() { if [ "$1" == "base" ]; then echo $R/humanoid-gym_amd/lib/libhgym_hip.so; else echo $R/humanoid-gym_amd/lib/variants/$1/libhgym_hip.so; fi; }
cd /tmp && export TMPDIR=/tmp
for v in "$@" "$@"; do
  echo "== $v" >> $out
  rm -rf /tmp/ap_$v
  HGYM_LIB=$(lib $v) HGYM_S=245760 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ap_$v -o a -- python $R/tools/bench_update.py > /tmp/ap_$v.log 2>&1
  f=$(find /tmp/ap_$v -name "*kernel_stats.csv" | head -1)
  python -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('apply_prologue','adam_kernel','reduce_slabs','dw_kernel','mlp_fb')):
        print('  %-34s calls %4s avg %9.1f ns' % (n.replace('void ','').replace('hgym::','')[:34], r['Calls'], float(r['AverageNs'])))
" "$f" >> $out
done
cat $out
