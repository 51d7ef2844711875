#!/bin/bash
# usage: tools/gpu_icache.sh <tag>  -- one rocprofv3 --pmc pass (counters + kernel-trace only) with the instruction-cache / instruction counters
# over tools/traffic_run.py; prints per-kernel means.  The fused rollout launch runs ~120 KB of code once per workgroup.
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift; timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/ic_${tag}_$name -o t -- python $R/tools/traffic_run.py > $O/${tag}_ic_$name.txt 2>&1; echo "exit $?" >> $O/${tag}_ic_$name.txt; tail -1 $O/${tag}_ic_$name.txt; }
pass A SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE GRBM_GUI_ACTIVE
pass B SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
python - $tag <<'P' | tee $O/${sys_tag:-$tag}_icache.txt
import csv, glob, sys
from collections import defaultdict
tag = sys.argv[1]
for name in "AB":
    fs = glob.glob("/tmp/ic_%s_%s/**/*counter_collection.csv" % (tag, name), recursive=True)
    if not fs:
        print("pass", name, ": no counter file"); continue
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][-60:]
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, cs in sorted(agg.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
        if not any(x in k for x in ("rollout_step", "mlp_fb", "dw_kernel", "env_step")): continue
        print("%-62s" % k, "  ".join("%s=%.4g" % (c, s / n) for c, (s, n) in sorted(cs.items())), " launches", max(v[1] for v in cs.values()))
P
