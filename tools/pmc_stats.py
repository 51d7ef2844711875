"""Average rocprofv3 counter_collection.csv values per kernel."""
import csv, sys, re
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in agg.items():
    if not any(x in k for x in ("mlp_", "dw_kernel", "env_step", "ppo_loss", "adam", "reduce")):
        continue
    print(k)
    for c, (s, n) in sorted(cs.items()):
        print("   %-32s avg %.4g  (n=%d)" % (c, s / n, n))
