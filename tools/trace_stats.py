"""Summarise a rocprofv3 kernel_trace.csv: per (kernel, grid) calls / total / avg / min / max, total busy time,
and the idle gaps between consecutive kernels (launch-bound check).

    python tools/trace_stats.py gpurun_out/prof7/r7_kernel_trace.csv [--grid] [--tail-frac 0.5]
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN4hgym\d+([a-z_0-9]+?)I(.*?)EEv", name)
    if m:
        args = m.group(2).replace("DF16b", "bf16").replace("Li", "").replace("E", ",").strip(",")
        return "%s<%s>" % (m.group(1), args)
    m = re.match(r"_ZN4hgym\d+([a-z_0-9]+?)E", name)
    if m:
        return m.group(1)
    name = re.sub(r"\(.*", "", name)
    return name[-70:]


def main(path, by_grid, tail_frac):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]),
                         int(r["Workgroup_Size_X"]), int(r["VGPR_Count"]), int(r["Accum_VGPR_Count"]), int(r["LDS_Block_Size"])))
    rows.sort()
    if tail_frac < 1.0:
        rows = rows[int(len(rows) * (1.0 - tail_frac)):]
    agg = defaultdict(lambda: [0, 0, 1 << 62, 0, None])
    for s, e, n, gx, gy, gz, wx, vg, ag, lds in rows:
        k = short(n)
        if by_grid:
            k += " g=%dx%dx%d" % (gx // max(wx, 1), gy, gz)
        a = agg[k]
        a[0] += 1
        a[1] += e - s
        a[2] = min(a[2], e - s)
        a[3] = max(a[3], e - s)
        a[4] = (vg, ag, lds)
    busy = sum(a[1] for a in agg.values())
    span = rows[-1][1] - rows[0][0]
    print("%-78s %7s %11s %9s %9s %9s %6s  %s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%", "vgpr/agpr/lds"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-78s %7d %11.1f %9.2f %9.2f %9.2f %6.2f  %s" % (k[:78], a[0], a[1] / 1e3, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3,
                                                              100.0 * a[1] / busy, "%d/%d/%d" % a[4]))
    gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
    pos = [g for g in gaps if g > 0]
    print("kernels %d  busy %.1f us  span %.1f us  busy/span %.3f  sum(+gaps) %.1f us  median gap %.2f us" %
          (len(rows), busy / 1e3, span / 1e3, busy / span, sum(pos) / 1e3, sorted(gaps)[len(gaps) // 2] / 1e3))


if __name__ == "__main__":
    tf = 1.0
    if "--tail-frac" in sys.argv:
        tf = float(sys.argv[sys.argv.index("--tail-frac") + 1])
    main(sys.argv[1], "--grid" in sys.argv, tf)
