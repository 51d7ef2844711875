"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel launches / total / avg / min / max, like --stats."""
import re
import sqlite3
import sys


def main(path, top=40, by_grid=False):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "info_kernel_symbol" in t][0]
    key = "s.kernel_name" + (", d.grid_size_x, d.grid_size_y, d.grid_size_z" if by_grid else "")
    rows = cur.execute(f"select {key}, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                       f"from {kd} d join {ks} s on d.kernel_id = s.id group by {key} order by 3 desc").fetchall()
    tot = sum(r[-4] for r in rows)
    print("%-86s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for r in rows[:top]:
        name = r[0]
        name = re.sub(r"\(.*", "", name)
        if by_grid:
            name = "%s grid=%dx%dx%d" % (name[:60], r[1], r[2], r[3])
        n, t, a, mn, mx = r[-5:]
        print("%-86s %8d %12.1f %10.2f %10.2f %10.2f %6.2f" % (name[:86], n, t / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
    print("total kernel time %.1f us over %d kernels" % (tot / 1e3, sum(r[-5] for r in rows)))


if __name__ == "__main__":
    main(sys.argv[1], by_grid="--grid" in sys.argv)
