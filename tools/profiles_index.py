"""Regenerates profiles/INDEX.md: one line per file under profiles/ -- what it is and the claim it supports (the file's own first line for
notes and patches, the kind of summary for rocprofv3 / bench outputs).  python tools/profiles_index.py"""
import json, os, re
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(R, "profiles")
KIND = [(r"_kernel_stats\.csv$", "rocprofv3 --kernel-trace --stats: per-kernel calls / total / average ns of one bench.py run"),
        (r"_trace_stats_by_grid\.txt$", "the same trace split per (kernel, grid) with min / median / max and launch gaps (tools/trace_stats.py)"),
        (r"_bench\.json$", "bench.py's JSON line of that run"), (r"pmc_traffic\.json$", "HBM bytes per launch per kernel (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, calibrated) + SQ / TCC counters: what bench.py quotes as roofline.traffic when it cannot measure in the run"),
        (r"\.patch$", "source of an experiment that is NOT in the library (git apply on the commit named inside / in the note next to it)")]


def first_line(path):
    try:
        with open(path, errors="replace") as f:
            for ln in f:
                ln = ln.strip().strip("=#-").strip()
                if len(ln) > 12:
                    return ln
    except OSError:
        pass
    return ""


def describe(name):
    path = os.path.join(P, name)
    for pat, what in KIND:
        if re.search(pat, name):
            if name.endswith("_bench.json"):
                try:
                    d = json.loads(open(path).read().strip().splitlines()[-1])
                    return "%s: %.1f M env-steps/s at N = %d, %.2f ms/iteration" % (what, d["value"] / 1e6, d.get("n_gpus", 1), d["ms_per_step"])
                except Exception:      # noqa: BLE001
                    return what
            if name.endswith(".patch"):
                return what
            return what
    if name.endswith((".txt", ".md")):
        return first_line(path)[:300]
    if name.endswith(".csv"):
        return "counter / trace table (see the .txt of the same tag)"
    if name.endswith(".json"):
        return "machine-readable result of the same tag's note"
    return ""


def main():
    files = sorted(f for f in os.listdir(P) if f != "INDEX.md")
    rounds = {}
    for f in files:
        m = re.match(r"r(\d\d)", f)
        rounds.setdefault("round %d" % int(m.group(1)) if m else "not tied to a round", []).append(f)
    out = ["# profiles/ -- index", "",
           "Every number DESIGN.md, README.md or bench.py quotes comes from a file here.  One line per file: what it is / the claim it supports",
           "(notes and patches: their own first line).  Regenerate with `python tools/profiles_index.py`.  The closing measurement of a round is",
           "`rNN_final_*` / the highest letter of that round (`r06s_*`: round 6, 141 GPU tests + smoke green, 40.8 M env-steps/s).", ""]
    for rnd in sorted(rounds, key=lambda r: (r[0] != "r", r)):
        out += ["## " + rnd, "", "| file | what it shows |", "|---|---|"]
        out += ["| `%s` | %s |" % (f, describe(f).replace("|", "/")) for f in rounds[rnd]]
        out.append("")
    open(os.path.join(P, "INDEX.md"), "w").write("\n".join(out))
    print(len(files), "files indexed")


if __name__ == "__main__":
    main()
