"""SQ / TCC passes of rocprofv3 (counter_collection.csv) -> per-kernel MFMA-pipe occupancy, wave-state split and L2 hit rate, merged
into a pmc_traffic.json as "counters":  python tools/pmc_counters.py <sq.csv> <tcc.csv> <traffic.json> <summary.txt>

  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / XCDs x 1024 SIMDs): share of the launch's SIMD-cycles with an MFMA in the pipe
  parked / stalled / issuing = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (MI355X_MICROARCH.md, PMC slots)
  l2_hit      = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
Per kernel: the mean over all its launches in the pass (the pass runs tools/traffic_run.py = 4 bench iterations at 4096 envs)."""
import csv, json, re, sys
from collections import defaultdict

NAMES = {"rollout_step_kernel<true, true, true": "rollout_step_kernel", "rollout_step_kernelILb1ELb1ELb1": "rollout_step_kernel",
         "mlp_fb_kernel": "mlp_fb_kernel", "env_step_kernel": "env_step_kernel",
         "dw_kernel": "dw_kernel", "reduce_slabs_kernel": "reduce_slabs_kernel", "adam_kernel": "sqnorm+adam_kernel", "gae_kernel": "gae_kernel",
         "mlp_fwd_kernelILi32": "mlp_fwd_kernel<32>", "mlp_fwd_kernel<32": "mlp_fwd_kernel<32>"}
XCDS, SIMDS = 8, 1024


def load(path):
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    with open(path) as f:
        for r in csv.DictReader(f):
            nm = next((v for k, v in NAMES.items() if k in r["Kernel_Name"]), None)
            if nm is None:
                continue
            a = agg[nm][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: {c: s / n for c, (s, n) in cs.items()} for k, cs in agg.items()}


def main(sq_csv, tcc_csv, traffic_json, summary):
    sq, tcc = load(sq_csv), load(tcc_csv)
    out, lines = {}, []
    for k in sorted(set(sq) | set(tcc)):
        e = {}
        s = sq.get(k, {})
        if s.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in s:
            e["mfma_busy"] = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (s["GRBM_GUI_ACTIVE"] / XCDS * SIMDS)
        if s.get("SQ_WAVE_CYCLES"):
            for nm, c in (("parked", "SQ_WAIT_ANY"), ("issue_stalled", "SQ_WAIT_INST_ANY"), ("issuing", "SQ_ACTIVE_INST_ANY")):
                if c in s:
                    e[nm] = s[c] / s["SQ_WAVE_CYCLES"]
        t = tcc.get(k, {})
        if t.get("TCC_HIT_sum") is not None and (t.get("TCC_HIT_sum", 0) + t.get("TCC_MISS_sum", 0)) > 0:
            e["l2_hit"] = t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"])
            e["l2_requests_per_launch"] = t.get("TCC_REQ_sum", t["TCC_HIT_sum"] + t["TCC_MISS_sum"])
        e["raw"] = dict(sq=s, tcc=t)
        out[k] = e
        lines.append("%-22s mfma_busy %s  waves: parked %s, issue-stalled %s, issuing %s  |  L2 hit %s, %s requests per launch" % (
            k, *("%.3f" % e[x] if x in e else "  -  " for x in ("mfma_busy", "parked", "issue_stalled", "issuing", "l2_hit")),
            "%.3g" % e["l2_requests_per_launch"] if "l2_requests_per_launch" in e else "-"))
        for blk, d in (("SQ ", s), ("TCC", t)):
            lines.append("    %s raw per launch: %s" % (blk, ", ".join("%s %.4g" % kv for kv in sorted(d.items()))))
    d = json.load(open(traffic_json))
    d["counters"] = {k: {x: v for x, v in e.items() if x != "raw"} for k, e in out.items()}
    d["counters_note"] = ("rocprofv3 --kernel-trace --pmc, one SQ pass and one TCC pass over the same workload as the traffic passes; "
                          "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS)")
    json.dump(d, open(traffic_json, "w"), indent=1)
    open(summary, "w").write("\n".join(lines) + "\n")
    print("\n".join(l for l in lines if not l.startswith("    ")))


if __name__ == "__main__":
    main(*sys.argv[1:5])
