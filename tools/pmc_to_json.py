"""rocprofv3 counter_collection.csv (FETCH_SIZE pass, WRITE_SIZE pass) -> profiles/pmc_traffic.json: HBM bytes per launch per kernel.
Correction (MI355X_MICROARCH.md §HBM): on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced streams; the
factor is calibrated here on a device-to-device copy of known size collected in the same pass (kernel name contains 'copy')."""
import csv, json, os, re, sys
from collections import defaultdict


def load(path):
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    with open(path) as f:
        for r in csv.DictReader(f):
            a = agg[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return agg


def main(fetch_csv, write_csv, out, copy_bytes, num_envs=4096):
    F, W = load(fetch_csv), load(write_csv)
    # calibration on the largest-FETCH copy/elementwise kernel
    cal_f = cal_w = None
    for k, cs in F.items():
        if "copy" in k.lower() or "elementwise" in k:
            v = cs["FETCH_SIZE"][0] / cs["FETCH_SIZE"][1] * 1024
            if v > 0.2 * copy_bytes:
                cal_f = copy_bytes / v
                wv = W[k]["WRITE_SIZE"]
                if wv[1]:
                    cal_w = copy_bytes / (wv[0] / wv[1] * 1024)
    res = {}
    names = {"rollout_step_kernel": "rollout_step_kernel", "mlp_fb_kernel": "mlp_fb_kernel",
             "env_step_kernel": "env_step_kernel", "mlp_fwd_kernelILi64": "mlp_fwd_kernel", "mlp_fwd_kernel<64": "mlp_fwd_kernel",
             "mlp_fwd_kernelILi32": "mlp_fwd_kernel<32>", "mlp_fwd_kernel<32": "mlp_fwd_kernel<32>", "mlp_bwd_kernel": "mlp_bwd_kernel",
             "dw_kernel": "dw_kernel", "ppo_loss_kernel": "ppo_loss_kernel", "reduce_slabs_kernel": "reduce_slabs_kernel",
             "adam_kernel": "sqnorm+adam_kernel", "gae_kernel": "gae_kernel"}
    # several instantiations of one kernel run in an iteration (the rollout's first vec-step without the carried-over side jobs, then 59
    # launches of the steady form): the class is represented by the instantiation with the MOST launches, named in `variants`
    variants = {}
    for k, cs in sorted(F.items(), key=lambda kv: kv[1]["FETCH_SIZE"][1]):
        for pat, nm in names.items():
            if pat in k:
                variants[nm] = k.split("(")[0][-120:]
                f = cs["FETCH_SIZE"][0] / cs["FETCH_SIZE"][1] * 1024
                w = W[k]["WRITE_SIZE"][0] / max(W[k]["WRITE_SIZE"][1], 1) * 1024
                res[nm] = dict(fetch_raw=f, write_raw=w, fetch_bytes=f * (cal_f or 1.0), write_bytes=w * (cal_w or 1.0),
                               hbm_bytes=f * (cal_f or 1.0) + w * (cal_w or 1.0), launches=cs["FETCH_SIZE"][1])
    import datetime, subprocess
    def sh(cmd):
        try:
            return subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=20).stdout.strip()
        except Exception:
            return ""
    prov = dict(collected_utc=datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%M:%SZ"),
                git_head=sh("git -C %s rev-parse --short HEAD" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) or
                         "(no .git on the GPU box: stamped by tools/stamp_traffic.py when the file is committed)",
                gpu=sh("rocm-smi --showproductname 2>/dev/null | grep -m1 'Card Series' | sed 's/.*: *//'") or sh("rocminfo | grep -m1 'Marketing Name' | sed 's/.*: *//'"),
                workload="tools/traffic_run.py: bench.py --steps 2 --warmup 2 (%d envs, bf16), averaged per kernel instantiation over all its launches" % num_envs)
    json.dump(dict(note="HBM bytes per launch: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), scaled by the factor that makes a "
                        "device copy of known size read right", provenance=prov, fetch_calibration=cal_f, write_calibration=cal_w, num_envs=num_envs, variants=variants,
                   kernels={k: v["hbm_bytes"] for k, v in res.items()}, detail=res), open(out, "w"), indent=1)
    print(json.dumps(dict(cal_f=cal_f, cal_w=cal_w, kernels={k: round(v["hbm_bytes"] / 1e6, 2) for k, v in res.items()})))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 4096)
