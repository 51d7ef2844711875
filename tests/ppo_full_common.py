"""Helpers of the full-width PPO-update fixture (tests/golden/ppo_update_full.npz + ppo_full_case.py): load the recorded
reference outputs, regenerate the inputs, compare gradients / parameter changes tensor by tensor."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import ppo_full_case as CASE  # noqa: E402


def load():
    G = np.load(os.path.join(HERE, "golden", "ppo_update_full.npz"))
    assert int(G["seed"]) == CASE.SEED and int(G["N"]) == CASE.N and int(G["T"]) == CASE.T
    return G, CASE.initial_parameters(), CASE.rollout_inputs()


def compare(G, prefix, got, report=None):
    """got: name -> flat-able fp32 array of what the reference recorded under `prefix` ("g0": clipped gradient of the first
    minibatch, "dP": parameter change over the 8 Adam steps).  Returns per-tensor dicts: rel-L2 error and cosine against the
    fp16 full copy (or the fp32 copy when the tensor is small), max error of the fp32-exact sample relative to the tensor's
    largest sampled magnitude, and the norm ratio."""
    out = {}
    scale = float(G[prefix + "_h16_scale"])
    for name in CASE.NAMES:
        key = name.replace(".", "_")
        a = np.asarray(got[name], dtype=np.float64).reshape(-1)
        idx = CASE.sample_index(name, a.size)
        s32 = G["%s_s32_%s" % (prefix, key)].astype(np.float64)
        ref = G["%s_h16_%s" % (prefix, key)].astype(np.float64) / scale if ("%s_h16_%s" % (prefix, key)) in G.files else s32
        nr = float(G["%s_norm_%s" % (prefix, key)])
        d = dict(rel_l2=float(np.linalg.norm(a - ref) / max(np.linalg.norm(ref), 1e-30)),
                 cos=float(a @ ref / max(np.linalg.norm(a) * np.linalg.norm(ref), 1e-30)),
                 sample_max_err=float(np.abs(a[idx] - s32).max() / max(np.abs(s32).max(), 1e-30)),
                 norm_ratio=float(np.linalg.norm(a) / max(nr, 1e-30)))
        out[name] = d
        if report is not None:
            report.append("%-16s rel_l2 %.3e  cos %.6f  sample_max_err %.3e  |got|/|ref| %.4f" % (name, d["rel_l2"], d["cos"], d["sample_max_err"], d["norm_ratio"]))
    return out


def total_cosine(G, prefix, got):
    num = da = db = 0.0
    scale = float(G[prefix + "_h16_scale"])
    for name in CASE.NAMES:
        key = name.replace(".", "_")
        a = np.asarray(got[name], dtype=np.float64).reshape(-1)
        ref = G["%s_h16_%s" % (prefix, key)].astype(np.float64) / scale if ("%s_h16_%s" % (prefix, key)) in G.files else \
            G["%s_s32_%s" % (prefix, key)].astype(np.float64)
        num += float(a @ ref); da += float(a @ a); db += float(ref @ ref)
    return num / (da ** 0.5 * db ** 0.5)
