"""GPU probe: phase clock of mlp_fb2_kernel (128-row tiles) per net, next to mlp_fb_kernel's (default), B = 61 440."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "humanoid-gym_amd"))
import torch
from hgym import NetBuffers, make_net_config, make_ppo_config, make_batch, _lib as L

dev = "cuda"
B = int(os.environ.get("HGYM_B", 61440))
S = int(os.environ.get("HGYM_S", 245760))
cfg = make_net_config(705, 219, 12, [512, 256, 128], [768, 256, 128], "bf16", B)
net = NetBuffers(cfg, dev, learning_rate=1e-5)
for k, v in net.views.items():
    v.copy_(torch.randn(v.shape, device=dev) * (0.05 if v.dim() > 1 else 0.01))
net.views["std"].fill_(1.0)
net.sync_shadow()
obs, priv = torch.randn(S, 705, device=dev), torch.randn(S, 219, device=dev)
act, mu_o = torch.randn(S, 12, device=dev), torch.randn(S, 12, device=dev) * 0.3
sg_o = torch.ones(S, 12, device=dev)
val, adv, ret = torch.randn(S, device=dev), torch.randn(S, device=dev), torch.randn(S, device=dev)
lp_o = -12.0 + torch.randn(S, device=dev)
idx = torch.randperm(S, device=dev)[:B].contiguous()
ppo = make_ppo_config()
so = torch.zeros(S, net.shadow_ld(0), dtype=torch.bfloat16, device=dev); so[:, :705] = obs.to(torch.bfloat16)
sp = torch.zeros(S, net.shadow_ld(1), dtype=torch.bfloat16, device=dev); sp[:, :219] = priv.to(torch.bfloat16)
batch = make_batch(obs, priv, act, val, adv, ret, lp_o, mu_o, sg_o, idx, obs_bf16=so, priv_bf16=sp)


def report(title, buf, nblk, names, half):
    t = buf[:nblk * 8].view(nblk, 8).cpu().double() * 0.01      # us (100 MHz clock)
    t0 = t[:, 0].min().item()
    for lo, hi, tag in ((0, half, "actor"), (half, nblk, "critic")):
        d = t[lo:hi]
        segs = [(d[:, i + 1] - d[:, i]).mean().item() for i in range(len(names))]
        tot = (d[:, len(names)] - d[:, 0])
        st = d[:, 0] - t0
        if os.environ.get("ROUNDS", "1") == "1":
            first = st < (st.min() + 5.0)
            for nm, sel in (("round 1 (synchronised start)", first), ("later rounds", ~first)):
                if int(sel.sum()) == 0:
                    continue
                dd = d[sel]
                sg = [(dd[:, i + 1] - dd[:, i]).mean().item() for i in range(len(names))]
                print("    %s %s, %d tiles, mean %.1f us: " % (tag, nm, int(sel.sum()), (dd[:, len(names)] - dd[:, 0]).mean().item()) +
                      ", ".join("%s %.1f" % (n, x) for n, x in zip(names, sg)))
        print("%s %s (%d tiles): tile mean %.1f us (min %.1f, p90 %.1f, max %.1f), starts: first %.1f median %.1f last %.1f, last end %.1f: " % (
            title, tag, hi - lo, tot.mean().item(), tot.min().item(), tot.quantile(0.9).item(), tot.max().item(), st.min().item(), st.median().item(), st.max().item(),
            (d[:, len(names)] - t0).max().item()) + ", ".join("%s %.1f" % (n, s) for n, s in zip(names, segs)))


def ev_time(reps=10):
    L.lib.hgym_prof_enable(1)
    for _ in range(reps):
        net.ppo_grad(ppo, batch)
    torch.cuda.synchronize()
    n, ms, work = L.prof_summary(4)
    L.lib.hgym_prof_enable(0)
    return ms / max(n, 1) * 1e3


for mode in ("fb2", "old"):
    if mode == "old":
        os.environ.pop("HGYM_FB2", None)
    else:
        os.environ["HGYM_FB2"] = "1"
    for _ in range(3):
        net.ppo_grad(ppo, batch)
    torch.cuda.synchronize()
    print(mode, "HIP events, no phase buffer: %.1f us per launch" % ev_time())
    nb = (B // 64) * 2
    buf = torch.zeros(nb * 8, dtype=torch.int64, device=dev)
    L.check(L.lib.hgym_prof_phase_buffer(C.c_void_p(buf.data_ptr()), buf.numel()))
    print(mode, "HIP events, phase buffer set: %.1f us per launch" % ev_time())
    buf.zero_()
    net.ppo_grad(ppo, batch)
    torch.cuda.synchronize()
    if mode == "fb2":
        T = int(round(B / 16 / 7.5)) if B == 61440 else None
        tiles = int((buf.view(-1, 8)[:, 0] != 0).sum()) // 2
        report("mlp_fb2<128>", buf, 2 * tiles, ["entry", "layer0", "H0 epilogue (+layer1 pieces)", "layer1 + H1", "layer2 + H2", "head + loss", "dZ chain"], tiles)
    else:
        report("mlp_fb<64>", buf, nb, ["input0", "layer0 k-loop", "epilogue0+sync", "layer1+sync", "layer2+sync", "head", "loss + dZ chain"], nb // 2)
    L.check(L.lib.hgym_prof_phase_buffer(None, 0))
