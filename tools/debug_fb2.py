"""Which tensors of the minibatch gradient differ between mlp_fb2_kernel (128-row tiles) and mlp_fb_kernel (default; HGYM_FB2=1 selects the 128-row kernel)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "humanoid-gym_amd")); sys.path.insert(0, ROOT)
import torch
from hgym import NetBuffers, make_net_config, make_ppo_config, make_batch

dev = "cuda"
for S, B in [(700, 333), (2000, 1024), (5000, 4096), (61440, 61440)]:
    cfg = make_net_config(705, 219, 12, [512, 256, 128], [768, 256, 128], "bf16", max(B, 512))
    net = NetBuffers(cfg, dev)
    g = torch.Generator(device=dev).manual_seed(S)
    for k, v in net.views.items():
        v.copy_(torch.randn(v.shape, device=dev, generator=g) * (0.05 if v.dim() > 1 else 0.01))
    net.views["std"].fill_(1.0)
    net.sync_shadow()
    obs, priv = torch.randn(S, 705, device=dev, generator=g).clamp_(-18, 18), torch.randn(S, 219, device=dev, generator=g).clamp_(-18, 18)
    act, mu_o = torch.randn(S, 12, device=dev, generator=g), torch.randn(S, 12, device=dev, generator=g) * 0.3
    sg_o = torch.rand(S, 12, device=dev, generator=g) * 0.5 + 0.75
    val, adv, ret = (torch.randn(S, device=dev, generator=g) for _ in range(3))
    lp_o = -12.0 + torch.randn(S, device=dev, generator=g)
    idx = torch.randperm(S, device=dev, generator=g)[:B].contiguous()
    so = torch.zeros(S, 768, dtype=torch.bfloat16, device=dev); so[:, :705] = obs.to(torch.bfloat16)
    sp = torch.zeros(S, 256, dtype=torch.bfloat16, device=dev); sp[:, :219] = priv.to(torch.bfloat16)
    cols = (obs, priv, act, val, adv, ret, lp_o, mu_o, sg_o, idx)
    res = {}
    for mode in ("old", "fb2", "fb2b"):
        if mode == "old":
            os.environ.pop("HGYM_FB2", None)
        else:
            os.environ["HGYM_FB2"] = "1"
        net.grads_ext.zero_(); net.opt_state[2:10] = 0.0
        net.ppo_grad(make_ppo_config(), make_batch(*cols, obs_bf16=so, priv_bf16=sp))
        torch.cuda.synchronize()
        res[mode] = ({k: v.clone() for k, v in net.grad_views().items()}, net.opt_state.clone(), net.grads_ext[-1].clone())
    print("S=%d B=%d" % (S, B))
    for other in ("fb2", "fb2b"):
        for k in res["old"][0]:
            a, b = res["old"][0][k], res[other][0][k]
            if not torch.equal(a, b):
                d = (a - b).abs()
                bad = (d > 0).nonzero()
                print("  %-5s %-18s differs: %d of %d entries, max abs %.3e (ref max %.3e), first at %s, rows hit %s" % (
                    other, k, int((d > 0).sum()), a.numel(), float(d.max()), float(a.abs().max()), bad[0].tolist(),
                    sorted(set(bad[:, 0].tolist()))[:12] if bad.dim() > 1 else ""))
        print("   old", res["old"][1][2:10].tolist()); print("   new", res[other][1][2:10].tolist())
        print("  %s opt equal: %s kl equal: %s" % (other, torch.equal(res["old"][1][2:9], res[other][1][2:9]), torch.equal(res["old"][2], res[other][2])))
