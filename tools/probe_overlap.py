"""Would the update's two big launches overlap usefully?  mlp_fb (write-heavy) and dw_kernel_rs (read-heavy) of two independent
nets on two streams, N launches each: sequential on one stream vs concurrent on two.  HGYM_GRAD_ONLY selects the launch."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "humanoid-gym_amd"))
import torch
from hgym import NetBuffers, make_net_config, make_ppo_config, make_batch, _lib as L

dev = "cuda"
B, S = 61440, 245760
ppo = make_ppo_config()


def mk():
    cfg = make_net_config(705, 219, 12, [512, 256, 128], [768, 256, 128], "bf16", B)
    net = NetBuffers(cfg, dev, learning_rate=1e-5)
    for k, v in net.views.items():
        v.copy_(torch.randn(v.shape, device=dev) * (0.05 if v.dim() > 1 else 0.01))
    net.views["std"].fill_(1.0)
    net.sync_shadow()
    return net


obs, priv = torch.randn(S, 705, device=dev), torch.randn(S, 219, device=dev)
act, mu_o = torch.randn(S, 12, device=dev), torch.randn(S, 12, device=dev) * 0.3
sg_o = torch.ones(S, 12, device=dev)
val, adv, ret = torch.randn(S, device=dev), torch.randn(S, device=dev), torch.randn(S, device=dev)
lp_o = -12.0 + torch.randn(S, device=dev)
n1, n2 = mk(), mk()
so = torch.zeros(S, n1.shadow_ld(0), dtype=torch.bfloat16, device=dev); so[:, :705] = obs.to(torch.bfloat16)
sp = torch.zeros(S, n1.shadow_ld(1), dtype=torch.bfloat16, device=dev); sp[:, :219] = priv.to(torch.bfloat16)
idx1 = torch.randperm(S, device=dev)[:B].contiguous()
idx2 = torch.randperm(S, device=dev)[:B].contiguous()
b1 = make_batch(obs, priv, act, val, adv, ret, lp_o, mu_o, sg_o, idx1, obs_bf16=so, priv_bf16=sp)
b2 = make_batch(obs, priv, act, val, adv, ret, lp_o, mu_o, sg_o, idx2, obs_bf16=so, priv_bf16=sp)
for n, b in ((n1, b1), (n2, b2)):
    for _ in range(2):
        n.ppo_grad(ppo, b)
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
N = 10


def run(mode):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode == "fb":
        os.environ["HGYM_GRAD_ONLY"] = "fb"
        with torch.cuda.stream(s1):
            for _ in range(N): n1.ppo_grad(ppo, b1)
    elif mode == "dw":
        os.environ["HGYM_GRAD_ONLY"] = "dw"
        with torch.cuda.stream(s1):
            for _ in range(N): n2.ppo_grad(ppo, b2)
    elif mode == "seq":
        with torch.cuda.stream(s1):
            for _ in range(N):
                os.environ["HGYM_GRAD_ONLY"] = "fb"; n1.ppo_grad(ppo, b1)
                os.environ["HGYM_GRAD_ONLY"] = "dw"; n2.ppo_grad(ppo, b2)
    elif mode == "par":
        for _ in range(N):
            os.environ["HGYM_GRAD_ONLY"] = "fb"
            with torch.cuda.stream(s1): n1.ppo_grad(ppo, b1)
            os.environ["HGYM_GRAD_ONLY"] = "dw"
            with torch.cuda.stream(s2): n2.ppo_grad(ppo, b2)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6


for fb2 in (True, False):
    if fb2: os.environ["HGYM_FB2"] = "1"
    else: os.environ.pop("HGYM_FB2", None)
    for rep in range(2):
        r = {m: run(m) for m in ("fb", "dw", "seq", "par")}
        print("%s rep %d: fb alone %.1f us, dw alone %.1f us, fb then dw on one stream %.1f us, fb || dw on two streams %.1f us per pair"
              % ("mlp_fb2<128>" if fb2 else "mlp_fb<64>  ", rep, r["fb"], r["dw"], r["seq"], r["par"]))
