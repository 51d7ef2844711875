"""GPU probe: per-WAVEFRONT clock of mlp_fb4_kernel's first layer (library variant built with -DFB3_WAVE_CLOCK -DFB3_WSLOTS=16 [-DFB4_CLOCK_PASS=p]):
compute waves stamp slot 2c after the barrier that opens chunk c, 2c + 1 after its MFMAs, 12 / 13 / 14 around the epilogue; service waves 2c when
chunk c is staged (before the barrier), 2c + 1 after it."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "humanoid-gym_amd"))
import torch
from hgym import NetBuffers, make_net_config, make_ppo_config, make_batch, _lib as L

dev = "cuda"
S = B = 61440
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
idx = torch.randperm(S, device=dev).contiguous()
if os.environ.get('HGYM_IDX0'):
    idx = torch.randint(0, 64, (S,), device=dev)      # every gather hits L2 / L1
so = torch.zeros(S, net.shadow_ld(0), dtype=torch.bfloat16, device=dev); so[:, :705] = obs.to(torch.bfloat16)
sp = torch.zeros(S, net.shadow_ld(1), dtype=torch.bfloat16, device=dev); sp[:, :219] = priv.to(torch.bfloat16)
batch = make_batch(obs, priv, act, val, adv, ret, lp_o, mu_o, sg_o, idx, obs_bf16=so, priv_bf16=sp)
ppo = make_ppo_config()
for _ in range(3):
    net.ppo_grad(ppo, batch)
torch.cuda.synchronize()
nt = B // 128
WS = 16
buf = torch.zeros(2 * nt * 16 * WS, dtype=torch.int64, device=dev)
L.check(L.lib.hgym_prof_phase_buffer(C.c_void_p(buf.data_ptr()), buf.numel()))
net.ppo_grad(ppo, batch)
torch.cuda.synchronize()
t = buf.view(2 * nt, 16, WS).cpu().double() * 0.01      # us
for lo, hi, tag, nc in ((0, nt, "actor", 6), (nt, 2 * nt, "critic", 2)):
    d = t[lo:hi]
    t0 = d[:, :8, 0].min(dim=1).values
    rel = lambda w, s: (d[:, w, s] - t0).mean().item()
    print("%s: layer 0 pass, us after the first compute wave left the chunk-0 barrier (mean over %d tiles)" % (tag, hi - lo))
    for w in (0, 3, 4, 7):
        print("  compute wave %d: " % w + " | ".join("c%d %.2f-%.2f" % (c, rel(w, 2 * c), rel(w, 2 * c + 1)) for c in range(nc)) +
              " | Yfree %.2f epi %.2f ready %.2f" % (rel(w, 12), rel(w, 13), rel(w, 14)))
    for w in (8, 11):
        print("  service wave %d: " % (w - 8) + " | ".join("c%d staged %.2f passed %.2f" % (c, rel(w, 2 * c), rel(w, 2 * c + 1)) for c in range(nc)))
L.check(L.lib.hgym_prof_phase_buffer(None, 0))
