"""GPU probe: per-WAVEFRONT clock of mlp_fb3_kernel's second layer (library variant built with -DFB3_WAVE_CLOCK: lane 0 of every wavefront
stamps slot 0 after the barrier that opens the layer, 1 after its MFMA stream, 2 after its epilogue, 3 after the barrier that closes it)."""
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
so = torch.zeros(S, net.shadow_ld(0), dtype=torch.bfloat16, device=dev); so[:, :705] = obs.to(torch.bfloat16)
sp = torch.zeros(S, net.shadow_ld(1), dtype=torch.bfloat16, device=dev); sp[:, :219] = priv.to(torch.bfloat16)
batch = make_batch(obs, priv, act, val, adv, ret, lp_o, mu_o, sg_o, idx, obs_bf16=so, priv_bf16=sp)
ppo = make_ppo_config()
for _ in range(3):
    net.ppo_grad(ppo, batch)
torch.cuda.synchronize()
nb = (B // 64) * 2
buf = torch.zeros(nb * 128, dtype=torch.int64, device=dev)
L.check(L.lib.hgym_prof_phase_buffer(C.c_void_p(buf.data_ptr()), buf.numel()))
net.ppo_grad(ppo, batch)
torch.cuda.synchronize()
t = buf.view(nb, 16, 8).cpu().double() * 0.01      # us
for lo, hi, tag in ((0, nb // 2, "actor"), (nb // 2, nb, "critic")):
    d = t[lo:hi]
    t0 = d[:, :8, 0].min(dim=1, keepdim=True).values          # first compute wave out of the opening barrier
    rel = lambda w, s: (d[:, w, s] - t0[:, 0])
    print("%s: layer 1, us after the first wave left the opening barrier (mean over %d tiles)" % (tag, hi - lo))
    for w in range(8):
        print("  compute wave %d: start %.2f  mma done %.2f  epilogue done %.2f  barrier passed %.2f" % (
            w, rel(w, 0).mean(), rel(w, 1).mean(), rel(w, 2).mean(), rel(w, 3).mean()))
    for w in range(8, 12):
        print("  service wave %d: start %.2f  copy issued %.2f  barrier passed %.2f" % (w - 8, rel(w, 0).mean(), rel(w, 2).mean(), rel(w, 3).mean()))
L.check(L.lib.hgym_prof_phase_buffer(None, 0))
