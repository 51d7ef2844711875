"""GPU probe: kernel-internal phase clock of the fused MLP kernels (hgym_prof_phase_buffer): per-phase mean durations
per workgroup for the update forward / backward (B = 61 440) and the rollout policy step (M = 4096)."""
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
    idx = torch.randint(0, 64, (S,), device=dev)      # every gather hits L2 / L1: the phases without the input rows' HBM latency
ppo = make_ppo_config()
shadow = {}
if os.environ.get('HGYM_BU_SHADOW', '1') != '0' and net.shadow_ld(0) > 0:
    so = torch.zeros(S, net.shadow_ld(0), dtype=torch.bfloat16, device=dev); so[:, :705] = obs.to(torch.bfloat16)
    sp = torch.zeros(S, net.shadow_ld(1), dtype=torch.bfloat16, device=dev); sp[:, :219] = priv.to(torch.bfloat16)
    shadow = dict(obs_bf16=so, priv_bf16=sp)
print("shadow:", bool(shadow))
batch = make_batch(obs, priv, act, val, adv, ret, lp_o, mu_o, sg_o, idx, **shadow)
for _ in range(3):
    net.ppo_grad(ppo, batch)
torch.cuda.synchronize()

FWD = ["input0", "layer0 k-loop", "epilogue0+sync", "layer1+sync", "layer2+sync", "head"]
BWD = ["load dZ3+sync", "through W3+sync", "through W2+sync", "through W1"]


def report(title, buf, nblk, names, half):
    t = buf[:nblk * 8].view(nblk, 8).cpu().double() * 0.01      # us
    for lo, hi, tag in ((0, half, "actor"), (half, nblk, "critic")):
        d = t[lo:hi]
        segs = [(d[:, i + 1] - d[:, i]).mean().item() for i in range(len(names))]
        tot = (d[:, len(names)] - d[:, 0]).mean().item()
        span = (d[:, len(names)].max() - d[:, 0].min()).item()
        print("%s %s: block mean %.1f us (grid span %.1f us): " % (title, tag, tot, span) +
              ", ".join("%s %.1f" % (n, s) for n, s in zip(names, segs)))


nb = (B // 64) * 2
TR = int(os.environ.get("HGYM_PROBE_TILE_ROWS", "64"))      # rows per tile of the update kernel under test (mlp_fb4_kernel: 128)
buf = torch.zeros(nb * 8, dtype=torch.int64, device=dev)
L.check(L.lib.hgym_prof_phase_buffer(C.c_void_p(buf.data_ptr()), buf.numel()))
net.ppo_grad(ppo, batch)      # fwd and bwd both write (same grid): run bwd-only view second
torch.cuda.synchronize()
# the bwd launch overwrote slots 0..4 of the fwd stamps; take fwd from a forward-only call
which = os.environ.get("PHASE", "both")
buf.zero_()
M = 61440
o = net.act(obs, priv, seed=1, step_counter=torch.zeros(1, dtype=torch.int64, device=dev))
torch.cuda.synchronize()
report("fwd<64> (no stores, M=61440)", buf, nb, FWD, nb // 2)
buf.zero_()
net.ppo_grad(ppo, batch)
torch.cuda.synchronize()
report("mlp_fb<%d> (forward + loss + dZ chain, B=61440)" % TR, buf, (B // TR) * 2, FWD + ["loss + dZ chain"], B // TR)
M = 4096
o4, p4 = torch.randn(M, 705, device=dev), torch.randn(M, 219, device=dev)
sc = torch.zeros(1, dtype=torch.int64, device=dev)
out = net.act(o4, p4, seed=1, step_counter=sc)
buf.zero_()
net.act(o4, p4, seed=1, step_counter=sc, out=out)
torch.cuda.synchronize()
report("policy<32> M=4096", buf, 256, FWD, 128)
L.check(L.lib.hgym_prof_phase_buffer(None, 0))
