"""GPU micro-benchmark of the PPO minibatch (B = 61 440) and the rollout policy step (M = 4096) through the C-ABI,
with a calibration line (torch bf16 matmul + device copy) so runs on different boxes / clock states can be compared."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "humanoid-gym_amd"))
import torch
from hgym import NetBuffers, make_net_config, make_ppo_config, make_batch, _lib as L

dev = "cuda"
def timeit(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16); b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
us = timeit(lambda: a @ b, 10)
src = torch.empty(1 << 28, device=dev, dtype=torch.uint8); dst = torch.empty_like(src)
uc = timeit(lambda: dst.copy_(src), 10)
print("calib: matmul8192 bf16 %.0f us (%.0f TF/s)  copy256MB %.0f us (%.2f TB/s r+w)" % (us, 2 * 8192**3 / us / 1e6, uc, 2 * (1 << 28) / uc / 1e6))

B = int(os.environ.get('HGYM_B', 61440))      # minibatch rows (61 440 = XBot-L: 4096 envs x 60 steps / 4)
S = int(os.environ.get('HGYM_S', B))      # storage rows the minibatch is drawn from (245760 = the real XBot-L storage)
cfg = make_net_config(705, 219, 12, [512, 256, 128], [768, 256, 128], "bf16", B)
net = NetBuffers(cfg, dev, learning_rate=1e-5)
g = torch.Generator(device=dev).manual_seed(0)
for k, v in net.views.items():
    v.copy_(torch.randn(v.shape, device=dev, generator=g) * (0.05 if v.dim() > 1 else 0.01))
net.views["std"].fill_(1.0)
net.sync_shadow()
obs, priv = torch.randn(S, 705, device=dev), torch.randn(S, 219, device=dev)
act, mu_o = torch.randn(S, 12, device=dev), torch.randn(S, 12, device=dev) * 0.3
sg_o = torch.ones(S, 12, device=dev)
val, adv, ret = torch.randn(S, device=dev), torch.randn(S, device=dev), torch.randn(S, device=dev)
lp_o = -12.0 + torch.randn(S, device=dev)
idx = torch.randperm(S, device=dev)[:B].contiguous()
if os.environ.get('HGYM_IDX_SEQ'):
    idx = torch.arange(B, device=dev)        # the indirection without the scatter: rows 0..B-1 in order
if os.environ.get('HGYM_SORT'):
    idx = idx.sort().values.contiguous()      # same minibatch (as a set), ascending storage order
if os.environ.get('HGYM_IDX0'):
    idx = torch.randint(0, 64, (B,), device=dev)   # every gather hits L2: isolates the input-latency share of mlp_fwd
ppo = make_ppo_config(grad_norm_ready=True)      # what PPO.update passes on one rank
shadow = {}
if os.environ.get('HGYM_BU_SHADOW', '1') != '0' and net.shadow_ld(0) > 0:      # bf16 shadows of the storage rows (what the rollout leaves behind)
    so = torch.zeros(S, net.shadow_ld(0), dtype=torch.bfloat16, device=dev); so[:, :705] = obs.to(torch.bfloat16)
    sp = torch.zeros(S, net.shadow_ld(1), dtype=torch.bfloat16, device=dev); sp[:, :219] = priv.to(torch.bfloat16)
    shadow = dict(obs_bf16=so, priv_bf16=sp)
print("shadow:", bool(shadow))
batch = make_batch(obs, priv, act, val, adv, ret, lp_o, mu_o, sg_o, idx, **shadow)
names = {0: "gemm(all)", 3: "loss", 4: "mlp_fwd", 5: "mlp_bwd", 6: "dw", 7: "reduce", 8: "apply", 9: "policy"}
CH = int(os.environ.get('HGYM_CHUNKS', '1'))      # timing experiment: the minibatch as CH back-to-back gradient calls over B / CH rows each
chunks = [make_batch(obs, priv, act, val, adv, ret, lp_o, mu_o, sg_o, idx[c * (B // CH):(c + 1) * (B // CH)].contiguous(), **shadow) for c in range(CH)] if CH > 1 else [batch]
PARTS = os.environ.get('HGYM_PARTS') == '1'       # the data-parallel update's two gradient halves (hgym_ppo_grad_part 0, 1) on one rank
def step():
    for bt in chunks:
        if PARTS:
            net.ppo_grad_part(ppo, bt, 0); net.ppo_grad_part(ppo, bt, 1)
        else:
            net.ppo_grad(ppo, bt)
    net.ppo_apply(ppo)
t = timeit(step, 20)
print("minibatch grad+apply: %.1f us" % t)
L.lib.hgym_prof_enable(1)
for _ in range(10): step()
torch.cuda.synchronize()
for c, nm in names.items():
    try:
        n, ms, work = L.prof_summary(c)
    except Exception:
        continue
    if n: print("  %-10s launches %3d avg %8.1f us  work/s %.1f T" % (nm, n, ms / n * 1e3, work / (ms * 1e-3) / 1e12 if ms > 0 else 0))
L.lib.hgym_prof_enable(0)
M = 4096
o4, p4 = torch.randn(M, 705, device=dev), torch.randn(M, 219, device=dev)
step_c = torch.zeros(1, dtype=torch.int64, device=dev)
out = net.act(o4, p4, seed=1, step_counter=step_c)
t = timeit(lambda: net.act(o4, p4, seed=1, step_counter=step_c, out=out), 50)
print("policy_act M=4096: %.1f us" % t)
M = 61440
obs_m, priv_m = obs[:M], priv[:M]       # (HGYM_S may have made the storage larger than the net's max batch)
out2 = net.act(obs_m, priv_m, seed=1, step_counter=step_c)
t = timeit(lambda: net.act(obs_m, priv_m, seed=1, step_counter=step_c, out=out2), 20)
print("policy_act M=61440 (64-row tiles, no activation stores, no gather): %.1f us" % t)
