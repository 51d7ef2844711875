import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "humanoid-gym_amd")); sys.path.insert(0, R)
import numpy as np, torch
from oracle import ppo_oracle as P
from oracle import xbot_constants as K
from hgym import NetBuffers, make_net_config, make_ppo_config, make_batch
NAMES = ["std"] + ["actor.%d.%s" % (i, k) for i in (0, 2, 4, 6) for k in ("weight", "bias")] + ["critic.%d.%s" % (i, k) for i in (0, 2, 4, 6) for k in ("weight", "bias")]
S, B = 5000, 4096
g = torch.Generator().manual_seed(S)
p = P.Params.random(705, 219, 12, K.ACTOR_HIDDEN, K.CRITIC_HIDDEN, g)
p.std = torch.rand(12, generator=g) * 0.5 + 0.75
obs, priv = torch.randn(S, 705, generator=g), torch.randn(S, 219, generator=g)
act, mu_o = torch.randn(S, 12, generator=g), torch.randn(S, 12, generator=g) * 0.3
sg_o = torch.rand(S, 12, generator=g) * 0.5 + 0.75
val, adv, ret = torch.randn(S, generator=g), torch.randn(S, generator=g), torch.randn(S, generator=g)
mu_now = P.mlp_forward(obs, p.actor)
lp_o = P.gaussian_log_prob(act, mu_now, mu_now * 0 + p.std) + torch.randn(S, generator=g) * 0.3
idx = torch.randperm(S, generator=g)[:B].contiguous()
out = P.ppo_loss_and_grads(p, obs[idx], priv[idx], act[idx], val[idx], adv[idx], ret[idx], lp_o[idx], mu_o[idx], sg_o[idx])
ref = dict(zip(NAMES, out["grads"].tensors()))
c = lambda t: t.cuda().contiguous()
keep = [c(obs), c(priv), c(act), c(val), c(adv), c(ret), c(lp_o), c(mu_o), c(sg_o), c(idx)]
res = {}
for tag in ("fused", "generic", "f32"):
    if tag == "generic": os.environ["HGYM_NO_FUSED"] = "1"
    net = NetBuffers(make_net_config(705, 219, 12, K.ACTOR_HIDDEN, K.CRITIC_HIDDEN, "f32" if tag == "f32" else "bf16", B), "cuda")
    net.load_state_dict(dict(zip(NAMES, p.tensors())))
    net.ppo_grad(make_ppo_config(), make_batch(*keep)); torch.cuda.synchronize()
    res[tag] = {k: v.cpu().double() for k, v in net.grad_views().items()}
    os.environ.pop("HGYM_NO_FUSED", None)
for k in NAMES:
    b = ref[k].double()
    print("%-16s" % k, " ".join("%s %.4f" % (t, float((res[t][k] - b).norm() / b.norm())) for t in res))
a, b = res["fused"]["actor.0.weight"], ref["actor.0.weight"].double()
e = (a - b)
cn = e.norm(dim=0) / b.norm(dim=0).clamp_min(1e-30)
print("worst cols", torch.topk(cn, 8))
rn = e.norm(dim=1) / b.norm(dim=1).clamp_min(1e-30)
print("worst rows", torch.topk(rn, 8), "median col %.4f row %.4f" % (float(cn.median()), float(rn.median())))
