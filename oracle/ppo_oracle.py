"""CPU oracle for the algo side of the hot path.

TEST INFRASTRUCTURE.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this; the product path never does.

A from-scratch restatement, as explicit array arithmetic WITHOUT autograd and without torch.nn /
torch.optim / torch.distributions, of
  algo/ppo/rollout_storage.py:122-136,146-182   (GAE, advantage normalisation, minibatch order)
  algo/ppo/actor_critic.py:53-128               (two ELU MLPs, state-independent Gaussian)
  algo/ppo/ppo.py:91-184                        (act, time-out bootstrap, clipped PPO loss, adaptive-KL
                                                 learning rate, grad-norm clip, Adam)
The backward pass is written out by hand (it is the specification the HIP loss / backward / Adam
kernels implement); tests/test_oracle_algo_golden.py pins values, clipped gradients, per-step learning
rates and final parameters against tests/golden/ppo_update.npz, recorded from the reference's own
PPO.update() (autograd + torch.optim.Adam).  torch CPU fp32 tensors are used as the array type because
the reference is torch fp32 code (same primitive kernels => tight pin).
"""
import math

import torch
import torch.nn.functional as F

HALF_LOG_2PI = 0.5 * math.log(2 * math.pi)


# ------------------------------------------------------------------------------------------------ A5
def gae_returns(rewards, values, dones, last_values, gamma, lam):
    """rollout_storage.py:122-133.  rewards/values (T,N) f32, dones (T,N) {0,1}, last_values (N,).
    Returns (returns, raw advantages = returns - values)."""
    T = rewards.shape[0]
    returns = torch.zeros_like(rewards)
    adv = torch.zeros_like(last_values)
    for t in reversed(range(T)):
        nxt = last_values if t == T - 1 else values[t + 1]
        not_term = 1.0 - dones[t].float()
        delta = rewards[t] + not_term * gamma * nxt - values[t]
        adv = delta + not_term * gamma * lam * adv
        returns[t] = adv + values[t]
    return returns, returns - values


def normalize_advantages(adv):
    """rollout_storage.py:135-136: unbiased std over all T*N samples."""
    return (adv - adv.mean()) / (adv.std() + 1e-8)


# ------------------------------------------------------------------------------------------------ A1/A2
class Params:
    """Flat list view of an ActorCritic: actor W/b x4, critic W/b x4, std (actor_critic.py:53-83)."""

    def __init__(self, actor, critic, std):
        self.actor = actor      # list of (W (out,in), b (out,))
        self.critic = critic
        self.std = std          # (12,)

    @staticmethod
    def from_npz(G, prefix):
        a = [(torch.from_numpy(G["%sactor_%d_weight" % (prefix, i)]).clone(),
              torch.from_numpy(G["%sactor_%d_bias" % (prefix, i)]).clone()) for i in (0, 2, 4, 6)]
        c = [(torch.from_numpy(G["%scritic_%d_weight" % (prefix, i)]).clone(),
              torch.from_numpy(G["%scritic_%d_bias" % (prefix, i)]).clone()) for i in (0, 2, 4, 6)]
        return Params(a, c, torch.from_numpy(G["%sstd" % prefix]).clone())

    @staticmethod
    def random(num_obs, num_priv, num_act, actor_hidden, critic_hidden, gen, init_std=1.0):
        """nn.Linear default init restated: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for W and b."""
        def mlp(dims):
            out = []
            for i in range(len(dims) - 1):
                k = 1.0 / math.sqrt(dims[i])
                out.append(((torch.rand(dims[i + 1], dims[i], generator=gen) * 2 - 1) * k,
                            (torch.rand(dims[i + 1], generator=gen) * 2 - 1) * k))
            return out
        return Params(mlp([num_obs] + list(actor_hidden) + [num_act]),
                      mlp([num_priv] + list(critic_hidden) + [1]), torch.full((num_act,), float(init_std)))

    def tensors(self):
        """state_dict order: std, actor.{0,2,4,6}.{weight,bias}, critic.{0,2,4,6}.{weight,bias}."""
        out = [self.std]
        for W, b in self.actor + self.critic:
            out += [W, b]
        return out

    def clone(self):
        return Params([(W.clone(), b.clone()) for W, b in self.actor],
                      [(W.clone(), b.clone()) for W, b in self.critic], self.std.clone())


def bf16_round(t):
    """Round-to-nearest-even to bfloat16 and back: the operand precision of the MFMA fast path."""
    return t.to(torch.bfloat16).to(torch.float32)


def mlp_forward(x, layers, keep=False, quant=None):
    """Linear -> ELU(alpha=1) -> ... -> Linear.  With keep=True also returns the layer inputs and
    pre-activations needed by mlp_backward.

    quant (e.g. bf16_round): the operand-precision model of the bf16 kernels -- NOT something the reference does.  Applied
    where csrc/hgym_fused.hpp rounds: the gathered input rows, every weight matrix, every hidden activation after the ELU;
    biases, accumulation and the head output stay fp32.  Lets a test separate "the kernel computes the bf16-operand network
    correctly" from "bf16 operands differ from fp32 ones"."""
    q = quant if quant is not None else (lambda t: t)
    h = q(x)
    acts, pres = [h], []
    for i, (W, b) in enumerate(layers):
        z = F.linear(h, q(W), b)
        if i < len(layers) - 1:
            pres.append(z)
            h = q(torch.where(z > 0, z, torch.exp(z) - 1.0))   # == F.elu(z) (expm1 on some builds; see test tol)
            acts.append(h)
        else:
            h = z
    return (h, acts, pres) if keep else h


def mlp_backward(dy, layers, acts, pres, quant=None):
    """Gradients of sum(dy * mlp(x)) w.r.t. every W, b.  acts[i] = input of layer i, pres[i] = its pre-activation.

    quant: as in mlp_forward -- every dZ is rounded where the kernels store it, elu' is taken from the ROUNDED activation y
    (y > 0 ? 1 : y + 1, what mlp_bwd_kernel does), the head's bias gradient is summed before the rounding (the loss kernel's
    fp32 partials), the hidden ones after it (column sums of the stored dZ)."""
    q = quant
    grads = [None] * len(layers)
    g = dy
    for i in reversed(range(len(layers))):
        W, _ = layers[i]
        if q is None:
            grads[i] = (g.t() @ acts[i], g.sum(dim=0))
        else:
            gb = g.sum(dim=0) if i == len(layers) - 1 else None
            g = q(g)
            grads[i] = (g.t() @ acts[i], gb if gb is not None else g.sum(dim=0))
        if i > 0:
            g = g @ (W if q is None else q(W))
            if q is None:
                z = pres[i - 1]
                g = g * torch.where(z > 0, torch.ones_like(z), torch.exp(z))
            else:
                y = acts[i]
                g = g * torch.where(y > 0, torch.ones_like(y), y + 1.0)
    return grads


def gaussian_log_prob(a, mu, sigma):
    """Normal(mu, sigma).log_prob(a).sum(-1)  (actor_critic.py:120)."""
    return (-((a - mu) ** 2) / (2 * sigma ** 2) - torch.log(sigma) - HALF_LOG_2PI).sum(dim=-1)


def gaussian_entropy(sigma):
    """Normal.entropy().sum(-1)  (actor_critic.py:109)."""
    return (0.5 + HALF_LOG_2PI + torch.log(sigma)).sum(dim=-1)


def policy_act(p, obs, priv, z):
    """ppo.py:91-101 with the standard-normal draw z (N,12) supplied: a = mu + sigma*z."""
    mu = mlp_forward(obs, p.actor)
    sigma = mu * 0.0 + p.std
    a = mu + sigma * z
    v = mlp_forward(priv, p.critic)
    return a, v, gaussian_log_prob(a, mu, sigma), mu, sigma


def bootstrap_rewards(rewards, values, time_outs, gamma):
    """ppo.py:107-108: r += gamma * V * time_outs."""
    return rewards + gamma * torch.squeeze(values * time_outs.unsqueeze(1), 1)


# ------------------------------------------------------------------------------------------------ A11
def ppo_loss_and_grads(p, obs, priv, actions, old_values, adv, returns, old_logp, old_mu, old_sigma,
                       clip=0.2, value_coef=1.0, entropy_coef=0.001, quant=None):
    """One minibatch of ppo.py:128-168 + the hand-written backward of `loss`.

    Inputs are (B,*) with old_values/adv/returns/old_logp shaped (B,).  Returns
    dict(loss, surrogate, value_loss, entropy, kl, grads=Params-shaped gradients)."""
    B = obs.shape[0]
    mu, a_acts, a_pres = mlp_forward(obs, p.actor, keep=True, quant=quant)
    sigma = mu * 0.0 + p.std
    logp = gaussian_log_prob(actions, mu, sigma)
    val, c_acts, c_pres = mlp_forward(priv, p.critic, keep=True, quant=quant)
    val = val.squeeze(-1)
    ent = gaussian_entropy(sigma)
    kl = torch.sum(torch.log(sigma / old_sigma + 1.e-5)
                   + (torch.square(old_sigma) + torch.square(old_mu - mu)) / (2.0 * torch.square(sigma)) - 0.5,
                   dim=-1).mean()
    ratio = torch.exp(logp - old_logp)
    s1 = -adv * ratio
    s2 = -adv * torch.clamp(ratio, 1.0 - clip, 1.0 + clip)
    surrogate = torch.max(s1, s2).mean()
    v_clipped = old_values + (val - old_values).clamp(-clip, clip)
    l1 = (val - returns).pow(2)
    l2 = (v_clipped - returns).pow(2)
    value_loss = torch.max(l1, l2).mean()
    loss = surrogate + value_coef * value_loss - entropy_coef * ent.mean()
    # ---- backward, by hand
    in_range = (ratio >= 1.0 - clip) & (ratio <= 1.0 + clip)
    w1 = torch.where(s1 > s2, torch.ones_like(s1), torch.where(s1 == s2, torch.full_like(s1, 0.5), torch.zeros_like(s1)))
    w2 = 1.0 - w1
    d_ratio = (-adv) * (w1 + w2 * in_range) / B
    d_logp = d_ratio * ratio
    diff = actions - mu
    d_mu = d_logp.unsqueeze(1) * diff / sigma ** 2
    d_sigma = d_logp.unsqueeze(1) * (diff ** 2 / sigma ** 3 - 1.0 / sigma) - (entropy_coef / B) / sigma
    d_std = d_sigma.sum(dim=0)
    v_in = ((val - old_values) >= -clip) & ((val - old_values) <= clip)
    u1 = torch.where(l1 > l2, torch.ones_like(l1), torch.where(l1 == l2, torch.full_like(l1, 0.5), torch.zeros_like(l1)))
    d_val = value_coef / B * (u1 * 2 * (val - returns) + (1.0 - u1) * 2 * (v_clipped - returns) * v_in)
    ga = mlp_backward(d_mu, p.actor, a_acts, a_pres, quant=quant)
    gc = mlp_backward(d_val.unsqueeze(1), p.critic, c_acts, c_pres, quant=quant)
    return dict(loss=loss, surrogate=surrogate, value_loss=value_loss, entropy=ent.mean(), kl=kl,
                grads=Params(ga, gc, d_std), d_mu=d_mu, d_val=d_val, mu=mu, val=val, logp=logp)


def adapt_lr(lr, kl, desired_kl=0.01):
    """ppo.py:142-145 (python-double learning rate)."""
    kl = float(kl)
    if kl > desired_kl * 2.0:
        return max(1e-5, lr / 1.5)
    if kl < desired_kl / 2.0 and kl > 0.0:
        return min(1e-2, lr * 1.5)
    return lr


def clip_grad_norm(grads, max_norm):
    """nn.utils.clip_grad_norm_ (ppo.py:173): norm of per-tensor norms, coef clamped to 1."""
    ts = grads.tensors()
    total = torch.norm(torch.stack([torch.norm(g) for g in ts]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in ts:
        g.mul_(coef)
    return total


class Adam:
    """torch.optim.Adam defaults restated (betas .9/.999, eps 1e-8, no weight decay, no amsgrad)."""

    def __init__(self, params):
        self.m = [torch.zeros_like(t) for t in params.tensors()]
        self.v = [torch.zeros_like(t) for t in params.tensors()]
        self.t = 0

    def step(self, params, grads, lr, b1=0.9, b2=0.999, eps=1e-8):
        self.t += 1
        bc1 = 1 - b1 ** self.t
        bc2 = 1 - b2 ** self.t
        step_size = lr / bc1
        for p, g, m, v in zip(params.tensors(), grads.tensors(), self.m, self.v):
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
            p.addcdiv_(m, denom, value=-step_size)


def ppo_update(p, opt, storage, perm, lr, epochs=2, minibatches=4, clip=0.2, value_coef=1.0, entropy_coef=0.001,
               max_grad_norm=1.0, desired_kl=0.01, adaptive=True, trace=None):
    """ppo.py:119-184 over a filled storage dict of (T,N,*) tensors (time-major, flattened t*N+n,
    rollout_storage.py:151-182); `perm` is the one permutation shared by every epoch."""
    flat = {k: v.flatten(0, 1) for k, v in storage.items()}
    n = perm.numel()
    mb = n // minibatches
    sum_v = sum_s = 0.0
    for _ in range(epochs):
        for i in range(minibatches):
            idx = perm[i * mb:(i + 1) * mb]
            out = ppo_loss_and_grads(p, flat["obs"][idx], flat["priv"][idx], flat["actions"][idx],
                                     flat["values"][idx].squeeze(-1), flat["advantages"][idx].squeeze(-1),
                                     flat["returns"][idx].squeeze(-1), flat["logp"][idx].squeeze(-1),
                                     flat["mu"][idx], flat["sigma"][idx], clip, value_coef, entropy_coef)
            if adaptive:
                lr = adapt_lr(lr, out["kl"], desired_kl)
            clip_grad_norm(out["grads"], max_grad_norm)
            if trace is not None:
                trace.append(dict(lr=lr, kl=float(out["kl"]), grads=out["grads"],
                                  value_loss=float(out["value_loss"]), surrogate=float(out["surrogate"])))
            opt.step(p, out["grads"], lr)
            sum_v += float(out["value_loss"])
            sum_s += float(out["surrogate"])
    k = epochs * minibatches
    return lr, sum_v / k, sum_s / k


# ----------------------------------------------------------------------------------------------
# The product's minibatch permutation (hgym_randperm).  The reference draws torch.randperm(T*N) (rollout_storage.py:149); any
# shuffle serves the update, and the product's is a keyed bijection evaluated per index.  This is its restatement for the
# tests (numpy, vectorised over the indices): 6-round balanced Feistel over 2^(2*half_bits) >= n, cycle-walked into [0, n).
def _perm_mix(x):
    import numpy as np
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x7feb352d)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x * np.uint32(0x846ca68b)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def feistel_permutation(n, seed, draw):
    import numpy as np
    M64 = (1 << 64) - 1
    half_bits = 1
    while (1 << (2 * half_bits)) < n:
        half_bits += 1
    key = (((seed ^ ((draw * 0x9e3779b97f4a7c15) & M64)) * 0xd1342543de82ef95) + draw) & M64
    k0, k1 = np.uint32(key & 0xFFFFFFFF), np.uint32(key >> 32)
    mask = np.uint32((1 << half_bits) - 1)
    rk = [_perm_mix(np.array([(int(k0) + r * 0x9e3779b9) & 0xFFFFFFFF], dtype=np.uint32))[0] for r in range(6)]
    x = np.arange(n, dtype=np.uint64)
    todo = np.ones(n, dtype=bool)
    with np.errstate(over="ignore"):
        while todo.any():
            v = x[todo]
            l = (v >> np.uint64(half_bits)).astype(np.uint32) & mask
            r = v.astype(np.uint32) & mask
            for q in range(6):
                f = _perm_mix(r ^ rk[q] ^ k1) & mask
                l, r = r, l ^ f
            v = (l.astype(np.uint64) << np.uint64(half_bits)) | r.astype(np.uint64)
            x[todo] = v
            todo[todo] = v >= n
    return x.astype(np.int64)
