"""Caller-owned buffers of the actor/critic + PPO optimiser and thin call wrappers over the C-ABI."""
import ctypes as C

import torch

from . import _lib as L


def make_net_config(num_obs, num_priv, num_actions, actor_hidden, critic_hidden, precision, max_batch, aux_hidden=None, aux_out=0,
                    aux_target_offset=0):
    """aux_hidden / aux_out / aux_target_offset: the optional auxiliary (denoising) head obs -> aux_hidden -> aux_out that regresses
    columns [aux_target_offset, aux_target_offset + aux_out) of the privileged row (HgymNetConfig.aux_*)."""
    c = L.NetConfig()
    c.num_obs, c.num_priv, c.num_actions = int(num_obs), int(num_priv), int(num_actions)
    ad = [num_obs] + list(actor_hidden) + [num_actions]
    cd = [num_priv] + list(critic_hidden) + [1]
    c.actor_layers, c.critic_layers = len(ad) - 1, len(cd) - 1
    for i, d in enumerate(ad):
        c.actor_dims[i] = int(d)
    for i, d in enumerate(cd):
        c.critic_dims[i] = int(d)
    c.precision = {"f32": L.F32, "fp32": L.F32, "bf16": L.BF16}[precision] if isinstance(precision, str) else int(precision)
    c.max_batch = int(max_batch)
    if aux_hidden is not None and aux_out > 0:
        xd = [num_obs] + list(aux_hidden) + [int(aux_out)]
        c.aux_layers = len(xd) - 1
        for i, d in enumerate(xd):
            c.aux_dims[i] = int(d)
        c.aux_target_offset = int(aux_target_offset)
    return c


def make_ppo_config(clip_param=0.2, value_loss_coef=1.0, entropy_coef=0.001, max_grad_norm=1.0, desired_kl=0.01,
                    adaptive=True, world_size=1, grad_norm_ready=False, aux_coef=0.0):
    p = L.PPOConfig()
    p.clip_param, p.value_loss_coef, p.entropy_coef = clip_param, value_loss_coef, entropy_coef
    p.max_grad_norm, p.desired_kl = max_grad_norm, desired_kl
    p.beta1, p.beta2, p.adam_eps = 0.9, 0.999, 1e-8
    p.lr_min, p.lr_max = 1e-5, 1e-2
    p.adaptive_lr = 1 if adaptive else 0
    p.world_size = int(world_size)
    p.aux_coef = float(aux_coef)
    p.grad_norm_ready = 1 if (grad_norm_ready and int(world_size) == 1) else 0    # see HgymPPOConfig
    return p


class NetBuffers:
    """Flat fp32 master parameters (state_dict order), Adam state, optimiser scalars and the zero-filled
    workspace; exposes per-tensor views named like the reference's ActorCritic.state_dict()."""

    def __init__(self, cfg, device, learning_rate=1e-5, grads_ext=None):
        """grads_ext: optional caller-owned (>= P + 1,) fp32 tensor to hold [gradient | KL] (the data-parallel update's direct exchange
        keeps it in peer-mapped memory: dist_utils.P2PComm)."""
        self.cfg = cfg
        self.device = torch.device(device)
        self.P = int(L.lib.hgym_net_param_count(C.byref(cfg)))
        nbytes = int(L.lib.hgym_net_workspace_bytes(C.byref(cfg)))
        if self.P <= 0 or nbytes <= 0:
            raise L.HgymError("bad net config: %s" % L.lib.hgym_last_error().decode())
        z = lambda n, dt=torch.float32: torch.zeros(n, dtype=dt, device=self.device)
        self.params, self.adam_m, self.adam_v = z(self.P), z(self.P), z(self.P)
        if grads_ext is not None:
            assert grads_ext.dtype == torch.float32 and grads_ext.is_contiguous() and grads_ext.numel() >= self.P + 1
            grads_ext.zero_()
        self.grads_ext = grads_ext[:self.P + 1] if grads_ext is not None else z(self.P + 1)      # flat gradient + the minibatch KL slot: what the ranks all-reduce, in one piece
        self.grads = self.grads_ext[:self.P]
        self.opt_state = z(16, torch.float64)
        self.opt_state[0] = learning_rate
        self.workspace = torch.zeros(nbytes + 256, dtype=torch.uint8, device=self.device)
        off = (-self.workspace.data_ptr()) % 256
        self._ws_ptr = self.workspace.data_ptr() + off
        self.struct = L.Net(L.fptr(self.params), L.fptr(self.grads), L.fptr(self.adam_m), L.fptr(self.adam_v),
                            L.f64ptr(self.opt_state), C.c_void_p(self._ws_ptr))
        # named views
        self.views = {}
        A = cfg.num_actions
        self.views["std"] = self.params[:A]
        off = A
        nets = [("actor", cfg.actor_dims, cfg.actor_layers), ("critic", cfg.critic_dims, cfg.critic_layers)]
        if cfg.aux_layers > 0:
            nets.append(("denoiser", cfg.aux_dims, cfg.aux_layers))
        for name, dims, n in nets:
            for l in range(n):
                k, o = dims[l], dims[l + 1]
                self.views["%s.%d.weight" % (name, 2 * l)] = self.params[off:off + o * k].view(o, k)
                off += o * k
                self.views["%s.%d.bias" % (name, 2 * l)] = self.params[off:off + o]
                off += o
        assert off == self.P

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream) if self.device.type == "cuda" else None

    def load_state_dict(self, sd):
        for k, v in self.views.items():
            v.copy_(torch.as_tensor(sd[k]).to(self.device).view_as(v))
        self.sync_shadow()

    def state_dict(self):
        return {k: v.detach().clone() for k, v in self.views.items()}

    def grad_views(self):
        out, base = {}, self.params.data_ptr()
        for k, v in self.views.items():
            o = (v.data_ptr() - base) // 4
            out[k] = self.grads[o:o + v.numel()].view_as(v)
        return out

    def sync_shadow(self):
        L.check(L.lib.hgym_net_sync_shadow(C.byref(self.cfg), C.byref(self.struct), self.stream()), "hgym_net_sync_shadow")

    def forward(self, which, x):
        M = x.shape[0]
        nout = self.cfg.num_actions if which == 0 else (1 if which == 1 else self.cfg.aux_dims[self.cfg.aux_layers])
        y = torch.empty(M, nout, device=self.device)
        L.check(L.lib.hgym_mlp_forward(C.byref(self.cfg), C.byref(self.struct), which, M, L.fptr(x), x.stride(0), L.fptr(y),
                                       self.stream()), "hgym_mlp_forward")
        return y

    def shadow_ld(self, which):
        """Leading dimension (elements) of the bf16 shadow of the actor's (0) / critic's (1) input rows; 0: no shadow on this path."""
        return int(L.lib.hgym_net_shadow_ld(C.byref(self.cfg), int(which)))

    def shadow_struct(self, obs_bf16, priv_bf16):
        """HgymObsShadow over two (M, ld) torch.bfloat16 tensors (the caller keeps them alive); either may be None (that member is
        not written by the launch)."""
        for t in (obs_bf16, priv_bf16):
            assert t is None or (t.dtype == torch.bfloat16 and t.is_contiguous())
        p = lambda t: (None, 0) if t is None else (C.c_void_p(t.data_ptr()), t.shape[-1])
        return L.ObsShadow(*p(obs_bf16), *p(priv_bf16))

    def critic_values(self, priv, values, priv_bf16=None):
        """hgym_critic_values: V of every row of `priv` ((M, num_priv) fp32, contiguous) into `values` ((M,) or (M, 1) fp32); priv_bf16:
        optional (M, ld) bfloat16 shadow rows to fill.  M may exceed the configuration's max_batch."""
        assert priv.is_contiguous() and values.is_contiguous() and priv.dtype == torch.float32 and values.numel() == priv.shape[0]
        sh = None if priv_bf16 is None else C.byref(self.shadow_struct(None, priv_bf16))
        L.check(L.lib.hgym_critic_values(C.byref(self.cfg), C.byref(self.struct), int(priv.shape[0]), L.fptr(priv), L.fptr(values), sh,
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)), "hgym_critic_values")

    def act(self, obs, priv, z=None, seed=0, step_counter=None, out=None, env_fin=None, shadow=None):
        """env_fin: optional (HgymEnvConfig, HgymEnvState, HgymEnvOut) of an env step whose finaliser was postponed
        (HgymEnvOut.defer_finalize): it runs as one extra workgroup of this launch (hgym_policy_act_fin).
        shadow: optional (obs_bf16, priv_bf16) tensors receiving the bf16 of the rows read (HgymObsShadow)."""
        sh = None if shadow is None else C.byref(self.shadow_struct(*shadow))
        M = obs.shape[0]
        A = self.cfg.num_actions
        if out is None:
            e = lambda *s: torch.empty(*s, device=self.device)
            out = dict(actions=e(M, A), mu=e(M, A), sigma=e(M, A), logp=e(M), values=e(M, 1))
        if env_fin is not None:
            ecfg, est, eout = env_fin
            L.check(L.lib.hgym_policy_act_fin(C.byref(self.cfg), C.byref(self.struct), M, L.fptr(obs), L.fptr(priv), L.fptr(z), int(seed),
                                              L.i64ptr(step_counter), L.fptr(out["actions"]), L.fptr(out["mu"]), L.fptr(out["sigma"]),
                                              L.fptr(out["logp"]), L.fptr(out["values"]), C.byref(ecfg), C.byref(est), C.byref(eout), sh,
                                              self.stream()), "hgym_policy_act_fin")
            return out
        L.check(L.lib.hgym_policy_act(C.byref(self.cfg), C.byref(self.struct), M, L.fptr(obs), L.fptr(priv), L.fptr(z), int(seed),
                                      L.i64ptr(step_counter), L.fptr(out["actions"]), L.fptr(out["mu"]), L.fptr(out["sigma"]),
                                      L.fptr(out["logp"]), L.fptr(out["values"]), sh, self.stream()), "hgym_policy_act")
        return out

    def ppo_grad(self, ppo, batch):
        L.check(L.lib.hgym_ppo_grad(C.byref(self.cfg), C.byref(ppo), C.byref(self.struct), C.byref(batch), self.stream()), "hgym_ppo_grad")

    def ppo_grad_part(self, ppo, batch, part):
        """hgym_ppo_grad in two halves (data-parallel update): after part 0 `grads_ext[:bucket_split]` (std | actor, the larger
        bucket) is final, after part 1 `grads_ext[bucket_split:]` (critic | auxiliary head | KL slot)."""
        L.check(L.lib.hgym_ppo_grad_part(C.byref(self.cfg), C.byref(ppo), C.byref(self.struct), C.byref(batch), int(part), self.stream()),
                "hgym_ppo_grad_part")

    @property
    def bucket_split(self):
        """Offset of the critic's first parameter in the flat vector: the boundary of the two gradient buckets."""
        return int(L.lib.hgym_net_param_offset(C.byref(self.cfg), 1))

    def ppo_apply(self, ppo):
        L.check(L.lib.hgym_ppo_apply(C.byref(self.cfg), C.byref(ppo), C.byref(self.struct), self.stream()), "hgym_ppo_apply")


def make_batch(obs, priv, actions, values, advantages, returns, logp, mu, sigma, idx, obs_bf16=None, priv_bf16=None):
    """All (T*N, *) flattened, contiguous fp32; idx int64 (B,).  obs_bf16 / priv_bf16: optional (T*N, ld) bfloat16 shadows of obs /
    priv (ld = NetBuffers.shadow_ld), both or neither."""
    for t in (obs, priv, actions, values, advantages, returns, logp, mu, sigma):
        assert t.is_contiguous() and t.dtype == torch.float32
    assert idx.dtype == torch.int64 and idx.is_contiguous()
    assert (obs_bf16 is None) == (priv_bf16 is None)
    sb = (None, None)
    if obs_bf16 is not None:
        assert obs_bf16.dtype == torch.bfloat16 and priv_bf16.dtype == torch.bfloat16 and obs_bf16.is_contiguous() and priv_bf16.is_contiguous()
        assert obs_bf16.shape[0] == obs.shape[0] and priv_bf16.shape[0] == priv.shape[0]
        sb = (C.c_void_p(obs_bf16.data_ptr()), C.c_void_p(priv_bf16.data_ptr()))
    return L.Batch(L.fptr(obs), L.fptr(priv), L.fptr(actions), L.fptr(values), L.fptr(advantages), L.fptr(returns), L.fptr(logp),
                   L.fptr(mu), L.fptr(sigma), L.i64ptr(idx), int(idx.numel()), sb[0], sb[1], int(obs.shape[0]))
