"""Time-major rollout buffer with the reference's fields and methods (algo/ppo/rollout_storage.py:36-182).

Observation buffers carry ONE extra time slot: the env writes the observations of step t+1 directly into slot
t+1 (LeggedRobot.bind_outputs), so nothing is copied per step; slot T holds the bootstrap observation and is
rotated to slot 0 when the buffer is cleared.  `observations` / `privileged_observations` are the (T, N, .) views.
GAE and advantage normalisation run in libhgym_hip.so (wavefront scan)."""
import ctypes as C

import torch


class RolloutStorage:
    class Transition:
        def __init__(self):
            self.observations = None
            self.critic_observations = None
            self.actions = None
            self.rewards = None
            self.dones = None
            self.values = None
            self.actions_log_prob = None
            self.action_mean = None
            self.action_sigma = None
            self.hidden_states = None

        def clear(self):
            self.__init__()

    def __init__(self, num_envs, num_transitions_per_env, obs_shape, privileged_obs_shape, actions_shape, device="cpu"):
        self.device = device
        self.obs_shape, self.privileged_obs_shape, self.actions_shape = obs_shape, privileged_obs_shape, actions_shape
        T, N = num_transitions_per_env, num_envs
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=self.device)
        self._obs_all = z(T + 1, N, *obs_shape)
        self.observations = self._obs_all[:T]
        if privileged_obs_shape[0] is not None:
            self._priv_all = z(T + 1, N, *privileged_obs_shape)
            self.privileged_observations = self._priv_all[:T]
        else:
            self._priv_all = None
            self.privileged_observations = None
        self.rewards = z(T, N, 1)
        self.actions = z(T, N, *actions_shape)
        self.dones = z(T, N, 1, dtype=torch.uint8)
        self.actions_log_prob = z(T, N, 1)
        self.values = z(T, N, 1)
        self.returns = z(T, N, 1)
        self.advantages = z(T, N, 1)
        self.mu = z(T, N, *actions_shape)
        self.sigma = z(T, N, *actions_shape)
        self.num_transitions_per_env, self.num_envs = T, N
        self.saved_hidden_states_a = self.saved_hidden_states_c = None
        self._stats = z(4 + 2 * ((N + 15) // 16), dtype=torch.float64)      # HGYM_GAE_STATS_DOUBLES(N): [sum, sum of squares, count | counter, partials]
        self.step = 0
        # optional bf16 shadows of the two observation buffers (enable_shadow): written by the policy launch that reads the slot,
        # read by the update instead of the fp32 rows
        self._obs_bf16 = self._priv_bf16 = None
        self.shadow_valid = [False] * T
        # deferred values (enable_deferred_values): the bootstrap flags of every step + the value of the bootstrap observation
        self.time_outs = self.last_values = None

    def enable_shadow(self, ld_obs, ld_priv):
        """Native extension: allocate (T, N, ld) bfloat16 copies of `observations` / `privileged_observations` (ld = the widths
        padded to 128, zero pad columns).  Slot s counts as valid once a policy launch has written it (shadow_slot)."""
        if self._priv_all is None or ld_obs <= 0 or ld_priv <= 0:
            return False
        T, N = self.num_transitions_per_env, self.num_envs
        self._obs_bf16 = torch.zeros(T, N, int(ld_obs), dtype=torch.bfloat16, device=self.device)
        self._priv_bf16 = torch.zeros(T, N, int(ld_priv), dtype=torch.bfloat16, device=self.device)
        return True

    def enable_deferred_values(self):
        """Native extension: columns for a rollout whose critic runs ONCE after collection (PPO.deferred_values): time_outs (T, N, 1)
        uint8 = the stale-by-design extras["time_outs"] the per-step bootstrap would have used (ppo.py:107-108), last_values (N, 1)."""
        if self.time_outs is None:
            T, N = self.num_transitions_per_env, self.num_envs
            self.time_outs = torch.zeros(T, N, 1, dtype=torch.uint8, device=self.device)
            self.last_values = torch.zeros(N, 1, dtype=torch.float32, device=self.device)

    def shadow_slot(self, s):
        """(obs_bf16[s], priv_bf16[s]) for the policy launch that reads slot s, which thereby becomes valid; None without shadows."""
        if self._obs_bf16 is None or s >= self.num_transitions_per_env:
            return None
        self.shadow_valid[s] = True
        return self._obs_bf16[s], self._priv_bf16[s]

    def shadows(self):
        """The flattened (T*N, ld) shadows when every slot of the current rollout was written by its policy launch, else None."""
        if self._obs_bf16 is None or not all(self.shadow_valid):
            return None
        return self._obs_bf16.flatten(0, 1), self._priv_bf16.flatten(0, 1)

    # ------------------------------------------------------------------
    def _same(self, a, b):
        return a.data_ptr() == b.data_ptr() and a.shape == b.shape

    def add_transitions(self, transition):
        if self.step >= self.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        s = self.step
        # (name, destination, source, does the slot's bf16 shadow describe this column?)
        pairs = [("observations", self.observations[s], transition.observations, True), ("actions", self.actions[s], transition.actions, False),
                 ("rewards", self.rewards[s], transition.rewards.view(-1, 1), False), ("dones", self.dones[s], transition.dones.view(-1, 1), False),
                 ("values", self.values[s], transition.values, False),
                 ("actions_log_prob", self.actions_log_prob[s], transition.actions_log_prob.view(-1, 1), False),
                 ("mu", self.mu[s], transition.action_mean, False), ("sigma", self.sigma[s], transition.action_sigma, False)]
        if self.privileged_observations is not None:
            pairs.append(("privileged_observations", self.privileged_observations[s], transition.critic_observations, True))
        for name, dst, src, shadowed in pairs:
            if not self._same(dst, src):          # producers that already wrote the slot are not copied again
                dst.copy_(src)
                if shadowed:                      # observations copied in from elsewhere: the slot's bf16 shadow does not describe them
                    self.shadow_valid[s] = False
        self.step += 1

    def check_shadows(self, rows=64):
        """Debug aid (HGYM_CHECK_SHADOW=1: PPO.update calls it before gathering from the shadows): a few rows of every slot of the
        bf16 shadows against bfloat16(fp32 rows) -- catches a caller that wrote `observations[s]` directly behind the policy launch,
        which the validity flags cannot see."""
        sh = self.shadows()
        if sh is None:
            return
        T, N = self.num_transitions_per_env, self.num_envs
        idx = torch.randint(0, T * N, (rows,), device=self.device)
        for name, full, bf in (("observations", self.observations, sh[0]), ("privileged_observations", self.privileged_observations, sh[1])):
            want = full.flatten(0, 1)[idx].to(torch.bfloat16)
            got = bf[idx][:, :want.shape[1]]
            if not torch.equal(want, got):
                raise AssertionError("bf16 shadow of %s is stale: a storage slot was written behind the policy launch that shadowed it" % name)

    def clear(self):
        self._obs_all[0].copy_(self._obs_all[self.num_transitions_per_env])
        if self._priv_all is not None:
            self._priv_all[0].copy_(self._priv_all[self.num_transitions_per_env])
        self.step = 0
        self.shadow_valid = [False] * self.num_transitions_per_env

    def compute_returns(self, last_values, gamma, lam, stats_hook=None, time_outs=None):
        """rollout_storage.py:122-136.  time_outs (native extension, deferred values): `rewards` holds RAW rewards; the time-out
        bootstrap r += gamma * V * time_outs is applied (and written back) as the scan loads them (hgym_gae_bootstrap)."""
        from hgym import _lib as L
        T, N = self.num_transitions_per_env, self.num_envs
        if not self.rewards.is_cuda:
            raise RuntimeError("RolloutStorage.compute_returns runs on the HIP path only (storage is on %s)" % self.device)
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        lv = last_values.reshape(N).contiguous()
        if time_outs is not None:
            L.check(L.lib.hgym_gae_bootstrap(T, N, L.fptr(self.rewards), L.fptr(self.values), L.u8ptr(self.dones), L.u8ptr(time_outs), L.fptr(lv),
                                             gamma, lam, L.fptr(self.returns), L.fptr(self.advantages), L.f64ptr(self._stats), s),
                    "hgym_gae_bootstrap")
        else:
            L.check(L.lib.hgym_gae(T, N, L.fptr(self.rewards), L.fptr(self.values), L.u8ptr(self.dones), L.fptr(lv), gamma, lam,
                                   L.fptr(self.returns), L.fptr(self.advantages), L.f64ptr(self._stats), s), "hgym_gae")
        if stats_hook is not None:
            stats_hook(self._stats[:3])           # multi-GPU: all-reduce (sum, sumsq, count) for a global normalisation
        L.check(L.lib.hgym_adv_normalize(T * N, L.fptr(self.advantages), L.f64ptr(self._stats), s), "hgym_adv_normalize")

    def get_statistics(self):
        done = self.dones.clone()
        done[-1] = 1
        flat = done.permute(1, 0, 2).reshape(-1, 1)
        idx = torch.cat((flat.new_tensor([-1], dtype=torch.int64), flat.nonzero(as_tuple=False)[:, 0]))
        return (idx[1:] - idx[:-1]).float().mean(), self.rewards.mean()

    def permutation(self, n, seed, draw):
        """The minibatch permutation (reference :149, torch.randperm) as one hgym_randperm launch: a bijection of [0, n) keyed
        by (seed, draw), written into a buffer this object keeps.  draw: an int, or a one-element int64 DEVICE tensor holding it
        (hgym_randperm_dev: the launch's arguments are then the same in every iteration -- what a captured update replays)."""
        from hgym import _lib as L
        if getattr(self, "_perm", None) is None or self._perm.numel() != n:
            self._perm = torch.empty(n, dtype=torch.int64, device=self.device)
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if torch.is_tensor(draw):
            assert draw.dtype == torch.int64 and draw.is_cuda and draw.numel() == 1
            L.check(L.lib.hgym_randperm_dev(n, int(seed) & 0xFFFFFFFFFFFFFFFF, L.i64ptr(draw), L.i64ptr(self._perm), s), "hgym_randperm_dev")
        else:
            L.check(L.lib.hgym_randperm(n, int(seed) & 0xFFFFFFFFFFFFFFFF, int(draw), L.i64ptr(self._perm), s), "hgym_randperm")
        return self._perm

    def mini_batch_generator(self, num_mini_batches, num_epochs=8):
        """API-compatible generator (gathers with torch); the native PPO.update does not use it -- it hands the
        index slices to hgym_ppo_grad, which fuses the gather into the operand packing."""
        batch = self.num_envs * self.num_transitions_per_env
        mb = batch // num_mini_batches
        indices = torch.randperm(num_mini_batches * mb, requires_grad=False, device=self.device)
        fl = lambda t: t.flatten(0, 1)
        obs = fl(self.observations)
        cobs = fl(self.privileged_observations) if self.privileged_observations is not None else obs
        cols = [fl(self.actions), fl(self.values), fl(self.advantages), fl(self.returns), fl(self.actions_log_prob), fl(self.mu),
                fl(self.sigma)]
        for _ in range(num_epochs):
            for i in range(num_mini_batches):
                idx = indices[i * mb:(i + 1) * mb]
                a, v, adv, ret, lp, mu, sg = (c[idx] for c in cols)
                yield obs[idx], cobs[idx], a, v, adv, ret, lp, mu, sg, (None, None), None
