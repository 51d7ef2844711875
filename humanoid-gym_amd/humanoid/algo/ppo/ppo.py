"""PPO with the reference's interface (algo/ppo/ppo.py:38-184) on the MI355X hot path.

act() = one hgym_policy_act call whose outputs land directly in the rollout-storage slot; process_env_step() =
hgym_store_step (time-out bootstrap) ; compute_returns() = wavefront-scan GAE ; update() = for every minibatch
hgym_ppo_grad (gather + forward + KL + loss + hand-written backward, all on the device, no host sync) ->
[when torch.distributed is initialised: one RCCL all-reduce of [flat gradient | minibatch KL]] -> hgym_ppo_apply (adaptive-KL
learning rate, grad-norm clip, Adam, operand-shadow refresh).  The host reads the loss sums back once per update.
"""
import os

import torch
import torch.nn as nn  # noqa: F401  (kept importable like the reference module)

from .actor_critic import ActorCritic
from .rollout_storage import RolloutStorage
from . import dist_utils


class _DeviceAdam:
    """What `alg.optimizer` exposes to OnPolicyRunner.save/load: torch.optim.Adam-shaped state dicts backed by the
    flat exp_avg / exp_avg_sq vectors the HIP Adam kernel updates."""

    # the hyper-parameters the HIP Adam kernel runs with (hgym.make_ppo_config); ONE definition for param_groups and for the
    # runner's background checkpoint writer, which must not read the learning rate from the device (a host sync)
    HYPER = dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False)

    def __init__(self, ppo):
        self._ppo = ppo

    @property
    def param_groups(self):
        return [dict(lr=self._ppo.learning_rate, **self.HYPER)]

    def state_dict(self):
        net = self._ppo.net
        state = {}
        if net is not None:
            step = float(net.opt_state[1])
            base = net.params.data_ptr()
            for i, (name, v) in enumerate(net.views.items()):
                o = (v.data_ptr() - base) // 4
                state[i] = dict(step=torch.tensor(step), exp_avg=net.adam_m[o:o + v.numel()].view_as(v).clone(),
                                exp_avg_sq=net.adam_v[o:o + v.numel()].view_as(v).clone())
        n = len(state)
        return dict(state=state, param_groups=[dict(self.param_groups[0], params=list(range(n)))])

    def load_state_dict(self, sd):
        net = self._ppo.net
        base = net.params.data_ptr()
        for i, (name, v) in enumerate(net.views.items()):
            if i in sd["state"]:
                o = (v.data_ptr() - base) // 4
                net.adam_m[o:o + v.numel()].copy_(sd["state"][i]["exp_avg"].flatten())
                net.adam_v[o:o + v.numel()].copy_(sd["state"][i]["exp_avg_sq"].flatten())
                net.opt_state[1] = float(sd["state"][i]["step"])
        if sd.get("param_groups"):
            self._ppo.learning_rate = sd["param_groups"][0]["lr"]

    def zero_grad(self):
        if self._ppo.net is not None:
            self._ppo.net.grads.zero_()


class PPO:
    actor_critic: ActorCritic
    # minibatch permutation: "device" = hgym_randperm (keyed bijection, one launch, reproducible from the seed);
    # "torch" = torch.randperm from torch's global generator, as the reference draws it (a device sort: 125 us per iteration)
    permutation = "device"
    # compute precision of the dense layers: "bf16" (MFMA fast path, BASELINE config) or "f32" (parity mode)
    precision = os.environ.get("HGYM_PRECISION", "bf16")

    def __init__(self, actor_critic, num_learning_epochs=1, num_mini_batches=1, clip_param=0.2, gamma=0.998, lam=0.95,
                 value_loss_coef=1.0, entropy_coef=0.0, learning_rate=1e-3, max_grad_norm=1.0, use_clipped_value_loss=True,
                 schedule="fixed", desired_kl=0.01, device="cpu", denoise_coef=0.0):
        if not (str(device).startswith("cuda") and torch.cuda.is_available()):
            raise RuntimeError("PPO runs on the MI355X hot path only (device=%r); there is no CPU training path" % (device,))
        if not use_clipped_value_loss:
            raise NotImplementedError("the loss kernel implements the clipped value loss (reference default)")
        self.device = device
        self.desired_kl, self.schedule = desired_kl, schedule
        self._lr0 = learning_rate
        self.actor_critic = actor_critic
        self.actor_critic.to(self.device)
        self.storage = None
        self.net = None
        self.optimizer = _DeviceAdam(self)
        self.transition = RolloutStorage.Transition()
        self.clip_param = clip_param
        self.num_learning_epochs, self.num_mini_batches = num_learning_epochs, num_mini_batches
        self.value_loss_coef, self.entropy_coef = value_loss_coef, entropy_coef
        self.gamma, self.lam = gamma, lam
        self.max_grad_norm = max_grad_norm
        self.use_clipped_value_loss = use_clipped_value_loss
        self.denoise_coef = float(denoise_coef)     # weight of the ActorCritic denoiser head's MSE (0: head absent / not trained)
        # True while a runner has the env store the scalar columns and bump the sampling step itself (transition_sink)
        self.env_stores_transitions = False
        self.comm_timing = None      # a list: update() appends a pair of HIP events around the wait for the gradient exchange
        self._world = 1
        self._rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self._world, self._rank = torch.distributed.get_world_size(), torch.distributed.get_rank()
            if not dist_utils.active():      # HGYM_DIST_OFF=1: this rank of a running group trains ALONE (bench.py's same-box N = 1 reference) --
                self._world = 1              # no exchange, so apply must not form a mean over ranks either

    # ------------------------------------------------------------------
    @property
    def learning_rate(self):
        return float(self.net.opt_state[0]) if self.net is not None else self._lr0

    @learning_rate.setter
    def learning_rate(self, v):
        self._lr0 = v
        if self.net is not None:
            self.net.opt_state[0] = v

    def init_storage(self, num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape):
        import hgym
        self.storage = RolloutStorage(num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape, self.device)
        ac = self.actor_critic
        mb = (num_envs * num_transitions_per_env) // self.num_mini_batches
        aux = getattr(ac, "denoiser_hidden_dims", None)
        cfg = hgym.make_net_config(ac.num_actor_obs, ac.num_critic_obs, ac.num_actions, ac.actor_hidden_dims, ac.critic_hidden_dims,
                                   self.precision, max(mb, num_envs), aux_hidden=aux, aux_out=getattr(ac, "denoiser_targets", 0),
                                   aux_target_offset=ac.num_critic_obs - getattr(ac, "denoiser_targets", 0))
        # data-parallel update: the exchange is chosen here, once (HGYM_COMM=auto: the direct kernel over peer mappings is set up,
        # checked and timed against the collective, and used when it is faster; any failure falls back on every rank -- dist_utils).
        # With the direct exchange the gradient vector lives in peer-mapped memory.
        import ctypes as C
        self._comm = dist_utils.make_comm(int(hgym._lib.lib.hgym_net_param_count(C.byref(cfg))) + 1, self.device)
        self._comm_p2p = self._comm is not None and dist_utils.comm_backend() == "p2p"
        self._comm_pin = None
        self.comm_report = dist_utils.comm_report()     # mode, used, fallback_reason, probe timings (bench.py prints it)
        self._comm_direct_used = False      # the direct kernel has carried at least one real gradient (also under HGYM_COMM=both)
        self.comm_flip = False      # bench.py: use the OTHER exchange for the next update() (timing both in its profiling iterations)
        self.net = hgym.NetBuffers(cfg, self.device, learning_rate=self._lr0, grads_ext=None if self._comm is None else self._comm.data)
        dist_utils.broadcast_parameters(ac.parameters())   # identical initial parameters on every rank
        ac.bind(self.net)
        # bf16 shadows of the stored observation rows: the policy launches leave them behind, the update gathers from them
        # (environment knobs are read ONCE, here: a captured rollout graph keeps whatever was active at capture)
        self._check_shadow = os.environ.get("HGYM_CHECK_SHADOW", "0") == "1"
        if os.environ.get("HGYM_SHADOW", "1") != "0":
            self.storage.enable_shadow(self.net.shadow_ld(0), self.net.shadow_ld(1))
        self._ppo_cfg = hgym.make_ppo_config(self.clip_param, self.value_loss_coef, self.entropy_coef, self.max_grad_norm,
                                             self.desired_kl if self.desired_kl is not None else 0.0,
                                             adaptive=(self.desired_kl is not None and self.schedule == "adaptive"),
                                             world_size=self._world,
                                             grad_norm_ready=True,   # update() applies exactly what hgym_ppo_grad produced
                                             aux_coef=self.denoise_coef if aux else 0.0)
        self.last_denoise_loss = None
        self._sample_step = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._perm_seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + 0xABCD + 104729 * self._rank) & 0xFFFFFFFFFFFFFFFF
        self._perm_draws = 0
        self._perm_draws_dev = torch.zeros(1, dtype=torch.int64, device=self.device)      # the same number where a captured update reads it
        ac._sample_step = self._sample_step
        # exploration-noise key: from the run's seed (as the permutation key above) and the rank
        ac._sample_seed = (torch.initial_seed() * 0xD1342543DE82EF95 + 0x5EED + 7919 * self._rank) & 0xFFFFFFFFFFFFFFFF
        self._hgym = hgym

    def seek(self, iteration, steps_per_iteration):
        """Position the device generators' counters where a run that has done `iteration` learning iterations has them
        (OnPolicyRunner.load): a resumed run continues the exploration-noise and permutation streams instead of replaying them
        from the start.  The reference's checkpoint carries no generator state (on_policy_runner.py:274-281); the counters are
        functions of the iteration number, so neither does this one.  (The env's own draw streams are positioned by
        LeggedRobot.seek, which the runner calls next to this.)"""
        self._sample_step.fill_(int(iteration) * int(steps_per_iteration))
        self._perm_draws = int(iteration)
        with torch.inference_mode():
            self._perm_draws_dev.fill_(int(iteration))

    def check_comm(self, words=None):
        """The direct exchange's status word, read where the host synchronises anyway (the end of a synchronous update, the end of
        learn(), the asynchronous log's snapshot): raises dist_utils.CommTimeout if a bounded wait of any call expired -- that
        minibatch's gradient was garbage and the replicas have diverged, so training must not go on silently.  words: a host copy
        of the status block taken earlier (comm_status_snapshot); None: synchronise and read it now."""
        if self._comm is None or not (self._comm_p2p or self._comm_direct_used):      # (HGYM_COMM=both: bench.py's comm_flip updates ran it)
            return
        self._comm.raise_if_expired(self._comm.read_status() if words is None else words)

    def comm_status_snapshot(self, slot):
        """Stream-ordered device -> pinned-host copy of the status block (no host wait); the caller reads it after its own event."""
        if self._comm is None or not (self._comm_p2p or self._comm_direct_used):
            return None
        if self._comm_pin is None:
            self._comm_pin = [torch.zeros(16, dtype=torch.int64).pin_memory() for _ in range(2)]
        self._comm_pin[slot].copy_(self._comm.status, non_blocking=True)
        return self._comm_pin[slot]

    def test_mode(self):
        self.actor_critic.eval()

    def train_mode(self):
        self.actor_critic.train()

    # ------------------------------------------------------------------ rollout
    def act(self, obs, critic_obs, env_fin=None):
        """env_fin (native extension): a postponed env-step finaliser to run inside this policy launch
        (LeggedRobot.take_pending_finalize)."""
        st, s = self.storage, self.storage.step
        out = None
        if s < st.num_transitions_per_env:     # write straight into the storage slot (no add_transitions copies)
            out = dict(actions=st.actions[s], mu=st.mu[s], sigma=st.sigma[s], logp=st.actions_log_prob[s].view(-1), values=st.values[s])
        t = self.transition
        # the slot's bf16 shadow only when `obs` IS the slot (the zero-copy runner): a copy made later by add_transitions
        # invalidates it again
        sh = st.shadow_slot(s) if (out is not None and obs.data_ptr() == st._obs_all[s].data_ptr()
                                   and st._priv_all is not None and critic_obs.data_ptr() == st._priv_all[s].data_ptr()) else None
        t.actions = self.actor_critic.act(obs, critic_obs, out=out, env_fin=env_fin, shadow=sh)
        last = self.actor_critic._last
        t.values, t.actions_log_prob = last["values"], last["logp"]
        t.action_mean, t.action_sigma = last["mu"], last["sigma"]
        t.observations, t.critic_observations = obs, critic_obs
        if not self.env_stores_transitions:
            self._sample_step += 1
        return t.actions

    def fused_rollout_step(self, env, i, obs, critic_obs, next_obs, next_critic_obs, ahead=None, deferred=False):
        """act() + env.step() + process_env_step() of rollout step i as ONE launch (LeggedRobot.rollout_step): the policy's outputs
        land in storage slot i, the env writes the next observations into the slots handed in, the finaliser riding in the
        following launch stores rewards / dones of slot i (time-out bootstrap included).  ahead: (obs, critic_obs) of the slot after
        next_obs, whose older frames this launch may write ahead (LeggedRobot.rollout_step).
        deferred: the launch has no critic tiles; slot i receives the RAW reward and the bootstrap's time-out flags, and
        deferred_values() -- once, after the last step -- fills storage.values and applies the bootstrap in compute_returns."""
        st, s = self.storage, self.storage.step
        if s >= st.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        if deferred:
            st.enable_deferred_values()
            out = dict(actions=st.actions[s], mu=st.mu[s], sigma=st.sigma[s], logp=st.actions_log_prob[s].view(-1), values=None)
            sink = dict(values=None, rewards=st.rewards[s], dones=st.dones[s], time_outs=st.time_outs[s], step=self._sample_step, gamma=self.gamma)
            own = obs.data_ptr() == st._obs_all[s].data_ptr() and critic_obs.data_ptr() == st._priv_all[s].data_ptr()
            sh = (st._obs_bf16[s], None) if (own and st._obs_bf16 is not None) else None      # (the priv shadow: deferred_values)
            self._deferred_shadow = bool(own and st._obs_bf16 is not None and (s == 0 or getattr(self, "_deferred_shadow", False)))
            env.rollout_step(self.net, i, obs, critic_obs, next_obs, next_critic_obs, sink, self.actor_critic._sample_seed, out, shadow=sh)
            st.step += 1
            return
        out = dict(actions=st.actions[s], mu=st.mu[s], sigma=st.sigma[s], logp=st.actions_log_prob[s].view(-1), values=st.values[s])
        sink = dict(values=st.values[s], rewards=st.rewards[s], dones=st.dones[s], step=self._sample_step, gamma=self.gamma)
        sh = st.shadow_slot(s) if (obs.data_ptr() == st._obs_all[s].data_ptr() and critic_obs.data_ptr() == st._priv_all[s].data_ptr()) else None
        # the bf16 shadow of the NEXT slot's observation rows: this launch writes the columns it already knows (the carried first layer)
        sh_next = None
        if (sh is not None and s + 1 < st.num_transitions_per_env and next_obs.data_ptr() == st._obs_all[s + 1].data_ptr()):
            sh_next = st._obs_bf16[s + 1]
        env.rollout_step(self.net, i, obs, critic_obs, next_obs, next_critic_obs, sink, self.actor_critic._sample_seed, out, shadow=sh,
                         ahead=ahead, shadow_next=sh_next)
        st.step += 1

    def transition_sink(self):
        """The scalar columns of the storage slot act() has just filled, for an env that can store them itself
        (LeggedRobot.bind_transition); the caller sets `env_stores_transitions` for the duration and follows env.step
        with process_env_step(..., stored=True)."""
        st, s = self.storage, self.storage.step
        return dict(values=st.values[s], rewards=st.rewards[s], dones=st.dones[s], step=self._sample_step, gamma=self.gamma)

    def process_env_step(self, rewards, dones, infos, stored=False):
        """`stored=True`: the env already wrote this step's rewards / dones slot (transition_sink)."""
        import ctypes as C
        L = self._hgym._lib
        st, s = self.storage, self.storage.step
        if s >= st.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        if not stored:
            to = infos.get("time_outs") if isinstance(infos, dict) else None
            d8 = dones if dones.dtype == torch.uint8 else dones.view(torch.uint8) if dones.dtype == torch.bool else dones.to(torch.uint8)
            t8 = None if to is None else (to.view(torch.uint8) if to.dtype == torch.bool else to.to(torch.uint8))
            L.check(L.lib.hgym_store_step(st.num_envs, L.fptr(rewards), L.fptr(self.transition.values), L.u8ptr(t8), L.u8ptr(d8),
                                          self.gamma, L.fptr(st.rewards[s]), L.u8ptr(st.dones[s]),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)), "hgym_store_step")
        self.transition.rewards = st.rewards[s].view(-1)
        self.transition.dones = st.dones[s].view(-1)
        st.add_transitions(self.transition)
        self.transition.clear()
        self.actor_critic.reset(dones)

    def deferred_values(self):
        """After a rollout of fused_rollout_step(deferred=True) launches: ActorCritic.evaluate over EVERY stored privileged row in one
        pass (the T slots -> storage.values, the bootstrap observation in slot T -> storage.last_values), 64-row tiles at the update's
        efficiency instead of T + 1 latency-bound launches; the tiles also leave the bf16 shadow of the rows they read.  The
        reference evaluates V(s_t) inside PPO.act (ppo.py:96, on_policy_runner.py:129) -- same weights (they do not change during
        collection), same rows, same kernel arithmetic."""
        st = self.storage
        T, N = st.num_transitions_per_env, st.num_envs
        shadow = st._priv_bf16.flatten(0, 1) if (st._priv_bf16 is not None and getattr(self, "_deferred_shadow", False)) else None
        self.net.critic_values(st._priv_all[:T].flatten(0, 1), st.values.view(-1), shadow)
        self.net.critic_values(st._priv_all[T], st.last_values.view(-1))
        if shadow is not None:
            st.shadow_valid = [True] * T
        self._deferred_ready = True

    def _adv_stats(self, stats):
        """(sum adv, sum adv^2, count) of this rank's shard -> of the global batch, in place: through the direct exchange's mappings when that
        is what the gradients use (hgym_comm_sum64: rank-ordered fp64 sum, no torch.distributed call -- the update stays capturable), else
        the collective."""
        if dist_utils.active() and self._comm is not None and self._comm_p2p and not self.comm_flip:
            self._comm.sum64(stats)
            return stats
        return dist_utils.allreduce_adv_stats(stats)

    def compute_returns(self, last_critic_obs):
        if getattr(self, "_deferred_ready", False):       # deferred_values() has evaluated the bootstrap observation with the rest
            self._deferred_ready = False
            self.storage.compute_returns(self.storage.last_values, self.gamma, self.lam, stats_hook=self._adv_stats,
                                         time_outs=self.storage.time_outs)
            return
        last_values = self.actor_critic.evaluate(last_critic_obs)
        self.storage.compute_returns(last_values, self.gamma, self.lam, stats_hook=self._adv_stats)

    # ------------------------------------------------------------------ update
    def update_capturable(self):
        """True when compute_returns() + update(sync=False) enqueue the same launches with the same arguments in every iteration, i.e. may
        be captured into a HIP graph and replayed (OnPolicyRunner): the device permutation (its draw number is read on the device), no
        timing probes inside the update, and either one rank or the direct gradient exchange with its call number on the device -- a
        torch.distributed collective (RCCL / gloo) stays eager."""
        if self.permutation != "device" or self.comm_timing is not None or self._check_shadow:
            return False
        if not dist_utils.active():
            return True
        return bool(self._comm is not None and self._comm_p2p and not self.comm_flip and getattr(self._comm, "capturable", False))

    def update_graph_key(self):
        """Everything host-side that compute_returns() / update() bake into their launches: a change re-captures."""
        import ctypes as C
        return (bytes(C.string_at(C.addressof(self._ppo_cfg), C.sizeof(self._ppo_cfg))), self.num_learning_epochs, self.num_mini_batches,
                float(self.gamma), float(self.lam), self.permutation, self._perm_seed, id(self.storage), id(self.net),
                bool(dist_utils.active()), bool(self._comm_p2p))

    def after_update_replay(self):
        """Host-side book-keeping of one replayed compute_returns() + update(sync=False): what the Python of those two calls changes on the
        host besides enqueueing work (kept next to update() so that the two stay in step)."""
        st = self.storage
        if self.permutation == "device":
            self._perm_draws += 1
        self._deferred_ready = False
        st.step = 0
        st.shadow_valid = [False] * st.num_transitions_per_env
        if dist_utils.active() and self._comm is not None and self._comm_p2p:
            self._comm_direct_used = True
            self._comm.seq += self.num_learning_epochs * self.num_mini_batches

    def update(self, sync=True):
        """ppo.py:119-184.  Returns (mean_value_loss, mean_surrogate_loss) as floats -- the one host read-back of the
        update.  sync=False (native extension, used by the runner when nothing is logged): no read-back, returns
        (None, None), so the host can already enqueue the next rollout while this update runs."""
        hgym, net, st = self._hgym, self.net, self.storage
        T, N = st.num_transitions_per_env, st.num_envs
        batch = T * N
        mb = batch // self.num_mini_batches
        # one permutation for every epoch (rollout_storage.py:149,165-170)
        if self.permutation == "device":
            # the draw number lives on the device as well: the launch reads it there, so a captured update (OnPolicyRunner's second HIP
            # graph) draws a NEW permutation at every replay.  Eager calls re-seat the device copy from the host count first (the host
            # count is what seek() / a checkpoint resume set); under capture nothing is re-seated -- replays continue the device count and
            # after_update_replay() advances the host's.
            with torch.inference_mode():        # (the counter may have been created under the runner's inference mode)
                if not torch.cuda.is_current_stream_capturing():
                    self._perm_draws_dev.fill_(self._perm_draws)
                self._perm_draws += 1
                self._perm_draws_dev.add_(1)
            perm = st.permutation(self.num_mini_batches * mb, self._perm_seed, self._perm_draws_dev)
        else:
            perm = torch.randperm(self.num_mini_batches * mb, device=self.device)
        fl = lambda t: t.flatten(0, 1)
        obs = fl(st.observations)
        priv = fl(st.privileged_observations) if st.privileged_observations is not None else obs
        cols = (obs, priv, fl(st.actions), st.values.view(-1), st.advantages.view(-1), st.returns.view(-1),
                st.actions_log_prob.view(-1), fl(st.mu), fl(st.sigma))
        sh = st.shadows() if hasattr(st, "shadows") else None
        if sh is not None and self._check_shadow:
            st.check_shadows()
        sh = dict(obs_bf16=sh[0], priv_bf16=sh[1]) if sh is not None else {}
        net.opt_state[2:8] = 0.0               # the per-update sums [2..5], [7] (and the informational last norm [6]): one fill
        if self._ppo_cfg.aux_coef > 0.0:
            net.opt_state[10:11] = 0.0     # (a slice: a fill kernel -- an indexed scalar store is a host-to-device copy, which a stream capture refuses)
        for _ in range(self.num_learning_epochs):
            for i in range(self.num_mini_batches):
                idx = perm[i * mb:(i + 1) * mb]
                batch = hgym.make_batch(*cols, idx, **sh)
                net.ppo_grad(self._ppo_cfg, batch)
                if dist_utils.active():
                    # ONE bucket, [flat gradient | KL]: nothing of this minibatch is left to run under the exchange (apply needs it),
                    # and two buckets pay the collective's latency twice.  (Rounds 1-2 split the weight-gradient launch in two so that
                    # the actor's bucket travelled under the critic's products: the two half-empty launches cost 106 us more per
                    # minibatch than the one launch -- more than the exchange they hid; profiles/r03_grad_parts_ab.txt.)
                    probe = self.comm_timing is not None
                    if self._comm is not None and (self._comm_p2p != bool(self.comm_flip)):
                        # HGYM_COMM=p2p: one kernel on this stream (reduce-scatter + all-gather over the peer mappings)
                        if probe:
                            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                            ev[0].record()
                        self._comm.allreduce()
                        self._comm_direct_used = True
                    else:
                        h = dist_utils.start_sum(net.grads_ext)
                        if probe:
                            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                            ev[0].record()
                        dist_utils.finish(h)
                    if probe:
                        ev[1].record()
                        self.comm_timing.append(ev + ("p2p" if (self._comm is not None and self._comm_p2p != bool(self.comm_flip)) else "collective",))
                net.ppo_apply(self._ppo_cfg)
        st.clear()
        if not sync:
            return None, None
        o = net.opt_state.cpu()                # the one host read-back of the update
        self.check_comm()                      # (the stream is idle now: reading the exchange's status costs one small copy)
        n = max(float(o[7]), 1.0)
        self.last_denoise_loss = float(o[10]) / n if self._ppo_cfg.aux_coef > 0.0 else None
        return float(o[4]) / n, float(o[3]) / n
