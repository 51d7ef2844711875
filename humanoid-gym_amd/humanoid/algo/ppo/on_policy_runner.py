"""OnPolicyRunner with the reference's interface (algo/ppo/on_policy_runner.py:45-307): rollout / learn loop,
checkpoint save / load (same dict keys), inference-policy getters, console + optional TensorBoard logging.

The rollout loop issues, per vec-step, one policy launch group, one fused env launch and one store launch; the
env writes its observations straight into the next rollout-storage slot.  Episode book-keeping stays on the
device and is read back once per iteration (the reference syncs the host every step, :146-152)."""
import atexit
import os
import sys
import queue
import statistics
import threading
import time
from collections import deque
from datetime import datetime

import torch

from .ppo import PPO
from .actor_critic import ActorCritic
from humanoid.algo.vec_env import VecEnv

try:  # optional: logging back-ends are not part of the hot path
    from torch.utils.tensorboard import SummaryWriter
except Exception:  # pragma: no cover
    SummaryWriter = None
try:
    import wandb
except Exception:  # pragma: no cover
    wandb = None


class _CheckpointWriter:
    """One background thread that turns pinned-host snapshots into checkpoint files (OnPolicyRunner.save): the training thread only
    enqueues device -> pinned-host copies behind the update and goes on; the pickling + file write (6-15 ms for XBot-L's 11 MB of
    parameters and Adam moments) happens here, under the next iteration.  Files appear atomically (written to a temporary name,
    then renamed)."""

    def __init__(self):
        self.q = queue.Queue()
        self.thread = None
        self.error = None
        self.lock = threading.Lock()

    def submit(self, job):
        with self.lock:
            if self.thread is None or not self.thread.is_alive():
                self.thread = threading.Thread(target=self._run, name="hgym-checkpoint-writer", daemon=True)
                self.thread.start()
        self.q.put(job)

    def _run(self):
        # this thread's CPU tensor work stays on this thread: the default intra-op pool is one thread per host core (128+ on the GPU boxes), and a
        # parallel region opened from here -- a 3.7 MB copy is enough -- wakes all of them next to the thread that launches the kernels
        # (measured: sporadic 50-350 ms stalls of a checkpoint job and, now and then, of the training thread; profiles/r06_async_checkpoint_default.txt)
        try:
            torch.set_num_threads(1)
        except Exception:      # noqa: BLE001
            pass
        while True:
            job = self.q.get()
            try:
                if job is not None:
                    job()
            except Exception as e:      # noqa: BLE001 -- re-raised on the training thread by wait()
                self.error = e
            finally:
                self.q.task_done()

    def wait(self):
        self.q.join()
        if self.error is not None:
            e, self.error = self.error, None
            raise e


_WRITER = _CheckpointWriter()
atexit.register(lambda: _WRITER.q.join())


class OnPolicyRunner:
    def __init__(self, env: VecEnv, train_cfg, log_dir=None, device="cpu"):
        self.cfg = train_cfg["runner"]
        self.alg_cfg = train_cfg["algorithm"]
        self.policy_cfg = train_cfg["policy"]
        self.all_cfg = train_cfg
        self.wandb_run_name = (datetime.now().strftime("%b%d_%H-%M-%S") + "_" + train_cfg["runner"]["experiment_name"] + "_"
                               + train_cfg["runner"]["run_name"])
        self.device = device
        self.env = env
        num_critic_obs = self.env.num_privileged_obs if self.env.num_privileged_obs is not None else self.env.num_obs
        actor_critic_class = eval(self.cfg["policy_class_name"])  # ActorCritic
        actor_critic = actor_critic_class(self.env.num_obs, num_critic_obs, self.env.num_actions, **self.policy_cfg).to(self.device)
        alg_class = eval(self.cfg["algorithm_class_name"])  # PPO
        self.alg = alg_class(actor_critic, device=self.device, **self.alg_cfg)
        self.num_steps_per_env = self.cfg["num_steps_per_env"]
        self.save_interval = self.cfg["save_interval"]
        self.alg.init_storage(self.env.num_envs, self.num_steps_per_env, [self.env.num_obs], [self.env.num_privileged_obs],
                              [self.env.num_actions])
        self.log_dir = log_dir
        self.writer = None
        self.tot_timesteps = 0
        self.tot_time = 0
        self.current_learning_iteration = 0
        self.last_collection_time = self.last_learn_time = 0.0
        self._graph = None           # captured rollout (HIP graph) and the tensors it owns
        self._graph_warm = False
        self._update_graph = None    # captured compute_returns() + update() (second HIP graph)
        self._update_warm = False
        _, _ = self.env.reset()

    # ------------------------------------------------------------------
    def _rollout_slots(self):
        st = self.alg.storage
        return getattr(st, "_obs_all", None), getattr(st, "_priv_all", None)

    def learn(self, num_learning_iterations, init_at_random_ep_len=False):
        if self.log_dir is not None and self.writer is None:
            if wandb is not None and hasattr(wandb, "init"):
                try:
                    wandb.init(project="XBot", sync_tensorboard=True, name=self.wandb_run_name, config=self.all_cfg)
                except Exception:
                    pass
            if SummaryWriter is not None:
                self.writer = SummaryWriter(log_dir=self.log_dir, flush_secs=10)
            os.makedirs(self.log_dir, exist_ok=True)
        if init_at_random_ep_len:
            self.env.episode_length_buf = torch.randint_like(self.env.episode_length_buf, high=int(self.env.max_episode_length))
        env, alg = self.env, self.alg
        obs = env.get_observations()
        privileged_obs = env.get_privileged_observations()
        critic_obs = privileged_obs if privileged_obs is not None else obs
        obs, critic_obs = obs.to(self.device), critic_obs.to(self.device)
        obs_all, priv_all = self._rollout_slots()
        zero_copy = obs_all is not None and priv_all is not None and hasattr(env, "bind_outputs") and privileged_obs is not None
        if zero_copy:
            obs_all[0].copy_(obs)
            priv_all[0].copy_(critic_obs)
            obs, critic_obs = obs_all[0], priv_all[0]
        alg.actor_critic.train()

        ep_infos = []
        rewbuffer, lenbuffer = deque(maxlen=100), deque(maxlen=100)
        N = env.num_envs
        cur_reward_sum = torch.zeros(N, dtype=torch.float, device=self.device)
        cur_episode_length = torch.zeros(N, dtype=torch.float, device=self.device)
        done_stats = torch.zeros(3, dtype=torch.float, device=self.device)     # sum of returns, sum of lengths, episodes

        # One rollout = num_steps_per_env x {act, env.step, process_env_step}: a few hundred small launches whose host
        # side (Python + ctypes) costs more than the kernels.  Every buffer the step touches has a fixed address (the env
        # writes into the storage slots, all counters live on the device), so after one eager warm-up iteration the whole
        # rollout is captured into a HIP graph and replayed with a single launch per iteration (HGYM_GRAPH=0 disables).
        use_graph = (zero_copy and str(self.device).startswith("cuda") and os.environ.get("HGYM_GRAPH", "1") != "0"
                     and hasattr(torch.cuda, "CUDAGraph"))
        log_on = self.log_dir is not None

        # with zero_copy the env's step finaliser also stores the scalar columns of the transition (bind_transition)
        sink_ok = (zero_copy and hasattr(env, "bind_transition") and hasattr(alg, "transition_sink")
                   and getattr(env.cfg.env, "send_timeouts", False) and os.environ.get("HGYM_ENV_SINK", "1") != "0")

        if sink_ok:
            alg.env_stores_transitions = True
        # With logging the reference reads rewards / dones / infos on the host after every vec-step (:143-156).  Here the step
        # finaliser keeps that book-keeping on the device (LeggedRobot.bind_log_sink: running episode returns / lengths, the
        # last-100-episodes rings, the per-step sums of extras["episode"]) and the host reads it once per iteration.
        log_sink = bool(log_on and sink_ok and hasattr(env, "bind_log_sink") and os.environ.get("HGYM_LOG_SINK", "1") != "0"
                        and env.bind_log_sink(True))
        host_log = log_on and not log_sink
        # ... and, when nothing on the host looks at the per-step extras, the finaliser of vec-step t is not
        # launched by the env at all: it rides as one extra workgroup of the policy launch of step t+1 (2 launches per vec-step)
        defer_ok = (sink_ok and not host_log and hasattr(env, "take_pending_finalize") and isinstance(alg, PPO)
                    and os.environ.get("HGYM_DEFER_FIN", "1") != "0")

        # ... and with the synthetic-physics backend the whole loop body -- act, env.step, process_env_step -- is ONE launch per
        # vec-step (hgym_rollout_step): the env step of a 32-env slice runs right behind the actor tile of the same slice
        fuse_mode = (env.rollout_fused_mode(alg.net) if (defer_ok and hasattr(env, "rollout_fused_mode") and hasattr(alg, "fused_rollout_step")
                                                         and os.environ.get("HGYM_FUSE_ROLLOUT", "1") != "0") else None)
        fuse_ok = fuse_mode is not None
        # "deferred": more than half a chip of envs (8192 per MI355X, BASELINE configs[3]) -- the launch has no critic tiles, the critic
        # runs once over the stored rows behind the last step (nothing inside the rollout needs V(s_t) but the time-out bootstrap,
        # which compute_returns then applies: ppo.py:107-108)
        deferred = fuse_mode == "deferred" and hasattr(alg, "deferred_values")

        def rollout(obs, critic_obs):
            if fuse_ok:
                env.rollout_begin(alg._sample_step, self.num_steps_per_env)
                T = self.num_steps_per_env
                for i in range(T):
                    if deferred:
                        alg.fused_rollout_step(env, i, obs, critic_obs, obs_all[i + 1], priv_all[i + 1], deferred=True)
                    else:
                        # (the slot after next: its older frames are written by this launch, off the next one's critical path)
                        alg.fused_rollout_step(env, i, obs, critic_obs, obs_all[i + 1], priv_all[i + 1],
                                               (obs_all[i + 2], priv_all[i + 2]) if i + 2 <= T else None)
                    obs, critic_obs = obs_all[i + 1], priv_all[i + 1]
                env.rollout_end()
                if deferred:
                    alg.deferred_values()       # part of the collection (and of the captured graph): V of all T + 1 slots in one pass
                return obs, critic_obs
            fin = None
            for i in range(self.num_steps_per_env):
                actions = alg.act(obs, critic_obs, env_fin=fin) if defer_ok else alg.act(obs, critic_obs)
                if zero_copy:
                    env.bind_outputs(obs_all[i + 1], priv_all[i + 1])
                if sink_ok:
                    env.bind_transition(alg.transition_sink(), defer_finalize=True) if defer_ok else env.bind_transition(alg.transition_sink())
                obs, privileged_obs, rewards, dones, infos = env.step(actions)
                if defer_ok:
                    fin = env.take_pending_finalize()
                critic_obs = privileged_obs if privileged_obs is not None else obs
                alg.process_env_step(rewards, dones, infos, **({"stored": True} if sink_ok else {}))
                if host_log:
                    if "episode" in infos:
                        ep_infos.append({k: v.clone() for k, v in infos["episode"].items()})
                    cur_reward_sum.add_(rewards)
                    cur_episode_length.add_(1)
                    d = dones.to(torch.float)
                    done_stats[0] += (cur_reward_sum * d).sum()
                    done_stats[1] += (cur_episode_length * d).sum()
                    done_stats[2] += d.sum()
                    cur_reward_sum.mul_(1.0 - d)
                    cur_episode_length.mul_(1.0 - d)
            if defer_ok:
                env.run_finalize(fin)           # the last step has no following policy launch
            return obs, critic_obs

        # Without logging nothing on the host needs the iteration's results: collection / learn time are then measured with
        # HIP events on the launch stream and read once at the end of learn(), and the update's loss read-back is skipped,
        # so the host enqueues iteration k+1 while the device still runs iteration k (no idle gap at the phase
        # boundaries).  With logging the reference's per-iteration host synchronisation is kept.
        async_iters = ((not log_on) and str(self.device).startswith("cuda") and isinstance(alg, PPO)
                       and os.environ.get("HGYM_ASYNC", "1") != "0")
        # With the device-side log sink the logging run does not synchronise per iteration either: everything log() prints is
        # copied to pinned host memory behind the update (stream-ordered), and the host formats iteration k's block while the
        # device runs iteration k + 1 -- the log appears one iteration late, with the same content in the same order.
        async_log = (log_sink and str(self.device).startswith("cuda") and isinstance(alg, PPO)
                     and os.environ.get("HGYM_ASYNC", "1") != "0")
        pending = None
        marks = []
        # the captured launches hold HgymEnvConfig and the sink's gamma BY VALUE: a change between learn() calls (reward scales,
        # command ranges, push / noise settings written into the env's native config, alg.gamma -- what a curriculum script does)
        # must re-capture, as the eager reference would simply see it
        gkey = (id(env), id(alg.storage), log_on, log_sink, sink_ok, defer_ok, fuse_mode, getattr(alg, "gamma", None),
                env.native_config_digest() if hasattr(env, "native_config_digest") else None,
                getattr(env, "_rows_ahead", None), getattr(env, "_l0_ahead", None), getattr(alg.storage, "_obs_bf16", None) is not None)
        # the update as a second captured graph (HGYM_GRAPH_UPDATE=0: eager, as until round 5): only where nothing on the host reads the
        # update's results inside the iteration (the asynchronous loops) and every launch argument is iteration-invariant
        graph_update = bool(use_graph and (async_iters or async_log) and hasattr(alg, "update_capturable") and alg.update_capturable()
                            and os.environ.get("HGYM_GRAPH_UPDATE", "1") != "0")
        ukey = (gkey, alg.update_graph_key(), deferred) if graph_update else None
        if not graph_update:
            self._update_graph = None
        tot_iter = self.current_learning_iteration + num_learning_iterations
        try:
            for it in range(self.current_learning_iteration, tot_iter):
                start = time.time()
                if async_iters or async_log:
                    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                    ev[0].record()
                with torch.inference_mode():
                    g = self._graph
                    if use_graph and g is not None and g["key"] == gkey:
                        g["graph"].replay()
                        alg.storage.step = self.num_steps_per_env
                        if deferred:
                            alg._deferred_ready = True              # (the captured rollout ends with deferred_values' launches)
                        alg.storage.shadow_valid = list(g["shadow_valid"])     # the replayed launches wrote the same shadow slots
                        obs, critic_obs = g["out"]
                        ep_infos = g["ep_infos"]
                        cur_reward_sum, cur_episode_length, done_stats = g["stats"]
                    elif use_graph and self._graph_warm:
                        graph = torch.cuda.CUDAGraph()
                        ep_infos = []
                        torch.cuda.synchronize()
                        # thread-local capture mode: with torch.distributed initialised, the RCCL watchdog thread polls events
                        # concurrently; only this thread's launches belong to the capture
                        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                            out = rollout(obs_all[0], priv_all[0])
                        self._graph = dict(graph=graph, out=out, ep_infos=ep_infos, key=gkey,
                                           stats=(cur_reward_sum, cur_episode_length, done_stats),
                                           shadow_valid=list(getattr(alg.storage, "shadow_valid", [])))
                        alg.storage.step = 0
                        graph.replay()                  # capture does not execute: run the captured rollout once
                        alg.storage.step = self.num_steps_per_env
                        obs, critic_obs = out
                    else:
                        obs, critic_obs = rollout(obs, critic_obs)
                        self._graph_warm = True
                    if async_iters or async_log:
                        ev[1].record()
                    elif str(self.device).startswith("cuda"):
                        torch.cuda.synchronize()
                    stop = time.time()
                    collection_time = stop - start
                    start = stop
                    ug = self._update_graph
                    replayed_update = False
                    if graph_update and ug is not None and ug["key"] == ukey and self._graph is not None and self._graph["key"] == gkey:
                        # steady state: the whole iteration is two graph launches (rollout, update) with the phase event between them
                        ug["graph"].replay()
                        alg.after_update_replay()
                        replayed_update = True
                    elif not (graph_update and self._update_warm and self._graph is not None and self._graph["key"] == gkey):
                        alg.compute_returns(critic_obs)
                if replayed_update:
                    mean_value_loss = mean_surrogate_loss = None
                elif graph_update and self._update_warm and self._graph is not None and self._graph["key"] == gkey:
                    # the rollout has just been captured (or is replayed) and the update has run eagerly at least once: capture
                    # compute_returns() + update(sync=False) -- ~45 launches per iteration that otherwise come from Python + ctypes on every
                    # rank -- into a second graph.  Nothing in their arguments changes between iterations: the permutation's draw number,
                    # Adam's step, the learning rate and the exchange's call number are read on the device (PPO.update_capturable)
                    ugraph = torch.cuda.CUDAGraph()
                    torch.cuda.synchronize()
                    # (inside inference mode like the rollout's capture: capture_begin updates the generator's graph-state tensors in
                    # place, and the first capture created them as inference tensors)
                    with torch.inference_mode():
                        with torch.cuda.graph(ugraph, capture_error_mode="thread_local"):
                            alg.compute_returns(critic_obs)
                            alg.update(sync=False)
                    self._update_graph = dict(graph=ugraph, key=ukey)
                    ugraph.replay()                 # capture does not execute (update()'s host book-keeping has run once, for this replay)
                    mean_value_loss = mean_surrogate_loss = None
                else:
                    mean_value_loss, mean_surrogate_loss = alg.update(sync=False) if (async_iters or async_log) else alg.update()
                    self._update_warm = True
                if zero_copy:                       # storage.clear() rotated slot T into slot 0
                    obs, critic_obs = obs_all[0], priv_all[0]
                if it % self.save_interval == 0:
                    self._check_replicas("iteration %d" % it)       # (data-parallel runs only; collective: every rank, same iteration)
                stop = time.time()
                learn_time = stop - start
                if async_iters:
                    ev[2].record()
                    marks.append(ev)
                elif async_log:
                    ev[2].record()
                    snap = self._log_snapshot(env, alg, it & 1)
                    if pending is not None:
                        self._log_flush(pending, num_learning_iterations)
                    pending = dict(it=it, ev=ev, snap=snap)
                    if it % self.save_interval == 0:
                        self.save(os.path.join(self.log_dir, "model_{}.pt".format(it)), wait=False)
                    continue
                else:
                    self.last_collection_time, self.last_learn_time = collection_time, learn_time
                if self.log_dir is not None:
                    if log_sink:
                        # one read-back: mean over this iteration's steps of extras["episode"] (what the reference's ep_infos list
                        # averages to), and the rings that ARE the reference's rewbuffer / lenbuffer (the last 100 finished episodes)
                        ep_mean, ring_r, ring_l = env.log_sink_read()
                        ep_infos = [ep_mean]
                        rewbuffer, lenbuffer = deque(ring_r, maxlen=100), deque(ring_l, maxlen=100)
                    else:
                        s = done_stats.cpu()
                        if float(s[2]) > 0:
                            rewbuffer.append(float(s[0] / s[2]))
                            lenbuffer.append(float(s[1] / s[2]))
                        done_stats.zero_()
                    self.log(locals())
                    if it % self.save_interval == 0:
                        self.save(os.path.join(self.log_dir, "model_{}.pt".format(it)), wait=False)
                if self._graph is None or ep_infos is not self._graph["ep_infos"]:
                    ep_infos.clear()
        finally:
            # also on an exception / KeyboardInterrupt inside the loop: the last finished iteration's log block is still printed
            # (async logging runs one iteration behind) and the env's bindings into the rollout storage are released
            if pending is not None:
                try:
                    self._log_flush(pending, num_learning_iterations)
                except Exception:      # the device may be the thing that failed
                    pass
            if zero_copy:
                env.bind_outputs(None, None)
            if sink_ok:
                env.bind_transition(None)
                alg.env_stores_transitions = False
            if log_sink:
                env.bind_log_sink(False)
        if marks:                               # mean device time per iteration of this call (HIP events, one sync)
            torch.cuda.synchronize()
            if hasattr(alg, "check_comm"):
                alg.check_comm()                # the asynchronous loop read nothing back: the exchange's status word, once per call
            self.last_collection_time = sum(a.elapsed_time(b) for a, b, _ in marks) * 1e-3 / len(marks)
            self.last_learn_time = sum(b.elapsed_time(c) for _, b, c in marks) * 1e-3 / len(marks)
            # per-iteration device times of this call (ms; iteration k's start event to iteration k + 1's, i.e. including whatever idles
            # between them): bench.py reports their median next to the mean ("median of >= 20")
            self.last_iteration_ms = [marks[k][0].elapsed_time(marks[k + 1][0]) for k in range(len(marks) - 1)] + [marks[-1][0].elapsed_time(marks[-1][2])]
        self.current_learning_iteration += num_learning_iterations
        self._check_replicas("end of learn() at iteration %d" % self.current_learning_iteration)
        if self.log_dir is not None:
            self.save(os.path.join(self.log_dir, "model_{}.pt".format(self.current_learning_iteration)), wait=False)
            if os.environ.get("HGYM_ASYNC_SAVE", "1") == "0":
                self.wait_for_saves()       # (synchronous path: nothing is pending; kept for symmetry)
            # else: the background writer finishes the file; wait_for_saves() / load() / interpreter exit wait for it (save()'s docstring)

    # ------------------------------------------------------------------
    def _check_replicas(self, what):
        """Data-parallel training: every save_interval iterations and at the end of learn() the ranks compare an exact digest of their
        parameters (and whether any rank's direct gradient exchange saw an expired wait) and ALL raise dist_utils.ReplicaMismatch if they
        differ -- a replica that silently diverged (a stale line over xGMI, a time-out only one rank noticed) would otherwise train on as
        a different policy.  No-op on one rank.  HGYM_REPLICA_CHECK=0 switches it off."""
        from . import dist_utils
        if not dist_utils.active() or os.environ.get("HGYM_REPLICA_CHECK", "1") == "0":
            return None
        alg = self.alg
        net = getattr(alg, "net", None)
        if net is None:
            return None
        expired = 0
        comm = getattr(alg, "_comm", None)
        if comm is not None and (getattr(alg, "_comm_p2p", False) or getattr(alg, "_comm_direct_used", False)):
            expired = int(comm.read_status()[0] != 0)
        self.last_replica_digest = dist_utils.check_replicas(net.params, lr=float(net.opt_state[0]), comm_expired=expired, what=what)
        return self.last_replica_digest

    def _log_snapshot(self, env, alg, slot):
        """Enqueue the device -> pinned-host copies of everything one iteration's log block needs (the optimiser's scalar state:
        loss sums, learning rate; the env's log sink: extras["episode"] sums and the last-100-episodes rings; the mean action
        std), then clear the sink's per-iteration sums.  Stream-ordered behind the update; read in _log_flush after the event."""
        if getattr(self, "_log_pin", None) is None:
            mk = lambda n, dt: [torch.empty(n, dtype=dt).pin_memory() for _ in range(2)]
            self._log_pin = dict(opt=mk(alg.net.opt_state.numel(), alg.net.opt_state.dtype), ls=mk(env._buf.log_stats.numel(), torch.float32),
                                 std=mk(1, torch.float32))
        pin = self._log_pin
        pin["opt"][slot].copy_(alg.net.opt_state, non_blocking=True)
        pin["ls"][slot].copy_(env._buf.log_stats, non_blocking=True)
        pin["std"][slot].copy_(alg.actor_critic.std.detach().mean().reshape(1), non_blocking=True)
        comm_words = alg.comm_status_snapshot(slot) if hasattr(alg, "comm_status_snapshot") else None
        env._buf.log_stats[:23].zero_()
        done = torch.cuda.Event()
        done.record()
        return dict(slot=slot, done=done, aux=alg._ppo_cfg.aux_coef > 0.0, comm=comm_words)

    def _log_flush(self, pending, num_learning_iterations):
        """Print / record the log block of a finished iteration from its host snapshot (no device access: the device is busy with
        the next iteration)."""
        from humanoid.envs.base.legged_robot import KERNEL_REWARD_TERMS
        snap, ev = pending["snap"], pending["ev"]
        snap["done"].synchronize()
        if snap.get("comm") is not None:       # N > 1 with the direct gradient exchange: an expired wait must not go unnoticed
            self.alg.check_comm(snap["comm"].tolist())
        pin, slot = self._log_pin, snap["slot"]
        o, ls = pin["opt"][slot], pin["ls"][slot]
        n = max(float(o[7]), 1.0)
        self.alg.last_denoise_loss = float(o[10]) / n if snap["aux"] else None
        steps = max(float(ls[22]), 1.0)
        ep = {"rew_" + nm: float(ls[KERNEL_REWARD_TERMS.index(nm)]) / steps for nm in self.env.reward_names}
        k = int(ls[25])
        collection_time, learn_time = ev[0].elapsed_time(ev[1]) * 1e-3, ev[1].elapsed_time(ev[2]) * 1e-3
        self.last_collection_time, self.last_learn_time = collection_time, learn_time
        self.log(dict(it=pending["it"], num_learning_iterations=num_learning_iterations, collection_time=collection_time,
                      learn_time=learn_time, mean_value_loss=float(o[4]) / n, mean_surrogate_loss=float(o[3]) / n, ep_infos=[ep],
                      rewbuffer=deque(ls[32:32 + k].tolist(), maxlen=100), lenbuffer=deque(ls[132:132 + k].tolist(), maxlen=100),
                      learning_rate=float(o[0]), mean_std=float(pin["std"][slot][0])))

    def log(self, locs, width=80, pad=35):
        self.tot_timesteps += self.num_steps_per_env * self.env.num_envs
        iteration_time = locs["collection_time"] + locs["learn_time"]
        self.tot_time += iteration_time
        scal = (lambda *a: self.writer.add_scalar(*a)) if self.writer is not None else (lambda *a: None)
        ep_string = ""
        if locs["ep_infos"]:
            for key in locs["ep_infos"][0]:
                if all(isinstance(e[key], float) for e in locs["ep_infos"]):        # the device-side log sink hands in host numbers
                    value = sum(e[key] for e in locs["ep_infos"]) / len(locs["ep_infos"])
                else:
                    vals = torch.stack([torch.as_tensor(e[key], device=self.device).float().reshape(()) for e in locs["ep_infos"]])
                    value = float(vals.mean())
                scal("Episode/" + key, value, locs["it"])
                ep_string += f"""{f'Mean episode {key}:':>{pad}} {value:.4f}\n"""
        mean_std = locs["mean_std"] if "mean_std" in locs else float(self.alg.actor_critic.std.detach().mean())
        learning_rate = locs["learning_rate"] if "learning_rate" in locs else self.alg.learning_rate
        fps = int(self.num_steps_per_env * self.env.num_envs / iteration_time)
        scal("Loss/value_function", locs["mean_value_loss"], locs["it"])
        scal("Loss/surrogate", locs["mean_surrogate_loss"], locs["it"])
        scal("Loss/learning_rate", learning_rate, locs["it"])
        if getattr(self.alg, "denoise_coef", 0.0) and getattr(self.alg, "last_denoise_loss", None) is not None:
            scal("Loss/denoise_mse", self.alg.last_denoise_loss, locs["it"])
        scal("Policy/mean_noise_std", mean_std, locs["it"])
        scal("Perf/total_fps", fps, locs["it"])
        scal("Perf/collection time", locs["collection_time"], locs["it"])
        scal("Perf/learning_time", locs["learn_time"], locs["it"])
        have_eps = len(locs["rewbuffer"]) > 0
        if have_eps:
            scal("Train/mean_reward", statistics.mean(locs["rewbuffer"]), locs["it"])
            scal("Train/mean_episode_length", statistics.mean(locs["lenbuffer"]), locs["it"])
            scal("Train/mean_reward/time", statistics.mean(locs["rewbuffer"]), self.tot_time)
            scal("Train/mean_episode_length/time", statistics.mean(locs["lenbuffer"]), self.tot_time)
        head = f" \033[1m Learning iteration {locs['it']}/{self.current_learning_iteration + locs['num_learning_iterations']} \033[0m "
        out = (f"""{'#' * width}\n{head.center(width, ' ')}\n\n"""
               f"""{'Computation:':>{pad}} {fps:.0f} steps/s (collection: {locs['collection_time']:.3f}s, learning {locs['learn_time']:.3f}s)\n"""
               f"""{'Value function loss:':>{pad}} {locs['mean_value_loss']:.4f}\n"""
               f"""{'Surrogate loss:':>{pad}} {locs['mean_surrogate_loss']:.4f}\n"""
               f"""{'Mean action noise std:':>{pad}} {mean_std:.2f}\n""")
        if have_eps:
            out += (f"""{'Mean reward:':>{pad}} {statistics.mean(locs['rewbuffer']):.2f}\n"""
                    f"""{'Mean episode length:':>{pad}} {statistics.mean(locs['lenbuffer']):.2f}\n""")
        out += ep_string
        done = locs["it"] + 1 - self.current_learning_iteration
        eta = self.tot_time / max(done, 1) * (locs["num_learning_iterations"] - done)
        out += (f"""{'-' * width}\n{'Total timesteps:':>{pad}} {self.tot_timesteps}\n"""
                f"""{'Iteration time:':>{pad}} {iteration_time:.2f}s\n{'Total time:':>{pad}} {self.tot_time:.2f}s\n"""
                f"""{'ETA:':>{pad}} {eta:.1f}s\n""")
        print(out)

    def invalidate_graph(self):
        """Drop the captured rollout; the next learn() iteration runs eagerly and the one after re-captures."""
        self._graph, self._graph_warm = None, False
        self._update_graph, self._update_warm = None, False

    def save(self, path, infos=None, wait=True):
        """on_policy_runner.py:274-281 (same dict, same keys).  wait=True (the reference's semantics, and what a direct caller gets): the
        file is on disk when the call returns.  On the device path the tensors are first copied to pinned host memory stream-side (behind
        whatever the update has enqueued -- the host does not wait) and pickled + written by a background thread, so a checkpoint costs the
        training thread ~0.1 ms instead of a device sync + 5-15 ms.  learn() passes wait=False for ALL its checkpoints, the final one
        included (round 6: default; `test_background_checkpoint_equals_the_synchronous_one`): the file of the last iteration is complete a
        few milliseconds AFTER learn() returns -- `wait_for_saves()` (called by load(), by the next save(wait=True), and at interpreter
        exit, which is when scripts/train.py ends) waits for it.  A caller that reads model_<it>.pt from the same process right after
        learn() calls runner.wait_for_saves() first; HGYM_ASYNC_SAVE=0 restores the reference's blocking torch.save everywhere.
        (Round 4 kept this opt-in because learn() then WAITED for the final file and the writer thread's wake-up made that wait 5-75 ms
        instead of a steady 5 ms, profiles/r04_async_checkpoint_ab.txt; with nothing waiting inside learn() that jitter is off the
        training thread.)"""
        t0 = time.time()
        net = getattr(self.alg, "net", None)
        if (net is None or not str(self.device).startswith("cuda") or os.environ.get("HGYM_ASYNC_SAVE", "1") == "0"
                or not hasattr(self.alg.actor_critic, "_net")):
            torch.save({"model_state_dict": self.alg.actor_critic.state_dict(),
                        "optimizer_state_dict": self.alg.optimizer.state_dict(),
                        "iter": self.current_learning_iteration, "infos": infos}, path)
        else:
            if getattr(self, "_save_pin", None) is None:
                mk = lambda n, dt: torch.empty(n, dtype=dt).pin_memory()
                self._save_pin = [dict(params=mk(net.P, torch.float32), m=mk(net.P, torch.float32), v=mk(net.P, torch.float32),
                                       opt=mk(net.opt_state.numel(), net.opt_state.dtype), busy=threading.Event()) for _ in range(2)]
                for b in self._save_pin:
                    b["busy"].set()         # set = free
                self._save_n = 0
            buf = self._save_pin[self._save_n & 1]
            self._save_n += 1
            buf["busy"].wait()              # (two snapshots in flight at most: the writer is two checkpoints behind only if the disk is)
            buf["busy"].clear()
            buf["params"].copy_(net.params, non_blocking=True)
            buf["m"].copy_(net.adam_m, non_blocking=True)
            buf["v"].copy_(net.adam_v, non_blocking=True)
            buf["opt"].copy_(net.opt_state, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
            # name -> (offset, shape) of every parameter in the flat vector: state_dict order = nn.Module's named_parameters order
            base = net.params.data_ptr()
            layout = [(k, (v.data_ptr() - base) // 4, tuple(v.shape)) for k, v in self.alg.actor_critic.state_dict().items()]
            opt_layout = [((v.data_ptr() - base) // 4, tuple(v.shape)) for v in net.views.values()]
            # (alg.optimizer.param_groups reads the learning rate from the device: a host sync; the snapshot carries it instead)
            it, group = self.current_learning_iteration, dict(lr=None, **type(self.alg.optimizer).HYPER)
            # every state_dict entry must be an fp32 view INTO the flat vector (the cut below is by offset and shape only)
            for (k, o, shp), v in zip(layout, self.alg.actor_critic.state_dict().values()):
                n = int(torch.Size(shp).numel())
                assert v.dtype == torch.float32 and v.numel() == n and v.is_contiguous(), "%s is not an fp32 view of the flat parameter vector" % k
                assert 0 <= o and o + n <= net.P and (v.data_ptr() - base) % 4 == 0, "%s lies outside the flat parameter vector" % k

            def job():
                try:
                    tj = [time.perf_counter()]
                    done.synchronize()
                    tj.append(time.perf_counter())
                    # views INTO the snapshot, no copies: torch.save writes each flat buffer once (tensors that share a storage are
                    # pickled as offset + shape into it), torch.load hands back the same named tensors
                    cut = lambda t, o, shp: t[o:o + int(torch.Size(shp).numel())].view(shp)
                    lr, step = float(buf["opt"][0]), float(buf["opt"][1])
                    model = {k: cut(buf["params"], o, shp) for k, o, shp in layout}
                    state = {i: dict(step=torch.tensor(step), exp_avg=cut(buf["m"], o, shp), exp_avg_sq=cut(buf["v"], o, shp))
                             for i, (o, shp) in enumerate(opt_layout)}
                    group["lr"] = lr
                    ck = {"model_state_dict": model,
                          "optimizer_state_dict": dict(state=state, param_groups=[dict(group, params=list(range(len(state))))]),
                          "iter": it, "infos": infos}
                    tmp = path + ".tmp%d" % os.getpid()
                    tj.append(time.perf_counter())
                    torch.save(ck, tmp)
                    tj.append(time.perf_counter())
                    os.replace(tmp, path)
                    tj.append(time.perf_counter())
                    if os.environ.get("HGYM_SAVE_TRACE"):
                        sys.stderr.write("writer job %s: event %.2f, cut %.2f, torch.save %.2f, rename %.2f ms\n" % (
                            os.path.basename(path), *[(b - a) * 1e3 for a, b in zip(tj, tj[1:])]))
                finally:
                    buf["busy"].set()
            _WRITER.submit(job)
            if wait:
                _WRITER.wait()
        self.save_time_s = getattr(self, "save_time_s", 0.0) + (time.time() - t0)     # host time the TRAINING thread spent in checkpoints (bench.py reports it)

    def wait_for_saves(self):
        """Block until every checkpoint handed to the background writer is on disk (re-raises a writer error)."""
        t0 = time.time()
        _WRITER.wait()
        self.save_time_s = getattr(self, "save_time_s", 0.0) + (time.time() - t0)

    def load(self, path, load_optimizer=True):
        _WRITER.wait()                      # a checkpoint this process is still writing
        loaded = torch.load(path, map_location=self.device)
        self.alg.actor_critic.load_state_dict(loaded["model_state_dict"])
        if load_optimizer:
            self.alg.optimizer.load_state_dict(loaded["optimizer_state_dict"])
        self.current_learning_iteration = loaded["iter"]
        if hasattr(self.alg, "seek"):
            self.alg.seek(self.current_learning_iteration, self.num_steps_per_env)
        if hasattr(self.env, "seek"):       # the env's draw streams (commands, pushes, noise, resets) continue as well
            self.env.seek(self.current_learning_iteration, self.num_steps_per_env)
        return loaded["infos"]

    def get_inference_policy(self, device=None):
        self.alg.actor_critic.eval()
        if device is not None and str(device) != str(self.device):
            import copy
            return copy.deepcopy(self.alg.actor_critic.actor).to(device)
        return self.alg.actor_critic.act_inference

    def get_inference_critic(self, device=None):
        self.alg.actor_critic.eval()
        return self.alg.actor_critic.evaluate
