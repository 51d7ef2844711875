"""CPU baseline, kind "reference": the UNMODIFIED reference (`/root/reference/humanoid`) timed on the host cores --
TEST / MEASUREMENT INFRASTRUCTURE (bench.py's `cpu_baseline` leg runs this file as a subprocess; nothing in the product
imports it).  Exists only where `/root/reference` does (the build container); on the GPU box bench.py times the port
(`oracle/ppo_oracle.py`, `oracle/xbot_env_oracle.py`) instead.

What runs: the reference's own `XBotLFreeEnv.step` (humanoid_env.py:189-197 -> legged_robot.py:84-108, PhysX calls no-ops, the
four sim tensors static: SURVEY.md Appendix B) and its own `PPO.act / process_env_step / compute_returns / update`
(algo/ppo/ppo.py:91-184) with XBotLCfgPPO's hyper-parameters, through tests/golden/ref_harness.py -- the same harness that
records the golden fixtures.  Sample = a QUARTER iteration at full width: 15 vec-steps at N envs (15 x N = one real
minibatch of 61 440 samples at N = 4096), GAE over them, and ONE full minibatch of `PPO.update` (forward, autograd
backward, clip_grad_norm_, Adam); one iteration = 4 x (rollout + GAE) + 8 x minibatch.  Thread count: the candidates in
--threads are tried on one vec-step and one small update; the best per phase is used and reported.

    python oracle/ref_timing.py --num-envs 4096 --threads 8,16,32,64      -> one JSON line on stdout
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=15)
    ap.add_argument("--threads", default="")
    a = ap.parse_args()
    import ref_harness as H
    with contextlib.redirect_stdout(io.StringIO()):
        R = H.load_reference()
    N, T = a.num_envs, a.steps
    ncpu = os.cpu_count() or 1
    cands = sorted({int(x) for x in a.threads.split(",") if x} or {min(ncpu, n) for n in (8, 16, 32, 64, 128)})
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        e, cfg = H.make_ref_env(N)
        H.write_sim_state(e, H.synth_sim_state(g, N))
        H.finish_init(e)
        tc = R.class_to_dict(R.XBotLCfgPPO())
        ac = R.ActorCritic(e.num_obs, e.num_privileged_obs, e.num_actions, **tc["policy"])

        def make_alg(steps, mini_batches, epochs):
            k = dict(tc["algorithm"], num_mini_batches=mini_batches, num_learning_epochs=epochs)
            alg = R.PPO(ac, device="cpu", **k)
            alg.init_storage(N, steps, [e.num_obs], [e.num_privileged_obs], [e.num_actions])
            return alg

    e.episode_length_buf = torch.randint_like(e.episode_length_buf, high=int(e.max_episode_length))

    def rollout(alg, steps):
        obs, priv = e.get_observations(), e.get_privileged_observations()
        with torch.inference_mode():
            for _ in range(steps):
                act = alg.act(obs, priv)
                obs, priv, rew, done, info = e.step(act)
                alg.process_env_step(rew, done, info)
        return priv

    # thread-count probe: one vec-step / one 4096-sample minibatch per candidate
    probe = make_alg(1, 1, 1)
    best = {}
    for nt in cands:
        torch.set_num_threads(nt)
        priv = rollout(probe, 1)
        probe.storage.clear()
        t0 = time.perf_counter()
        priv = rollout(probe, 1)
        t_r = time.perf_counter() - t0
        with torch.inference_mode():
            probe.compute_returns(priv)
        t0 = time.perf_counter()
        probe.update()
        t_u = time.perf_counter() - t0
        for k, v in (("rollout", t_r), ("update", t_u)):
            if k not in best or v < best[k][1]:
                best[k] = (nt, v)
    alg = make_alg(T, 1, 1)
    torch.set_num_threads(best["rollout"][0])
    t0 = time.perf_counter()
    priv = rollout(alg, T)
    t_roll = time.perf_counter() - t0
    t0 = time.perf_counter()
    with torch.inference_mode():
        alg.compute_returns(priv)
    t_gae = time.perf_counter() - t0
    torch.set_num_threads(best["update"][0])
    t0 = time.perf_counter()
    alg.update()
    t_mb = time.perf_counter() - t0
    scale = 60.0 / T
    t_iter = scale * (t_roll + t_gae) + 8 * t_mb * (61440.0 * (N / 4096.0)) / (T * N)
    print(json.dumps(dict(
        value=60 * N / t_iter, unit="env-steps/s", cores=max(best["rollout"][0], best["update"][0]), kind="reference",
        threads_rollout=best["rollout"][0], threads_update=best["update"][0], host_cpus=ncpu,
        ms_per_vec_step=t_roll / T * 1e3, gae_ms=t_gae * scale * 1e3, s_per_minibatch=t_mb, s_per_iteration=t_iter,
        sample="the reference's own XBotLFreeEnv.step + PPO (unmodified, via tests/golden/ref_harness.py, PhysX no-op): %d vec-steps at "
               "N=%d + GAE + 1 full PPO.update minibatch of %d samples; iteration = %gx(rollout+GAE) + 8 minibatches; threads tried %s"
               % (T, N, T * N, scale, cands))))


if __name__ == "__main__":
    main()
