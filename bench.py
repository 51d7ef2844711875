#!/usr/bin/env python
"""Headline benchmark: XBot-L PPO, 4096 envs per GPU, synthetic physics step, bf16 MFMA dense layers
(BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one learning iteration of the reference's OnPolicyRunner.learn: 60 vec-steps of rollout
(policy act -> env step -> store), GAE, and the PPO update (2 epochs x 4 minibatches of 61 440 samples with
grad-norm clip + Adam) -- nothing is skipped.  metric = env-steps/s = T*N*G / (collection_time + learn_time),
the reference's own Perf/total_fps definition (algo/ppo/on_policy_runner.py:199-203).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      dominant kernel measured live with HIP events on its launch stream (library hooks)
  cpu_baseline  the CPU oracle (torch fp32 port of the reference's algorithm) timed on the host cores, bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "humanoid-gym_amd"))

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--num-envs", type=int, default=4096, help="envs per GPU (weak scaling)")
    p.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    p.add_argument("--task", default="humanoid_ppo", choices=["humanoid_ppo", "humanoid_dwl_ppo"],
                   help="humanoid_ppo = BASELINE configs[1] (the headline); humanoid_dwl_ppo adds the denoising head (configs[4])")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    return p.parse_args()


def cpu_baseline(num_envs, T=60):
    """The oracle timed on the host: a bounded sample scaled to one full iteration.
    Sample: 2 vec-steps of the env oracle + 2 policy evaluations at N=num_envs, one GAE at (60, N), and one
    PPO minibatch (forward + hand-written backward + clip + Adam) on 4096 samples, scaled to 8 x 61 440."""
    from oracle import ppo_oracle as P
    from oracle import xbot_constants as K
    from oracle.xbot_env_oracle import XBotEnvOracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from env_common import synth_frames
    g = torch.Generator().manual_seed(0)
    N = num_envs
    cores = torch.get_num_threads()
    o = XBotEnvOracle(N)
    o.prime(torch.rand(N, 12, generator=g), torch.rand(N, 3, generator=g), torch.randn(N, 47, generator=g))
    o.ep_len = torch.randint(0, 2400, (N,), generator=g)
    p = P.Params.random(705, 219, 12, K.ACTOR_HIDDEN, K.CRITIC_HIDDEN, g)
    frames = [synth_frames(g, N) for _ in range(3)]
    z = torch.randn(N, 12, generator=g)

    def vec_step(f):
        a, v, lp, mu, sg = P.policy_act(p, torch.clip(o.obs, -18, 18), torch.clip(o.priv, -18, 18), z)
        o.pre_physics(a, torch.rand(N, generator=g), torch.randn(N, 12, generator=g))
        o.pd_torques()
        o.sim.load(*f)
        o.post_physics(torch.rand(N, 6, generator=g), torch.rand(N, 12, generator=g), torch.rand(N, 5, generator=g),
                       torch.randn(N, 47, generator=g))
    vec_step(frames[0])
    t0 = time.perf_counter()
    vec_step(frames[1])
    vec_step(frames[2])
    t_step = (time.perf_counter() - t0) / 2
    r, v = torch.rand(T, N, generator=g), torch.randn(T, N, generator=g)
    d, lv = torch.rand(T, N, generator=g) < 0.01, torch.randn(N, generator=g)
    t0 = time.perf_counter()
    ret, adv = P.gae_returns(r, v, d, lv, K.GAMMA, K.LAM)
    P.normalize_advantages(adv)
    t_gae = time.perf_counter() - t0
    B = 4096
    obs, priv = torch.randn(B, 705, generator=g), torch.randn(B, 219, generator=g)
    act, mu_o, sg_o = torch.randn(B, 12, generator=g), torch.randn(B, 12, generator=g) * 0.3, torch.ones(B, 12)
    val, ad, rt, lp_o = (torch.randn(B, generator=g) for _ in range(4))
    opt = P.Adam(p)
    t0 = time.perf_counter()
    out = P.ppo_loss_and_grads(p, obs, priv, act, val, ad, rt, lp_o - 12.0, mu_o, sg_o)
    P.clip_grad_norm(out["grads"], 1.0)
    opt.step(p, out["grads"], 1e-5)
    t_mb = time.perf_counter() - t0
    mb_full = (T * N) // 4
    t_update = 8 * t_mb * (mb_full / B)
    t_iter = T * t_step + t_gae + t_update
    return dict(value=T * N / t_iter, unit="env-steps/s", cores=cores, kind="port",
                sample="oracle (torch-CPU fp32 port of the reference algorithm): 2 vec-steps + policy at N=%d, 1 GAE (60xN), "
                       "1 PPO minibatch of %d samples scaled to 8x%d; rollout %.1f ms/vec-step, update %.2f s/iter (extrapolated)"
                       % (N, B, mb_full, t_step * 1e3, t_update))


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py measures the MI355X hot path; no GPU visible"
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("HGYM_DIST_BACKEND", "nccl")        # "nccl" IS RCCL on ROCm; "gloo" lets ranks share one GPU (tests)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    os.environ["HGYM_PRECISION"] = args.precision

    from humanoid.algo import PPO
    PPO.precision = args.precision
    from humanoid.envs import task_registry
    from humanoid.utils import get_args
    from hgym import _lib as L
    dev = "cuda:%d" % local
    a = get_args(["--task=" + args.task, "--headless", "--num_envs", str(args.num_envs), "--sim_device", dev, "--rl_device", dev,
                  "--seed", str(5 + rank)])
    env, env_cfg = task_registry.make_env(name=a.task, args=a)
    runner, train_cfg = task_registry.make_alg_runner(env=env, name=a.task, args=a, log_root=None)
    T = runner.num_steps_per_env
    N = env.num_envs

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed warm-up: W iterations, but never fewer than 2 -- the first iteration runs the rollout eagerly, the second captures it
    # into the HIP graph the timed iterations replay (a capture inside the timed region would not be the steady state)
    runner.learn(num_learning_iterations=max(args.warmup, 2), init_at_random_ep_len=True)
    barrier()
    t0 = time.perf_counter()
    # K learning iterations, enqueued back to back (the runner reads nothing back between iterations when it does not log)
    runner.learn(num_learning_iterations=args.steps, init_at_random_ep_len=False)
    coll = runner.last_collection_time * args.steps      # mean per iteration over this call, HIP events on the launch stream
    learn = runner.last_learn_time * args.steps
    barrier()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax)
    value = T * N * world * args.steps / elapsed

    roofline = None
    if not args.no_roofline:
        # live per-kernel timing with HIP events on the launch stream (hgym_prof_*).  Events cannot be recorded inside a
        # replayed HIP graph, so these two iterations run the rollout eagerly (same kernels, same launch order).  EVERY rank
        # runs them (the update contains the gradient all-reduce); only rank 0 records and reports.
        os.environ["HGYM_GRAPH"] = "0"
        if rank == 0:
            L.lib.hgym_prof_enable(1)
        runner.learn(num_learning_iterations=2, init_at_random_ep_len=False)
        torch.cuda.synchronize()
        os.environ["HGYM_GRAPH"] = "1"
    if rank == 0 and not args.no_roofline:
        iter_ms = elapsed / args.steps * 1e3
        mfma_peak = MFMA_BF16_PEAK_TFLOPS if args.precision == "bf16" else 157.3
        classes = [  # (class id, kernel, bound, unit of `work`)
            (L.PROF_ENV_STEP, "env_step_kernel", "hbm"), (L.PROF_MLP_FWD, "mlp_fwd_kernel", "mfma"),
            (L.PROF_POLICY, "mlp_fwd_kernel<32>", "mfma"),
            (L.PROF_MLP_BWD, "mlp_bwd_kernel", "mfma"), (L.PROF_DW, "dw_kernel", "mfma"), (L.PROF_GEMM, "gemm_nt_kernel", "mfma"),
            (L.PROF_LOSS, "ppo_loss_kernel", "hbm"), (L.PROF_REDUCE, "reduce_slabs_kernel", "hbm"),
            (L.PROF_APPLY, "sqnorm+adam_kernel", "hbm"), (L.PROF_GAE, "gae_kernel", "hbm")]
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")     # HBM bytes per launch from rocprofv3 --pmc passes
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("kernels", {})
        ks = []
        for cid, name, bound in classes:
            n, ms, work = L.prof_summary(cid)
            if n == 0 or ms <= 0:
                continue
            peak, unit, scale = (HBM_PEAK_GBS, "GB/s", 1e9) if bound == "hbm" else (mfma_peak, "TFLOP/s", 1e12)
            ach = work / (ms * 1e-3) / scale
            ks.append(dict(kernel=name, bound=bound, launches_per_iter=n // 2, avg_launch_us=ms / n * 1e3, achieved=ach, peak=peak, unit=unit,
                           frac=ach / peak, share_of_iteration=ms / 2 / iter_ms, traffic=traffic.get(name)))
        L.lib.hgym_prof_enable(0)
        ks.sort(key=lambda k: -k["share_of_iteration"])
        dom = ks[0]
        roofline = dict(bound=dom["bound"], achieved=dom["achieved"], peak=dom["peak"], unit=dom["unit"], frac=dom["frac"],
                        traffic=dom["traffic"], kernel=dom["kernel"], launches_per_iter=dom["launches_per_iter"],
                        avg_launch_us=dom["avg_launch_us"], share_of_iteration=dom["share_of_iteration"], kernels=ks[1:])

    if dist is not None:
        dist.barrier()
    if rank == 0:
        out = {
            "metric": "env-steps/s (XBot-L PPO, whole job)", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "XBot-L PPO %d envs/GPU, synthetic physics step, T=60, 2 epochs x 4 minibatches (BASELINE configs[%s])"
                                   % (N, "1" if args.task == "humanoid_ppo" else "4: + denoising head"),
                       "envs_per_gpu": N, "steps_per_env": T, "obs": env.num_obs, "privileged_obs": env.num_privileged_obs,
                       "minibatch": T * N // 4, "parallelism": "dp%d (env shards, RCCL grad all-reduce)" % world},
            "ppo_update_ms": learn / args.steps * 1e3, "collection_ms": coll / args.steps * 1e3,
        }
        if roofline is not None:
            out["roofline"] = roofline
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline leg belongs to the N=1 run only
            out["cpu_baseline"] = cpu_baseline(N, T)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
