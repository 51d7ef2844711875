#!/usr/bin/env python
"""Headline benchmark: XBot-L PPO, 4096 envs per GPU, synthetic physics step, bf16 MFMA dense layers
(BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Both forms work for N > 1: started WITHOUT a torch.distributed environment (no WORLD_SIZE), `--gpus N` re-executes itself
under torch.distributed.run with N ranks on 127.0.0.1 (one process per GPU, RCCL); started by torch.distributed.run it
reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.

A "step" is one learning iteration of the reference's OnPolicyRunner.learn: 60 vec-steps of rollout
(policy act -> env step -> store), GAE, and the PPO update (2 epochs x 4 minibatches of 61 440 samples with
grad-norm clip + Adam) -- nothing is skipped.  metric = env-steps/s = T*N*G / wall time, the reference's own
Perf/total_fps definition (algo/ppo/on_policy_runner.py:199-203) over the whole job.

Prints ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline      dominant kernel measured live with HIP events on its launch stream (library hooks), others under "kernels"
  cpu_baseline  N=1 only: the reference itself (kind "reference", when /root/reference exists: the build container) or the
                CPU oracle (kind "port", the GPU box) timed on the host cores, bounded sample, best thread count
  comm          N>1 only: the gradient all-reduce as the update sees it (exposed microseconds per minibatch)
  configs       N=1 only: further single-GPU configurations measured in the same run, each with its own numbers --
                BASELINE configs[3] (8192 envs/GPU, history stack 15; own roofline), the headline config with logging ON
                (train.py's default: per-iteration host sync, console / TensorBoard writes), configs[4] (denoising head)
"""
import argparse
import contextlib
import io
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "humanoid-gym_amd"))

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak
REFERENCE_ROOT = "/root/reference"
EXTRA_CONFIGS = ("envs8192", "logging", "dwl", "fp32")


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)         # warm-up 5, >= 20 timed iterations
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--num-envs", type=int, default=4096, help="envs per GPU (weak scaling)")
    p.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    p.add_argument("--task", default="humanoid_ppo", choices=["humanoid_ppo", "humanoid_dwl_ppo"],
                   help="humanoid_ppo = BASELINE configs[1] (the headline); humanoid_dwl_ppo adds the denoising head (configs[4])")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--pmc", action="store_true", help="(default at --gpus 1 since round 4; kept for old command lines)")
    p.add_argument("--no-pmc", action="store_true",
                   help="do NOT collect roofline.traffic in this run.  By default (one GPU, roofline on) two extra rocprofv3 passes (--kernel-trace "
                        "--pmc FETCH_SIZE, then WRITE_SIZE; counters + kernel-trace only) run over a short run of the same workload, ~40 s each, "
                        "300 s limit; if they fail or are switched off the committed profiles/pmc_traffic.json is quoted, with its provenance")
    p.add_argument("--configs", default=",".join(EXTRA_CONFIGS),
                   help="extra single-GPU configurations reported under \"configs\" (N=1 only); \"\" or none: skip")
    return p.parse_args()


# ------------------------------------------------------------------------------------------------ CPU baseline
def _thread_candidates():
    n = os.cpu_count() or 1
    return sorted({min(n, c) for c in (8, 16, 32, 64, 128)})


def _best_threads(fn, cands):
    """fn() timed once per candidate thread count (after one untimed call at the first); returns (threads, seconds)."""
    best = None
    for i, nt in enumerate(cands):
        torch.set_num_threads(nt)
        if i == 0:
            fn()
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        if best is None or dt < best[1]:
            best = (nt, dt)
    return best


def sim2sim_policy_step_us(steps=2000):
    """The CPU half of BASELINE configs[0] that can run without MuJoCo (absent, no network): per 100 Hz control step the
    observation-frame assembly, 15-frame history and TorchScript actor call of the reference's deployment loop
    (scripts/sim2sim.py:124-150), restated on a random-init actor of XBot-L's shape; microseconds per control step."""
    import math
    from collections import deque
    import numpy as np
    from oracle import xbot_constants as K
    layers, d = [], 705
    for h in K.ACTOR_HIDDEN:
        layers += [torch.nn.Linear(d, h), torch.nn.ELU()]
        d = h
    layers.append(torch.nn.Linear(d, 12))
    policy = torch.jit.script(torch.nn.Sequential(*layers))
    rng = np.random.default_rng(0)
    hist = deque(np.zeros([1, 47], dtype=np.double) for _ in range(15))
    action = np.zeros(12, dtype=np.double)
    torch.set_num_threads(1)                      # a 1 x 705 forward: more threads only add wake-up latency
    t0 = None
    for k in range(steps + 50):
        if k == 50:
            t0 = time.perf_counter()
        q, dq, omega, eu = rng.standard_normal(12), rng.standard_normal(12), rng.standard_normal(3), rng.standard_normal(3)
        obs = np.zeros([1, 47], dtype=np.float32)
        obs[0, 0] = math.sin(2 * math.pi * k * 10 * 0.001 / 0.64)
        obs[0, 1] = math.cos(2 * math.pi * k * 10 * 0.001 / 0.64)
        obs[0, 2:5] = (0.4 * 2.0, 0.0, 0.0)
        obs[0, 5:17] = q
        obs[0, 17:29] = dq * 0.05
        obs[0, 29:41] = action
        obs[0, 41:44] = omega
        obs[0, 44:47] = eu
        obs = np.clip(obs, -18.0, 18.0)
        hist.append(obs)
        hist.popleft()
        x = np.zeros([1, 705], dtype=np.float32)
        for i in range(15):
            x[0, i * 47:(i + 1) * 47] = hist[i][0, :]
        action[:] = policy(torch.tensor(x))[0].detach().numpy()
        action = np.clip(action, -18.0, 18.0)
    return (time.perf_counter() - t0) / steps * 1e6


def cpu_baseline_port(num_envs, T=60, full_minibatch=True):
    """The oracle (torch-CPU fp32 port of the reference algorithm) timed on the host cores, thread count chosen per phase.
    Sample: 2 vec-steps of the env oracle + policy at N = num_envs, one GAE at (T, N), ONE FULL PPO minibatch (T*N/4 samples:
    forward, hand-written backward, clip, Adam); one iteration = T vec-steps + GAE + 8 such minibatches."""
    from oracle import ppo_oracle as P
    from oracle import xbot_constants as K
    from oracle.xbot_env_oracle import XBotEnvOracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from env_common import synth_frames
    g = torch.Generator().manual_seed(0)
    N = num_envs
    cands = _thread_candidates()
    o = XBotEnvOracle(N)
    o.prime(torch.rand(N, 12, generator=g), torch.rand(N, 3, generator=g), torch.randn(N, 47, generator=g))
    o.ep_len = torch.randint(0, 2400, (N,), generator=g)
    p = P.Params.random(705, 219, 12, K.ACTOR_HIDDEN, K.CRITIC_HIDDEN, g)
    frames = [synth_frames(g, N) for _ in range(2)]
    z = torch.randn(N, 12, generator=g)
    k = [0]

    def vec_steps():
        for _ in range(2):
            f = frames[k[0] % 2]
            k[0] += 1
            a, v, lp, mu, sg = P.policy_act(p, torch.clip(o.obs, -18, 18), torch.clip(o.priv, -18, 18), z)
            o.pre_physics(a, torch.rand(N, generator=g), torch.randn(N, 12, generator=g))
            o.pd_torques()
            o.sim.load(*f)
            o.post_physics(torch.rand(N, 6, generator=g), torch.rand(N, 12, generator=g), torch.rand(N, 5, generator=g),
                           torch.randn(N, 47, generator=g))
    nt_r, t2 = _best_threads(vec_steps, cands)
    t_step = t2 / 2
    r, v = torch.rand(T, N, generator=g), torch.randn(T, N, generator=g)
    d, lv = torch.rand(T, N, generator=g) < 0.01, torch.randn(N, generator=g)
    torch.set_num_threads(nt_r)
    t0 = time.perf_counter()
    ret, adv = P.gae_returns(r, v, d, lv, K.GAMMA, K.LAM)
    P.normalize_advantages(adv)
    t_gae = time.perf_counter() - t0
    B = (T * N) // 4 if full_minibatch else min(4096, (T * N) // 4)
    obs, priv = torch.randn(B, 705, generator=g), torch.randn(B, 219, generator=g)
    act, mu_o, sg_o = torch.randn(B, 12, generator=g), torch.randn(B, 12, generator=g) * 0.3, torch.ones(B, 12)
    val, ad, rt, lp_o = (torch.randn(B, generator=g) for _ in range(4))
    opt = P.Adam(p)

    def minibatch():
        out = P.ppo_loss_and_grads(p, obs, priv, act, val, ad, rt, lp_o - 12.0, mu_o, sg_o)
        P.clip_grad_norm(out["grads"], 1.0)
        opt.step(p, out["grads"], 1e-5)
    nt_u, t_mb = _best_threads(minibatch, cands)
    t_update = 8 * t_mb * ((T * N) // 4) / B
    t_iter = T * t_step + t_gae + t_update
    return dict(value=T * N / t_iter, unit="env-steps/s", cores=max(nt_r, nt_u), kind="port", threads_rollout=nt_r,
                threads_update=nt_u, host_cpus=os.cpu_count(), ms_per_vec_step=t_step * 1e3, s_per_minibatch=t_mb,
                s_per_iteration=t_iter,
                sample="oracle (torch-CPU fp32 port of the reference algorithm; /root/reference is not on this host): 2 vec-steps + "
                       "policy at N=%d, 1 GAE (%dxN), 1 PPO minibatch of %d samples (forward, backward, clip, Adam); iteration = %d "
                       "vec-steps + GAE + 8 minibatches; threads tried %s" % (N, T, B, T, cands))


def cpu_baseline(num_envs, T=60, full_minibatch=True, allow_reference=True):
    """kind "reference" where the reference exists (oracle/ref_timing.py, own interpreter: its package is also called
    `humanoid`), else the port.  Either way the sim2sim deployment loop's CPU half is timed beside it."""
    out = None
    if allow_reference and os.path.isdir(os.path.join(REFERENCE_ROOT, "humanoid")):
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_timing.py"), "--num-envs", str(num_envs),
                                "--threads", ",".join(map(str, _thread_candidates()))],
                               capture_output=True, text=True, timeout=600)
            out = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:      # fall through to the port, say why
            out = None
            sys.stderr.write("cpu_baseline: reference timing failed (%s); using the port\n" % (e,))
    if out is None:
        out = cpu_baseline_port(num_envs, T, full_minibatch)
    us = sim2sim_policy_step_us(200 if not full_minibatch else 2000)
    out["sim2sim"] = dict(policy_step_us=us, env_steps_per_s=1e6 / us, envs=1, cores=1, kind="port",
                          sample="BASELINE configs[0], the half that runs without MuJoCo (absent): observation frame + 15-frame history "
                                 "+ TorchScript actor + action clip per 100 Hz control step (reference scripts/sim2sim.py:124-150)")
    return out


# ------------------------------------------------------------------------------------------------ GPU runs
def collect_pmc_traffic(num_envs=4096):
    """roofline.traffic measured in this run: rocprofv3 --kernel-trace --pmc <counter> (one pass per counter, as
    MI355X_MICROARCH.md's HBM section prescribes) around tools/traffic_run.py -- a 256 MiB streaming kernel for the FETCH_SIZE
    calibration, then 4 iterations of this bench's headline workload -- converted by tools/pmc_to_json.py.  Returns the dict of
    profiles/pmc_traffic.json's shape, or None (with the reason on stderr)."""
    import glob
    import shutil
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        sys.stderr.write("--pmc: rocprofv3 not found\n")
        return None
    tmp = tempfile.mkdtemp(prefix="hgym_pmc_")
    csvs = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(tmp, ctr)
        try:
            r = subprocess.run([rp, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "t", "--", sys.executable,
                                os.path.join(ROOT, "tools", "traffic_run.py")], cwd=tmp,
                               env=dict(os.environ, TMPDIR=tmp, HGYM_TRAFFIC_ENVS=str(num_envs)),
                               capture_output=True, text=True, timeout=300)
        except subprocess.TimeoutExpired:
            sys.stderr.write("--pmc: %s pass exceeded its 300 s limit; quoting the committed file instead\n" % ctr)
            return None
        found = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not found:
            sys.stderr.write("--pmc: %s pass failed (rc %d): %s\n" % (ctr, r.returncode, r.stderr[-300:]))
            return None
        csvs[ctr] = found[0]
    out = os.path.join(tmp, "pmc_traffic.json")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_to_json
    with contextlib.redirect_stdout(io.StringIO()):
        pmc_to_json.main(csvs["FETCH_SIZE"], csvs["WRITE_SIZE"], out, float(1 << 28), num_envs)
    res = json.load(open(out))
    shutil.rmtree(tmp, ignore_errors=True)
    return res


def _roofline(L, runner, elapsed_per_iter_ms, precision, n_profiled_iters, traffic_ok=True, pmc=None, pmc_only=False):
    mfma_peak = MFMA_BF16_PEAK_TFLOPS if precision == "bf16" else 157.3
    classes = [  # (class id, kernel, bound)
        (L.PROF_ROLLOUT, "rollout_step_kernel", "hbm"),      # policy act + env step + previous finaliser, one launch per vec-step
        (L.PROF_ENV_STEP, "env_step_kernel", "hbm"), (L.PROF_MLP_FWD, "mlp_fb_kernel", "mfma"),        # the update: forward + loss + dZ chain of a 64-row tile
        (L.PROF_POLICY, "mlp_fwd_kernel<32>", "mfma"),
        (L.PROF_MLP_BWD, "mlp_bwd_kernel", "mfma"), (L.PROF_DW, "dw_kernel", "mfma"), (L.PROF_GEMM, "gemm_nt_kernel", "mfma"),
        (L.PROF_LOSS, "ppo_loss_kernel", "hbm"), (L.PROF_REDUCE, "reduce_slabs_kernel", "hbm"),
        (L.PROF_APPLY, "sqnorm+adam_kernel", "hbm"), (L.PROF_GAE, "gae_kernel", "hbm")]
    traffic, source, counters = {}, None, {}
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")     # HBM bytes per launch from rocprofv3 --pmc passes
    committed = os.path.exists(tpath) and not pmc_only     # pmc_only: another size than the committed file's -- only this run's own passes count
    if committed and traffic_ok:                    # SQ / TCC counter passes are not repeated in the run: quoted from the committed file
        counters = json.load(open(tpath)).get("counters", {})
    if pmc is not None and traffic_ok:
        traffic = pmc.get("kernels", {})
        counters = pmc.get("counters", counters)
        source = dict(kind="measured in this run", how="rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes (bench.py --pmc)",
                      fetch_calibration=pmc.get("fetch_calibration"), variants=pmc.get("variants"))
    elif committed and traffic_ok:                 # the counters were collected on the headline workload: they say nothing about another size
        tj = json.load(open(tpath))
        traffic = tj.get("kernels", {})
        counters = tj.get("counters", {})
        # NOT a measurement of this run: a committed file, quoted with where it came from so that the reader can tell
        source = dict(kind="committed file, not measured in this run", path="profiles/pmc_traffic.json", **tj.get("provenance", {}))
    ks = []
    for cid, name, bound in classes:
        n, ms, work = L.prof_summary(cid)
        if n == 0 or ms <= 0:
            continue
        peak, unit, scale = (HBM_PEAK_GBS, "GB/s", 1e9) if bound == "hbm" else (mfma_peak, "TFLOP/s", 1e12)
        ach = work / (ms * 1e-3) / scale
        tr = traffic.get(name)
        if name == "mlp_fwd_kernel<32>" and traffic.get("mlp_fwd_kernel") is not None:
            # the policy class holds the 64-row form too where the critic runs once behind the rollout (8192 envs: four passes of it + the
            # 32-row bootstrap launch): launch-weighted mean of the two instantiations' bytes, as avg_launch_us is the mean of their times
            det = (pmc or {}).get("detail", {})
            a, b = det.get("mlp_fwd_kernel<32>"), det.get("mlp_fwd_kernel")
            if a and b:
                tr = (a["hbm_bytes"] * a["launches"] + b["hbm_bytes"] * b["launches"]) / (a["launches"] + b["launches"])
            elif tr is None:
                tr = traffic.get("mlp_fwd_kernel")
        ks.append(dict(kernel=name, bound=bound, launches_per_iter=n // n_profiled_iters, avg_launch_us=ms / n * 1e3, achieved=ach,
                       peak=peak, unit=unit, frac=ach / peak, share_of_iteration=ms / n_profiled_iters / elapsed_per_iter_ms,
                       traffic=tr, traffic_source=source if tr else None,
                       # rocprofv3 SQ / TCC passes (tools/gpu_counters.sh), same provenance as `traffic`: MFMA-pipe occupancy
                       # (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)) and L2 hit rate of this kernel
                       mfma_busy=counters.get(name, {}).get("mfma_busy"), l2_hit=counters.get(name, {}).get("l2_hit"),
                       counters_source=("profiles/pmc_traffic.json (rocprofv3 SQ / TCC passes, tools/gpu_counters.sh; committed file, not "
                                        "measured in this run)" if counters.get(name) else None),
                       # counter traffic per launch over this run's launch time, as a fraction of the HBM peak: how close the kernel
                       # is to being bound by the bytes it actually moves, whatever its algorithmic bound says
                       traffic_frac_of_hbm_peak=(tr / (ms / n * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr else None))
    ks.sort(key=lambda k: -k["share_of_iteration"])
    return ks


def run_config(args, task, num_envs, rank, world, local, dist, steps, warmup, log_root=None, want_roofline=True, quiet=False):
    """Builds env + runner for one configuration, W warm-up iterations, K timed ones (barrier + synchronize on both sides, max
    over ranks), optionally two more eager iterations under the library's HIP-event hooks for the roofline."""
    from humanoid.envs import task_registry
    from humanoid.utils import get_args
    from hgym import _lib as L
    dev = "cuda:%d" % local
    a = get_args(["--task=" + task, "--headless", "--num_envs", str(num_envs), "--sim_device", dev, "--rl_device", dev,
                  "--seed", str(5 + rank)])
    sink = io.StringIO() if (quiet or rank != 0) else None
    with (contextlib.redirect_stdout(sink) if sink is not None else contextlib.nullcontext()):
        env, env_cfg = task_registry.make_env(name=a.task, args=a)
        runner, train_cfg = task_registry.make_alg_runner(env=env, name=a.task, args=a, log_root=log_root)
    T, N = runner.num_steps_per_env, env.num_envs

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed warm-up: W iterations, but never fewer than 2 -- the first iteration runs the rollout eagerly, the second captures it
    # into the HIP graph the timed iterations replay (a capture inside the timed region would not be the steady state)
    with (contextlib.redirect_stdout(io.StringIO()) if log_root is not None else contextlib.nullcontext()):
        runner.learn(num_learning_iterations=max(warmup, 2), init_at_random_ep_len=True)
        if hasattr(runner, "wait_for_saves"):
            # the warm-up's checkpoints (model_0.pt, model_<W>.pt) belong to the warm-up: with the background writer (HGYM_ASYNC_SAVE, default
            # on) the process's FIRST torch.save -- ~70 ms of cold pickling in the writer thread -- would otherwise run under the timed
            # iterations and take the interpreter lock from the launching thread (18.6 instead of 6.4 ms/iteration over 6 iterations,
            # profiles/r06_async_checkpoint_default.txt); a run of 150 iterations with three checkpoints inside shows no such cost
            runner.wait_for_saves()
        barrier()
        t0 = time.perf_counter()
        # K learning iterations, enqueued back to back (the runner reads nothing back between iterations when it does not log)
        save0 = getattr(runner, "save_time_s", 0.0)
        runner.learn(num_learning_iterations=steps, init_at_random_ep_len=False)
        coll, learn = runner.last_collection_time, runner.last_learn_time     # per iteration (HIP events / host clock when logging)
        barrier()
        elapsed = time.perf_counter() - t0
        ckpt_s = getattr(runner, "save_time_s", 0.0) - save0                  # checkpoint time of the TRAINING thread inside the timed region (logging runs)
        # the final checkpoint is handed to the background writer (HGYM_ASYNC_SAVE, default on): learn() does not wait for the file.  What
        # the writer still needs is measured here, outside the timed region, and reported next to the value
        t1 = time.perf_counter()
        if hasattr(runner, "wait_for_saves"):
            runner.wait_for_saves()
        ckpt_wait_s = time.perf_counter() - t1
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax)
    res = dict(value=T * N * world * steps / elapsed, ms_per_step=elapsed / steps * 1e3, ppo_update_ms=learn * 1e3,
               collection_ms=coll * 1e3, T=T, N=N, obs=env.num_obs, priv=env.num_privileged_obs)
    it_ms = sorted(getattr(runner, "last_iteration_ms", None) or [])
    it_ms_timed = list(it_ms)                   # (the roofline leg below calls learn() again: keep the timed call's own list)
    update_graph = getattr(runner, "_update_graph", None) is not None     # compute_returns() + update() replayed from the second HIP graph
    res["update_graph"] = update_graph
    if len(it_ms) == steps and log_root is None:
        # per-iteration device times of the timed call (HIP events on the launch stream, read after the region): the contract's `value` is
        # K steps over the region's wall time (a mean); SURVEY 8(d) asks for the median of >= 20 iterations -- both are in the line
        med = it_ms[len(it_ms) // 2] if len(it_ms) % 2 else 0.5 * (it_ms[len(it_ms) // 2 - 1] + it_ms[len(it_ms) // 2])
        res["iteration_ms"] = dict(median=med, mean=sum(it_ms) / len(it_ms), min=it_ms[0], max=it_ms[-1], n=len(it_ms),
                                   value_at_median=T * N * world / (med * 1e-3), source="HIP events per iteration, this rank")
    if log_root is not None:
        res["checkpoint_ms_total"] = ckpt_s * 1e3
        res["value_without_checkpoints"] = T * N * world * steps / max(elapsed - ckpt_s, 1e-9)
        res["final_checkpoint_wait_ms_after_learn"] = ckpt_wait_s * 1e3
        res["value_including_final_checkpoint_wait"] = T * N * world * steps / (elapsed + ckpt_wait_s)
        res["async_save"] = os.environ.get("HGYM_ASYNC_SAVE", "1") != "0"
    if want_roofline:
        # live per-kernel timing with HIP events on the launch stream (hgym_prof_*).  Events cannot be recorded inside a
        # replayed HIP graph, so these two iterations run the rollout eagerly (same kernels, same launch order).  EVERY rank
        # runs them (the update contains the gradient all-reduce); only rank 0 records and reports.
        os.environ["HGYM_GRAPH"] = "0"
        if rank == 0:
            L.lib.hgym_prof_enable(1)
        runner.alg.comm_timing = [] if world > 1 else None
        with (contextlib.redirect_stdout(io.StringIO()) if log_root is not None else contextlib.nullcontext()):
            # two eager iterations; with several ranks the second one runs the OTHER gradient exchange (the direct kernel over peer
            # mappings vs the collective) when the peer-mapped buffer exists, so that the line carries both
            runner.learn(num_learning_iterations=1, init_at_random_ep_len=False)
            both = world > 1 and getattr(runner.alg, "_comm", None) is not None
            runner.alg.comm_flip = both
            runner.learn(num_learning_iterations=1, init_at_random_ep_len=False)
            runner.alg.comm_flip = False
        torch.cuda.synchronize()
        os.environ["HGYM_GRAPH"] = "1"
        if rank == 0:
            pmc_by_envs = getattr(args, "pmc_by_envs", None) or {}
            std = task == "humanoid_ppo" and args.precision == "bf16"
            res["kernels"] = _roofline(L, runner, res["ms_per_step"], args.precision, 2,
                                       traffic_ok=std and (num_envs == 4096 or num_envs in pmc_by_envs),
                                       pmc=pmc_by_envs.get(num_envs) if num_envs != 4096 else getattr(args, "pmc_result", None),
                                       pmc_only=num_envs != 4096)
            L.lib.hgym_prof_enable(0)
        if world > 1 and runner.alg.comm_timing:
            ev = runner.alg.comm_timing
            nmb = runner.alg.num_learning_epochs * runner.alg.num_mini_batches
            # what 0.9 weak-scaling efficiency leaves for the exchange: T_N <= T_1 / 0.9 with T_1 ~ this run's iteration minus its exposed waits
            p2p_on = bool(getattr(runner.alg, "_comm_p2p", False))
            names = {"p2p": "direct reduce-scatter + all-gather kernel over hipIpc peer mappings of the ranks' gradient buffers (HGYM_COMM=auto picked it / p2p)",
                     "collective": ("RCCL all-reduce (torch.distributed 'nccl')" if dist.get_backend() == "nccl" else
                                    "%s all-reduce (ranks may share a GPU; host-staged, not representative of RCCL over xGMI)" % dist.get_backend())}
            per = {}
            for tag in ("p2p", "collective"):
                xs = [a_.elapsed_time(b_) * 1e3 for a_, b_, t_ in ev if t_ == tag]
                if xs:
                    per[tag] = dict(backend=names[tag], minibatches_timed=len(xs), exposed_us_per_minibatch=sum(xs) / len(xs), exposed_us_max=max(xs))
            used = "p2p" if p2p_on else "collective"
            exposed_used = per[used]["exposed_us_per_minibatch"] if used in per else 0.0
            t1_ms = max(res["ms_per_step"] - exposed_used * nmb * 1e-3, 1e-6)
            budget = t1_ms * (1.0 / 0.9 - 1.0) / nmb * 1e3
            rep = getattr(runner.alg, "comm_report", {}) or {}
            comm = dict(collective="SUM of [flat fp32 gradient | minibatch KL], one exchange per minibatch, fully exposed by construction",
                        mode=rep.get("mode"), used_in_timed_run=used, fallback_reason=rep.get("fallback_reason"), startup_probe=rep.get("probe"),
                        host_placement=getattr(args, "host_placement", None), bytes_per_minibatch=4 * (runner.alg.net.P + 1), minibatches_per_iter=nmb,
                        exposed_us_per_minibatch=exposed_used, exposed_us_max=per[used]["exposed_us_max"] if used in per else 0.0,
                        exposed_ms_per_iter=exposed_used * nmb * 1e-3, backend=names[used],
                        budget_us_per_minibatch_for_0p9_weak_scaling=budget, within_budget=bool(exposed_used <= budget),
                        backends=per,
                        note="stream time between the last backward kernel and the start of hgym_ppo_apply (HIP events on the compute stream, "
                             "one eager profiling iteration per backend, rank 0); includes waiting for the slowest rank.  budget = what an "
                             "efficiency of 0.9 leaves per minibatch: (iteration - exposed) x (1 / 0.9 - 1) / minibatches")
            # SURVEY 8(e)'s parity target for N > 1: exact equality of the parameters across ranks after every step.  Checked here after all the
            # timed + profiling iterations: a 64-bit sum of the parameter bits and the learning rate of every rank, gathered and compared
            # (a stale cache line or a lost exchange in either backend would show up as a difference)
            try:
                pbits = runner.alg.net.params.view(torch.int32).to(torch.int64)
                sig = torch.stack([pbits.sum(), (pbits * (torch.arange(pbits.numel(), device=pbits.device) % 8191 + 1)).sum(),
                                   torch.tensor(runner.alg.learning_rate, dtype=torch.float64, device=pbits.device).view(torch.int64)])
                sigs = [torch.zeros_like(sig) for _ in range(world)]
                dist.all_gather(sigs, sig)
                comm["replicas_identical_after_run"] = bool(all(torch.equal(x, sigs[0]) for x in sigs))
                comm["optimizer_steps"] = int(float(runner.alg.net.opt_state[1]))
            except Exception as e:      # noqa: BLE001
                comm["replicas_check_error"] = str(e)
            # every rank's own iteration times of the timed call (HIP events on its stream): a rank that is slow on the host or on the device
            # shows up here, next to the exchange's exposed wait (which contains the wait for the slowest rank)
            try:
                it_ms = it_ms_timed or [0.0]
                mine = dict(rank=rank, min=it_ms[0], median=it_ms[len(it_ms) // 2], max=it_ms[-1], n=len(it_ms), update_graph=update_graph)
                allr = [None] * world
                dist.all_gather_object(allr, mine)
                comm["per_rank_iteration_ms"] = allr
                comm["slowest_over_fastest_rank_median"] = max(r_["median"] for r_ in allr) / max(min(r_["median"] for r_ in allr), 1e-9)
            except Exception as e:      # noqa: BLE001
                comm["per_rank_error"] = str(e)
            c = getattr(runner.alg, "_comm", None)
            if c is not None:
                try:
                    w_us, x_us = c.check()
                    comm["p2p_last_call"] = dict(wait_for_slowest_rank_us=w_us, exchange_us=x_us,
                                                 note="the direct kernel's own 100 MHz timestamps: start -> every rank arrived (rank skew), -> every shard delivered")
                except Exception as e:      # noqa: BLE001
                    comm["p2p_error"] = str(e)
            res["comm"] = comm
        runner.alg.comm_timing = None
    del runner, env
    torch.cuda.empty_cache()
    return res


def _workload(task, N, T):
    return ("XBot-L PPO %d envs/GPU, synthetic physics step, T=%d, 2 epochs x 4 minibatches (BASELINE configs[%s])"
            % (N, T, "4: + denoising head" if task == "humanoid_dwl_ppo" else "3: history stack 15, 8192 envs/GPU" if N == 8192 else "1"))


def _roofline_obj(ks, pick=None):
    dom = ks[0] if pick is None else next((k for k in ks if k["kernel"] == pick), ks[0])
    rest = [k for k in ks if k is not dom]
    return dict(bound=dom["bound"], achieved=dom["achieved"], peak=dom["peak"], unit=dom["unit"], frac=dom["frac"],
                traffic=dom["traffic"], traffic_source=dom.get("traffic_source"), traffic_frac_of_hbm_peak=dom.get("traffic_frac_of_hbm_peak"),
                mfma_busy=dom.get("mfma_busy"), l2_hit=dom.get("l2_hit"),
                kernel=dom["kernel"], launches_per_iter=dom["launches_per_iter"],
                avg_launch_us=dom["avg_launch_us"], share_of_iteration=dom["share_of_iteration"], kernels=rest)


def place_rank_on_host(local, world):
    """N > 1: eight Python launch threads on one host are the rank-skew source SURVEY 8(e) warns about.  Each rank gets its own block
    of the cores this process may run on (no migration across ranks' blocks, no two launch threads on one core) and one intra-op
    thread (the hot path is on the GPU; torch's CPU pool only adds wake-ups).  HGYM_PIN=0 leaves the affinity alone."""
    torch.set_num_threads(1)
    if os.environ.get("HGYM_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return dict(threads=1, pinned=False)
    try:
        cores = sorted(os.sched_getaffinity(0))
        per = len(cores) // max(world, 1)
        if per < 2:
            return dict(threads=1, pinned=False, note="fewer than 2 cores per rank")
        per = min(per, 16)
        mine = cores[local * per:(local + 1) * per]
        # every thread that exists already (the HIP runtime's, torch's pools) as well as the calling one: sched_setaffinity(0, ..) alone
        # moves only the caller, and threads created from now on inherit ITS mask -- which is why main() calls this before
        # the process group (RCCL / gloo helper threads) exists
        tids = [int(t) for t in os.listdir("/proc/self/task")] if os.path.isdir("/proc/self/task") else [0]
        moved = 0
        for tid in tids:
            try:
                os.sched_setaffinity(tid, mine)
                moved += 1
            except OSError:
                pass
        os.sched_setaffinity(0, mine)
        return dict(threads=1, pinned=True, cores=[mine[0], mine[-1]], cores_per_rank=per, threads_moved=moved, before_process_group=True)
    except OSError as e:
        return dict(threads=1, pinned=False, note=str(e))


def compact_summary(out, head, extra, world):
    """The block the JSON line ENDS with (< 1.5 KB): a reader that keeps only the tail of the line (the driver's record keeps 2 000 bytes) still
    has the headline's split, the big kernels and the baselines.  Everything in it repeats a field of the line."""
    summ = dict(value=head["value"], n_gpus=world, ms_per_step=head["ms_per_step"], collection_ms=head["collection_ms"],
                ppo_update_ms=head["ppo_update_ms"])
    if head.get("iteration_ms"):
        summ["iteration_ms_median"] = round(head["iteration_ms"]["median"], 4)
        summ["value_at_median"] = round(head["iteration_ms"]["value_at_median"])
    if head.get("kernels"):
        big = {}
        for k in head["kernels"]:
            if k["kernel"] in ("mlp_fb_kernel", "dw_kernel", "rollout_step_kernel", "env_step_kernel", "mlp_fwd_kernel<32>"):
                big[k["kernel"]] = dict(us=round(k["avg_launch_us"], 2), n=k["launches_per_iter"], frac=round(k["frac"], 4), bound=k["bound"],
                                        traffic_MB=None if not k.get("traffic") else round(k["traffic"] / 1e6, 1),
                                        share=round(k["share_of_iteration"], 3))
        summ["kernels"] = big
    if out.get("cpu_baseline"):
        summ["cpu_baseline"] = dict(value=out["cpu_baseline"]["value"], cores=out["cpu_baseline"]["cores"], kind=out["cpu_baseline"]["kind"])
    if extra:
        summ["configs"] = {e["name"]: dict(value=round(e["value"]), collection_ms=round(e["collection_ms"], 3),
                                           ppo_update_ms=round(e["ppo_update_ms"], 3)) for e in extra}
    if head.get("comm"):
        c = head["comm"]
        summ["comm"] = dict(used=c.get("used_in_timed_run"), fallback_reason=c.get("fallback_reason"),
                            exposed_us_per_minibatch=c.get("exposed_us_per_minibatch"), within_budget=c.get("within_budget"),
                            replicas_identical_after_run=c.get("replicas_identical_after_run"))
    return summ


def spawn_ranks(args):
    """`python bench.py --gpus N` without a torch.distributed environment: run the same command line under
    torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    n_dev = torch.cuda.device_count()
    if os.environ.get("HGYM_DIST_BACKEND", "nccl") == "nccl" and n_dev < args.gpus:
        sys.stderr.write("bench.py: --gpus %d but only %d device(s) visible; RCCL needs one device per rank\n" % (args.gpus, n_dev))
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # host placement FIRST (LOCAL_RANK / WORLD_SIZE are in the environment): whatever creates threads below inherits the rank's core block
    host_placement = place_rank_on_host(local % max(world, 1), world) if world > 1 else None
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
    args.host_placement = host_placement
    os.environ["HGYM_PRECISION"] = args.precision
    from humanoid.algo import PPO
    PPO.precision = args.precision

    args.pmc_result = None
    want_pmc = (args.pmc or not args.no_pmc) and os.environ.get("HGYM_BENCH_PMC", "1") != "0"
    if (want_pmc and world == 1 and not args.no_roofline and not os.environ.get("HGYM_BENCH_CHILD") and args.task == "humanoid_ppo"
            and args.num_envs == 4096 and args.precision == "bf16"):
        os.environ["HGYM_BENCH_CHILD"] = "1"            # tools/traffic_run.py runs this file again: not recursively
        args.pmc_result = collect_pmc_traffic()
        args.pmc_by_envs = {}
        if "envs8192" in args.configs.split(","):
            # BASELINE configs[3] (8192 envs, SURVEY 8d's HBM stress): its own counter passes -- the launches differ (no critic tiles, one
            # critic pass behind the rollout, gae_kernel<BOOT>), so the 4096-env bytes say nothing about them
            r8 = collect_pmc_traffic(8192)
            if r8 is not None:
                args.pmc_by_envs[8192] = r8
        del os.environ["HGYM_BENCH_CHILD"]
    head = run_config(args, args.task, args.num_envs, rank, world, local, dist, args.steps, args.warmup,
                      want_roofline=not args.no_roofline)
    T, N = head["T"], head["N"]
    extra = []
    want = [c for c in args.configs.split(",") if c and c != "none"]
    if world == 1 and want and args.task == "humanoid_ppo" and args.num_envs == 4096:
        k_steps, k_warm = min(args.steps, 6), min(args.warmup, 2)
        for c in want:
            if c == "envs8192":
                r = run_config(args, "humanoid_ppo", 8192, 0, 1, local, None, k_steps, k_warm, want_roofline=not args.no_roofline, quiet=True)
                e = dict(name="envs8192", logging=False)
            elif c == "logging":
                with tempfile.TemporaryDirectory() as tmp:
                    r = run_config(args, "humanoid_ppo", 4096, 0, 1, local, None, k_steps, k_warm, log_root=tmp, want_roofline=False, quiet=True)
                e = dict(name="logging_on", logging=True,
                         note="train.py's default: log_dir set -> episode statistics kept by the step finaliser, loss / episode statistics copied to "
                              "pinned host memory behind every update and printed (console, + TensorBoard if installed) while the next "
                              "iteration runs, checkpoints inside the timed region")
            elif c == "dwl":
                r = run_config(args, "humanoid_dwl_ppo", 4096, 0, 1, local, None, k_steps, k_warm, want_roofline=False, quiet=True)
                e = dict(name="dwl_head", logging=False, note="BASELINE configs[4] on one GPU; parity of the head is unpinned (no reference code)")
            elif c == "fp32":
                if args.precision == "f32":
                    continue
                # the headline workload at the REFERENCE's own arithmetic (fp32 end to end, actor_critic.py:53-80): context beside
                # the bf16 headline BASELINE.json names; the generic layer-by-layer path (gemm_nt_kernel<float>), same env kernels
                PPO.precision = "f32"
                os.environ["HGYM_PRECISION"] = "f32"
                try:
                    r = run_config(args, "humanoid_ppo", 4096, 0, 1, local, None, k_steps, k_warm, want_roofline=False, quiet=True)
                finally:
                    PPO.precision = args.precision
                    os.environ["HGYM_PRECISION"] = args.precision
                e = dict(name="fp32", logging=False, dtype="f32",
                         note="the headline workload with fp32 dense layers (f32-input MFMA, 157 TFLOP/s peak) -- the reference's own "
                              "arithmetic; env / GAE / loss / Adam are fp32 in both")
            else:
                continue
            if "checkpoint_ms_total" in r:
                e.update(checkpoint_ms_total=r["checkpoint_ms_total"], value_without_checkpoints=r["value_without_checkpoints"],
                         final_checkpoint_wait_ms_after_learn=r["final_checkpoint_wait_ms_after_learn"],
                         value_including_final_checkpoint_wait=r["value_including_final_checkpoint_wait"], async_save=r["async_save"],
                         checkpoint_note="`value`: learn() of K iterations incl. what its checkpoint (the final model_<it>.pt) costs the training "
                                         "thread and the device (snapshot to pinned host memory behind the update); the file itself is written by a "
                                         "background thread (HGYM_ASYNC_SAVE, default on) that learn() does not wait for -- "
                                         "final_checkpoint_wait_ms_after_learn is how long the bench then waited for it, "
                                         "value_including_final_checkpoint_wait charges that wait to the run; value_without_checkpoints = the run "
                                         "with the training thread's checkpoint time taken out")
            e.update(workload=_workload("humanoid_dwl_ppo" if c == "dwl" else "humanoid_ppo", r["N"], r["T"]), envs_per_gpu=r["N"],
                     value=r["value"], unit="env-steps/s", steps=k_steps, warmup=k_warm, ms_per_step=r["ms_per_step"],
                     collection_ms=r["collection_ms"], ppo_update_ms=r["ppo_update_ms"])
            if r.get("kernels"):
                # configs[3] is the HBM stress of the env step: report the roofline of the kernel that contains it -- since round 5 the fused
                # rollout launch without critic tiles (the critic runs once behind the rollout: "mlp_fwd_kernel<32>" class in kernels[])
                e["roofline"] = _roofline_obj(r["kernels"], pick="rollout_step_kernel" if c == "envs8192" else None)
            extra.append(e)

    n1 = None
    if dist is not None and world > 1 and os.environ.get("HGYM_BENCH_N1_REF", "1") != "0":
        # the same workload on ONE rank of this very job (rank 0 alone, the others idle at the barrier): what the N-rank value is to be
        # divided by -- same box, same minute, same library (the driver computes its own efficiency from separate runs; this one is printed
        # so that a single N-rank call is self-contained)
        if rank == 0:
            os.environ["HGYM_DIST_OFF"] = "1"
            try:
                r1 = run_config(args, args.task, args.num_envs, 0, 1, local, None, min(args.steps, 10), min(args.warmup, 3), want_roofline=False, quiet=True)
                n1 = dict(value=r1["value"], ms_per_step=r1["ms_per_step"], steps=min(args.steps, 10),
                          weak_scaling_efficiency_in_this_call=head["value"] / (world * r1["value"]),
                          note="rank 0 alone (HGYM_DIST_OFF=1) right after the %d-rank run, the other ranks idle" % world)
            except Exception as e:      # noqa: BLE001
                n1 = dict(error=str(e))
            del os.environ["HGYM_DIST_OFF"]
    if dist is not None:
        dist.barrier()
    if rank == 0:
        out = {
            "metric": "env-steps/s (XBot-L PPO, whole job)", "value": head["value"], "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": _workload(args.task, N, T),
                       "envs_per_gpu": N, "steps_per_env": T, "obs": head["obs"], "privileged_obs": head["priv"],
                       "minibatch": T * N // 4, "parallelism": "dp%d (env shards, RCCL grad all-reduce)" % world, "logging": False},
            "ppo_update_ms": head["ppo_update_ms"], "collection_ms": head["collection_ms"],
        }
        out["launches"] = ("two HIP-graph replays per iteration (rollout: 60 launches; compute_returns + update: ~45 launches)" if head.get("update_graph")
                           else "rollout replayed from a HIP graph, update issued from Python (HGYM_GRAPH_UPDATE=0 or not capturable)")
        if head.get("iteration_ms"):
            out["iteration_ms"] = head["iteration_ms"]      # per-iteration HIP-event times of the timed call: median / mean / min / max
        if head.get("kernels"):
            out["roofline"] = _roofline_obj(head["kernels"])
        if head.get("comm"):
            out["comm"] = head["comm"]
            if n1 is not None:
                out["comm"]["n1_reference"] = n1
        if extra:
            out["configs"] = extra
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline leg belongs to the N=1 run only
            out["cpu_baseline"] = cpu_baseline(N, T)
        out["summary"] = compact_summary(out, head, extra, world)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
