"""-m gpu: the N>1 training path end to end on ONE GPU -- two ranks (processes) share cuda:0 and talk over gloo (RCCL refuses
two ranks on one device; the collectives are backend-agnostic torch.distributed calls).  Checks what SURVEY.md §8e asks of the
multi-GPU path: parameters identical on every rank after every update (same averaged gradient, same adaptive-KL learning
rate), the graph-captured rollout and the asynchronous iteration loop work with a process group alive, and nothing
dead-locks (every rank issues the same sequence of collectives)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _all_done_or_timed_out(out_dir, world, prefix, limit_s=60.0):
    """File rendezvous in front of the final collective barrier: True when every rank has left its result file, False as soon as
    any rank has left a time-out marker (the caller then exits without touching the process group: a rank that bailed out of
    the direct exchange will never reach the barrier) or after limit_s."""
    import time
    t0 = time.time()
    while time.time() - t0 < limit_s:
        names = os.listdir(out_dir)
        if any(n.startswith("timeout") for n in names):
            return False
        if all(("%s%d.pt" % (prefix, r)) in names for r in range(world)):
            return True
        time.sleep(0.05)
    return False


def _worker(rank, world, port, out_dir, backend="gloo", num_envs=256, iters=4, comm="rccl", inject="", timing=True):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "humanoid-gym_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["HGYM_COMM"] = comm
    os.environ["HGYM_COMM_FAIL_INJECT"] = inject
    if backend == "nccl":            # RCCL: one device per rank
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
        dev = "cuda:%d" % rank
    else:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = "cuda:0"
    from humanoid.algo import PPO
    PPO.precision = "bf16"
    from humanoid.envs import task_registry
    from humanoid.utils import get_args
    args = get_args(["--task=humanoid_ppo", "--headless", "--num_envs", str(num_envs), "--seed", str(5 + rank), "--sim_device", dev,
                     "--rl_device", dev])
    env, _ = task_registry.make_env(name=args.task, args=args)
    runner, _ = task_registry.make_alg_runner(env=env, name=args.task, args=args, log_root=None)
    assert runner.alg._world == world
    p_init = runner.alg.net.params.clone()
    friction, commands0, seed = env.env_frictions.clone().cpu(), env.commands.clone().cpu(), int(env._ncfg.seed)
    if timing:          # (event pairs around every exchange: keeps the update eager -- PPO.update_capturable)
        runner.alg.comm_timing = []
    runner.learn(num_learning_iterations=iters, init_at_random_ep_len=True)     # eager, capture + replay, replay, replay
    torch.cuda.synchronize()
    net = runner.alg.net
    p2p = None
    # the gradient vector lives in the peer-mapped buffer when the direct exchange is in use (HGYM_COMM=p2p, or auto when its start-up
    # probe picked it: always over gloo, whose all-reduce is host-staged) or asked for beside the collective (both)
    rep = dict(runner.alg.comm_report)
    want_p2p = comm == "p2p" or (comm == "auto" and not inject)
    assert runner.alg._comm_p2p == want_p2p and (runner.alg._comm is not None) == (want_p2p or comm == "both"), rep
    assert rep["mode"] == comm and rep["used"] == ("p2p" if want_p2p else "collective"), rep
    if comm == "auto":
        assert (rep["fallback_reason"] is None) == (not inject), rep
        if not inject:
            assert rep["probe"]["p2p_us_per_call"] < rep["probe"]["collective_us_per_call"], rep
    if runner.alg._comm is not None:
        assert net.grads_ext.data_ptr() == runner.alg._comm.data.data_ptr()
        try:
            t = runner.alg._comm.check()        # raises if any bounded wait expired; (wait for the slowest rank, exchange) of the last call, us
        except Exception as e:                  # noqa: BLE001 -- dist_utils.CommTimeout
            if type(e).__name__ != "CommTimeout":
                raise
            open(os.path.join(out_dir, "timeout%d" % rank), "w").write(str(e))
            os._exit(0)
        p2p = t if want_p2p else None
    else:
        assert not want_p2p
    torch.save(dict(p_init=p_init.cpu(), params=net.params.cpu(), lr=float(net.opt_state[0]), steps=float(net.opt_state[1]),
                    obs=runner.alg.storage._obs_all[1].cpu(), graph=runner._graph is not None, friction=friction, commands0=commands0,
                    env_seed=seed, comm_events=len(runner.alg.comm_timing or []), update_graph=runner._update_graph is not None,
                    comm_calls=(runner.alg._comm.seq if runner.alg._comm is not None else 0), split=net.bucket_split, P=net.P, p2p=p2p, report=rep),
               os.path.join(out_dir, "r%d.pt.tmp" % rank))
    os.replace(os.path.join(out_dir, "r%d.pt.tmp" % rank), os.path.join(out_dir, "r%d.pt" % rank))
    if want_p2p and not _all_done_or_timed_out(out_dir, world, "r"):
        os._exit(0)
    if runner.alg._comm is not None:
        runner.alg._comm.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_one_gpu_stay_in_lockstep(tmp_path):
    port = 29700 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = (torch.load(os.path.join(str(tmp_path), "r%d.pt" % i)) for i in range(2))
    assert torch.equal(a["p_init"], b["p_init"])                  # rank 0's initial parameters everywhere (broadcast)
    assert torch.isfinite(a["params"]).all() and not torch.equal(a["params"], a["p_init"])
    assert torch.equal(a["params"], b["params"])                  # bit-identical after 32 synchronised Adam steps
    assert a["lr"] == b["lr"] and a["steps"] == b["steps"] == 4 * 8
    assert a["graph"] and b["graph"]
    assert not torch.equal(a["obs"], b["obs"])                    # different env shards
    # every rank owns an independent env stream: Philox key, frictions and the first command draws all differ (helpers.shard_seed)
    assert a["env_seed"] != b["env_seed"] and a["env_seed"] == 5
    assert not torch.equal(a["friction"], b["friction"]) and not torch.equal(a["commands0"], b["commands0"])
    # the gradient exchange ran once per minibatch (one event pair each: 4 iterations x 8)
    assert a["comm_events"] == 32 and 0 < a["split"] < a["P"]


@pytest.mark.timeout(900)
def test_eight_ranks_one_gpu_stay_in_lockstep(tmp_path):
    """BASELINE configs[2]'s rank count (8 ranks, here 128 envs each sharing one GPU over gloo): after 3 iterations = 24 synchronised
    Adam steps every rank holds bit-identical parameters and the same learning rate, from 8 different env shards; the 24
    gradient exchanges and the per-iteration advantage-statistics all-reduce complete on all ranks (no dead-lock with the
    graph-captured rollout and the asynchronous iteration loop alive in 8 processes)."""
    port = 30100 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(8, port, str(tmp_path), "gloo", 128, 3), nprocs=8, join=True)
    r = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % i)) for i in range(8)]
    assert torch.isfinite(r[0]["params"]).all() and not torch.equal(r[0]["params"], r[0]["p_init"])
    for i in range(1, 8):
        assert torch.equal(r[i]["p_init"], r[0]["p_init"]) and torch.equal(r[i]["params"], r[0]["params"]), i
        assert r[i]["lr"] == r[0]["lr"] and r[i]["steps"] == 24 and r[i]["graph"]
        assert not torch.equal(r[i]["obs"], r[0]["obs"]) and r[i]["env_seed"] != r[0]["env_seed"]
        assert r[i]["comm_events"] == 24
    assert len({x["env_seed"] for x in r}) == 8


def _p2p_unit_worker(rank, world, port, out_dir, count, calls):
    """hgym_comm_allreduce on its own: every rank's vector is a function of (rank, call), so each rank can form the expected
    rank-ordered fp32 sum itself."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "humanoid-gym_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["HGYM_COMM"] = "p2p"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from humanoid.algo.ppo import dist_utils
    assert dist_utils.comm_mode() == "p2p"
    comm = dist_utils.P2PComm(count, "cuda:0")   # (PPO goes through make_comm: the same steps, each followed by an agreement over all ranks)
    handles = [None] * world
    dist.all_gather_object(handles, comm.handle())
    comm.connect(handles)
    dist.barrier()                               # every rank has mapped every buffer before anyone's first kernel stores into them
    assert comm.count % 4 == 0 and comm.count >= count

    def vec(r, k):
        g = torch.Generator().manual_seed(1000 * k + r)
        return (torch.randn(comm.count, generator=g) * (1.0 + r)).cuda()
    worst = 0.0
    times = []
    try:
        _p2p_unit_calls(comm, vec, rank, world, calls, times)
    except dist_utils.CommTimeout as e:
        open(os.path.join(out_dir, "timeout%d" % rank), "w").write(str(e))
        os._exit(0)              # (peers are stuck in the same bounded wait; no collective clean-up is possible)
    torch.save(dict(times=times, last=comm.data.cpu()), os.path.join(out_dir, "u%d.pt.tmp" % rank))
    os.replace(os.path.join(out_dir, "u%d.pt.tmp" % rank), os.path.join(out_dir, "u%d.pt" % rank))
    if not _all_done_or_timed_out(out_dir, world, "u"):
        os._exit(0)
    comm.close()
    dist.barrier()
    dist.destroy_process_group()


def _p2p_unit_calls(comm, vec, rank, world, calls, times):
    import time

    def sum_in_order(xs):       # python floats are doubles: the kernel's rank-ordered fp64 sum
        t = xs[0]
        for x in xs[1:]:
            t = t + x
        return t
    worst = 0.0
    for k in range(calls):
        comm.data.copy_(vec(rank, k))
        if k % 2 == 1:                           # uneven arrival: odd ranks dawdle on odd calls (the kernel waits for the slowest rank)
            torch.cuda.synchronize()
            if rank % 2 == 1:
                time.sleep(0.05)
        comm.allreduce()
        # header v9: three doubles through the same mappings (the advantage statistics' path), a function of (rank, call) as well
        st = torch.tensor([1.5 + rank + 0.25 * k, (rank + 1) * 1e-3 * (k + 1), 4096.0 * (k + 1)], dtype=torch.float64, device="cuda")
        comm.sum64(st)
        times.append(comm.check())
        want64 = [sum_in_order([1.5 + q + 0.25 * k for q in range(world)]), sum_in_order([(q + 1) * 1e-3 * (k + 1) for q in range(world)]),
                  sum_in_order([4096.0 * (k + 1)] * world)]
        assert st.tolist() == want64, (rank, k, st.tolist(), want64)
        want = vec(0, k)
        for q in range(1, world):
            want = want + vec(q, k)               # rank order, fp32: what the kernel forms
        assert torch.equal(comm.data, want), (rank, k, float((comm.data - want).abs().max()))
        worst = max(worst, float((comm.data - want).abs().max()))


def _skip_if_not_coscheduled(tmp_path, world):
    """The direct exchange needs every rank's kernel RUNNING at the same time.  With one rank per GPU that is a given; with eight
    processes sharing ONE device it is up to the hardware scheduler's queue rotation -- when a bounded wait expired there (the
    kernel's own 15 s limit, no hang), that is the test environment, not the protocol: skip.  Two and four ranks must work."""
    marks = [f for f in os.listdir(str(tmp_path)) if f.startswith("timeout")]
    if marks:
        msg = open(os.path.join(str(tmp_path), marks[0])).read()
        assert world > 4, msg
        pytest.skip("8 processes on one GPU were not co-scheduled (%s)" % msg)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_p2p_allreduce_kernel_sums_in_rank_order(tmp_path, world):
    """The direct gradient exchange (csrc/hgym_comm.hip, HGYM_COMM=p2p): `world` processes share cuda:0, their buffers are mapped into
    each other through hipIpcMemHandles, 6 calls on the flat-gradient size (926 106 floats) with uneven arrival: every rank ends
    with the rank-ordered fp32 sum, bit for bit, the same on all ranks; no bounded wait expires.  The reference has no multi-GPU
    path to match (/root/reference/humanoid/utils/helpers.py:207-212 is a dead flag)."""
    port = 30500 + (os.getpid() % 2000) + world
    mp.spawn(_p2p_unit_worker, args=(world, port, str(tmp_path), 926106, 6), nprocs=world, join=True)
    _skip_if_not_coscheduled(tmp_path, world)
    r = [torch.load(os.path.join(str(tmp_path), "u%d.pt" % i)) for i in range(world)]
    for i in range(1, world):
        assert torch.equal(r[i]["last"], r[0]["last"]), i
    print("p2p all-reduce, %d ranks on one GPU, 3.7 MB: (wait for the slowest rank, exchange) us per call on rank 0: %s"
          % (world, ["%.0f / %.0f" % t for t in r[0]["times"]]))


@pytest.mark.timeout(900)
def test_two_ranks_one_gpu_p2p_exchange_equals_collective(tmp_path):
    """The whole data-parallel run (graph-captured rollout, asynchronous iterations, 32 synchronised Adam steps) with the direct
    exchange in place of the all-reduce: for two ranks a + b = b + a exactly, so the parameters must equal the gloo run's bit for bit."""
    port = 30900 + (os.getpid() % 2000)
    os.makedirs(str(tmp_path / "p2p"))
    os.makedirs(str(tmp_path / "coll"))
    mp.spawn(_worker, args=(2, port, str(tmp_path / "p2p"), "gloo", 256, 4, "p2p"), nprocs=2, join=True)
    mp.spawn(_worker, args=(2, port + 1, str(tmp_path / "coll"), "gloo", 256, 4, "rccl"), nprocs=2, join=True)
    a, b = (torch.load(os.path.join(str(tmp_path / "p2p"), "r%d.pt" % i)) for i in range(2))
    c = torch.load(os.path.join(str(tmp_path / "coll"), "r0.pt"))
    assert torch.equal(a["params"], b["params"]) and a["lr"] == b["lr"] and a["steps"] == 32 and a["comm_events"] == 32
    assert torch.isfinite(a["params"]).all() and not torch.equal(a["params"], a["p_init"])
    assert torch.equal(a["params"], c["params"]) and a["lr"] == c["lr"]
    assert a["p2p"] is not None and c["p2p"] is None


@pytest.mark.timeout(900)
def test_two_ranks_one_gpu_captured_update_with_the_direct_exchange(tmp_path):
    """Header v9 / VERDICT r05 item 6: with the direct exchange nothing in the update needs the host -- the gradient exchange's call
    number and the permutation's draw number are read on the device, the advantage statistics travel through hgym_comm_sum64 -- so
    compute_returns() + update() are captured into the second HIP graph on every rank and the iteration is two graph launches per rank.
    Two ranks on one GPU, 5 iterations (eager, capture, three replays): both ranks hold bit-identical parameters, and they equal the run
    whose update is issued from Python (the eager p2p run, which in turn equals the gloo run: test above)."""
    port = 32100 + (os.getpid() % 2000)
    os.makedirs(str(tmp_path / "graph"))
    os.makedirs(str(tmp_path / "eager"))
    mp.spawn(_worker, args=(2, port, str(tmp_path / "graph"), "gloo", 256, 5, "p2p", "", False), nprocs=2, join=True)
    mp.spawn(_worker, args=(2, port + 1, str(tmp_path / "eager"), "gloo", 256, 5, "p2p"), nprocs=2, join=True)
    a, b = (torch.load(os.path.join(str(tmp_path / "graph"), "r%d.pt" % i)) for i in range(2))
    c = torch.load(os.path.join(str(tmp_path / "eager"), "r0.pt"))
    assert a["update_graph"] and b["update_graph"] and not c["update_graph"]
    assert torch.equal(a["params"], b["params"]) and a["lr"] == b["lr"] and a["steps"] == b["steps"] == 40
    assert torch.isfinite(a["params"]).all() and not torch.equal(a["params"], a["p_init"])
    assert torch.equal(a["params"], c["params"]) and a["lr"] == c["lr"]
    assert a["p2p"] is not None and a["comm_calls"] == c["comm_calls"]


@pytest.mark.timeout(900)
def test_auto_picks_the_direct_exchange_and_falls_back_on_injected_failures(tmp_path):
    """HGYM_COMM=auto (the default for N > 1): two ranks on one GPU over gloo.  Without a fault the start-up probe
    checks the direct kernel's sum, times both exchanges and picks the direct one (gloo's all-reduce is host-staged).  With an injected
    fault on rank 1 -- the allocation, the peer mapping, a rank whose first kernel never runs (the peers' bounded waits expire after
    2 s), a wrong sum -- EVERY rank falls back to the collective, says why, and trains on.  For two ranks a + b = b + a exactly, so all
    five runs must end with the same parameters, bit for bit, on both ranks."""
    port = 31700 + (os.getpid() % 2000)
    runs = [("auto", ""), ("auto", "alloc"), ("auto", "map"), ("auto", "timeout"), ("auto", "sum")]
    res = []
    for k, (comm, inject) in enumerate(runs):
        d = tmp_path / ("run%d" % k)
        os.makedirs(str(d))
        mp.spawn(_worker, args=(2, port + k, str(d), "gloo", 256, 2, comm, inject), nprocs=2, join=True)
        a, b = (torch.load(os.path.join(str(d), "r%d.pt" % i)) for i in range(2))
        assert torch.equal(a["params"], b["params"]) and a["lr"] == b["lr"] and a["steps"] == 16, (comm, inject)
        assert a["report"]["used"] == b["report"]["used"] == ("collective" if inject else "p2p"), (inject, a["report"], b["report"])
        if inject:
            assert a["report"]["fallback_reason"] and b["report"]["fallback_reason"], (inject, a["report"])
        res.append(a)
    words = dict(alloc="allocation", map="hipIpcOpenMemHandle", timeout="bounded wait", sum="wrong sum")
    for (comm, inject), r in zip(runs[1:], res[1:]):
        assert words[inject] in r["report"]["fallback_reason"], (inject, r["report"])
        assert torch.equal(r["params"], res[0]["params"]) and r["lr"] == res[0]["lr"], inject
    print("HGYM_COMM=auto start-up probe, 2 ranks on one GPU over gloo: %s" % (res[0]["report"]["probe"],))


@pytest.mark.timeout(900)
def test_eight_ranks_one_gpu_p2p_exchange_stays_in_lockstep(tmp_path):
    """BASELINE configs[2]'s rank count over the direct exchange: 8 processes share one GPU, 24 exchanges; bit-identical parameters."""
    port = 31300 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(8, port, str(tmp_path), "gloo", 128, 3, "p2p"), nprocs=8, join=True)
    _skip_if_not_coscheduled(tmp_path, 8)
    r = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % i)) for i in range(8)]
    assert torch.isfinite(r[0]["params"]).all() and not torch.equal(r[0]["params"], r[0]["p_init"])
    for i in range(1, 8):
        assert torch.equal(r[i]["params"], r[0]["params"]) and r[i]["lr"] == r[0]["lr"] and r[i]["steps"] == 24, i
        assert r[i]["comm_events"] == 24 and r[i]["p2p"] is not None


@pytest.mark.timeout(600)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank; this box has one GPU")
def test_two_ranks_two_gpus_rccl(tmp_path):
    """The same run over RCCL (backend "nccl"), rank r on device r: bit-identical parameters after 32 synchronised steps."""
    port = 29900 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path), "nccl"), nprocs=2, join=True)
    a, b = (torch.load(os.path.join(str(tmp_path), "r%d.pt" % i)) for i in range(2))
    assert torch.equal(a["p_init"], b["p_init"]) and torch.equal(a["params"], b["params"])
    assert torch.isfinite(a["params"]).all() and not torch.equal(a["params"], a["p_init"])
    assert a["lr"] == b["lr"] and a["steps"] == b["steps"] == 32 and a["graph"] and b["graph"]
    assert not torch.equal(a["friction"], b["friction"])


def _single_rank_worker(rank, world, port, out_dir, collectives):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "humanoid-gym_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    if collectives:
        os.environ["HGYM_DIST_SINGLE"] = "1"
        os.environ["HGYM_COMM"] = "rccl"       # this test is about the COLLECTIVE's stream ordering (auto would probe and may pick the direct kernel)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from humanoid.algo import PPO
    PPO.precision = "bf16"
    from humanoid.algo.ppo import dist_utils
    from humanoid.envs import task_registry
    from humanoid.utils import get_args
    assert dist_utils.active() == bool(collectives)
    args = get_args(["--task=humanoid_ppo", "--headless", "--num_envs", "256", "--seed", "5"])
    env, _ = task_registry.make_env(name=args.task, args=args)
    runner, _ = task_registry.make_alg_runner(env=env, name=args.task, args=args, log_root=None)
    runner.alg.comm_timing = []
    runner.learn(num_learning_iterations=4, init_at_random_ep_len=True)
    torch.cuda.synchronize()
    net = runner.alg.net
    torch.save(dict(params=net.params.cpu(), lr=float(net.opt_state[0]), comm_events=len(runner.alg.comm_timing)),
               os.path.join(out_dir, "single%d.pt" % int(collectives)))
    if collectives:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_one_rank_rccl_collective_path_changes_nothing(tmp_path):
    """A box with one GPU cannot run two RCCL ranks, but it can run ONE: HGYM_DIST_SINGLE=1 sends the update through everything
    the N > 1 path does -- parameter broadcast, the advantage-statistics all-reduce, the gradient in two parts with an asynchronous
    all-reduce (backend "nccl" = RCCL, its own stream) behind each, the stream-side waits before hgym_ppo_apply -- on a one-rank
    group, where every collective is the identity.  Parameters after 32 Adam steps must equal the plain run's bit for bit: a
    missing stream dependency between the kernels and the collectives would not."""
    port = 29300 + (os.getpid() % 2000)
    for c in (0, 1):
        mp.spawn(_single_rank_worker, args=(1, port + c, str(tmp_path), c), nprocs=1, join=True)
    a, b = (torch.load(os.path.join(str(tmp_path), "single%d.pt" % c)) for c in (0, 1))
    assert torch.isfinite(a["params"]).all()
    assert torch.equal(a["params"], b["params"]) and a["lr"] == b["lr"]
    assert a["comm_events"] == 0 and b["comm_events"] == 32


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no torch.distributed environment (how the driver starts the N = 1 run) must launch its two
    ranks itself and print ONE JSON line with n_gpus = 2.  On a one-GPU box the ranks share the device over gloo
    (HGYM_DIST_BACKEND); with two devices the default RCCL path runs."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if torch.cuda.device_count() < 2:
        env["HGYM_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "2", "--num-envs", "512"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["envs_per_gpu"] == 512 and "cpu_baseline" not in out and "configs" not in out
    assert out["comm"]["minibatches_per_iter"] == 8 and out["comm"]["exposed_us_per_minibatch"] >= 0
    assert out["comm"]["mode"] == "auto" and out["comm"]["used_in_timed_run"] in ("p2p", "collective")
    assert (out["comm"]["fallback_reason"] is None) == (out["comm"]["used_in_timed_run"] == "p2p"), out["comm"]
    assert out["comm"]["replicas_identical_after_run"] is True and out["comm"]["optimizer_steps"] > 0, out["comm"]
    assert out["roofline"]["frac"] > 0
    # the line ends with the compact block: the last 2 000 bytes (what the driver's record keeps) carry the headline's split
    tail = lines[0][-2000:]
    assert '"summary"' in tail and '"collection_ms"' in tail and '"ppo_update_ms"' in tail and '"rollout_step_kernel"' in tail, tail
