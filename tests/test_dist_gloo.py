"""CPU, world_size 2 over gloo: the data-parallel exchanges of the N>1 path -- gradient + KL
averaging in one collective, global advantage normalisation statistics, parameter broadcast -- reproduce the
single-process computation over the concatenated shards (checked with the oracle)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "humanoid-gym_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from humanoid.algo.ppo import dist_utils as D
    g = torch.Generator().manual_seed(100 + rank)
    P = 1000
    grads = torch.randn(P, generator=g)
    # [gradient | KL] exactly as hgym_ppo_grad leaves net.grads (P + 1 floats); hgym_ppo_apply then multiplies by
    # 1 / world_size on the device (fp32 product) -- done here on the host: this file has no GPU.  The device-side mean is checked by
    # tests/test_net_gpu.py::test_apply_with_world_size_forms_the_rank_mean_itself, the whole loop by tests/test_dist_gpu.py
    ext = torch.cat([grads, torch.tensor([0.01 * (rank + 1)])])
    g_in = grads.clone()
    D.sum_grads_and_kl(ext)
    inv_w = torch.tensor(1.0 / world, dtype=torch.float32)
    grads = ext[:P] * inv_w
    opt = torch.zeros(16, dtype=torch.float64)
    opt[8] = float(ext[P] * inv_w)
    # advantage statistics of this rank's shard
    adv = torch.randn(60, 7, generator=g) * (1 + rank) + rank
    stats = torch.tensor([adv.double().sum(), (adv.double() ** 2).sum(), adv.numel()], dtype=torch.float64)
    D.allreduce_adv_stats(stats)
    w = torch.full((5,), float(rank))
    D.broadcast_parameters([torch.nn.Parameter(w)])
    torch.save(dict(g_in=g_in, g_out=grads, kl=opt[8].clone(), adv=adv, stats=stats, w=w), os.path.join(out_dir, "r%d.pt" % rank))
    dist.destroy_process_group()


def _comm_worker(rank, world, port, out_dir, mode):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "humanoid-gym_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HGYM_COMM"] = mode
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from humanoid.algo.ppo import dist_utils as D
    err = None
    try:
        comm = D.make_comm(1001, "cuda:0")
    except RuntimeError as e:
        comm, err = None, str(e)
    # whatever make_comm did, every rank is in step again: a collective issued now must complete
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    torch.save(dict(none=comm is None, report=D.comm_report(), err=err, total=float(t)), os.path.join(out_dir, "c%d.pt" % rank))
    dist.destroy_process_group()


def test_make_comm_falls_back_on_every_rank_when_no_rank_can_allocate(tmp_path):
    """HGYM_COMM=auto on a host WITHOUT a GPU: the direct exchange's allocation fails on every rank (hgym_comm_alloc needs a device), so
    make_comm must come back with None on every rank, name the reason, and leave the ranks in step (the next collective completes) --
    ADVICE r04: a set-up failure must never leave some ranks inside a different collective.  HGYM_COMM=p2p raises instead; rccl does
    not try."""
    import pytest
    if torch.cuda.is_available():
        pytest.skip("this is the no-device case; tests/test_dist_gpu.py injects the failures on a GPU")
    port = 29650 + (os.getpid() % 2000)
    for k, mode in enumerate(("auto", "p2p", "rccl", "both")):
        d = tmp_path / mode
        os.makedirs(str(d))
        mp.spawn(_comm_worker, args=(2, port + k, str(d), mode), nprocs=2, join=True)
        r = [torch.load(os.path.join(str(d), "c%d.pt" % i)) for i in range(2)]
        for x in r:
            assert x["none"] and x["total"] == 3.0 and x["report"]["mode"] == mode and x["report"]["used"] == "collective", (mode, x)
            if mode == "rccl":
                assert x["report"]["fallback_reason"] is None and x["err"] is None
            elif mode == "p2p":
                assert x["err"] and "could not be set up" in x["err"]
            else:
                assert "allocation" in x["report"]["fallback_reason"] and x["err"] is None, x


def test_two_rank_exchanges(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % i)) for i in range(2)]
    mean_g = (r[0]["g_in"] + r[1]["g_in"]) / 2
    for i in range(2):
        np.testing.assert_allclose(r[i]["g_out"].numpy(), mean_g.numpy(), rtol=1e-6, atol=1e-7)   # same averaged gradient
        assert abs(float(r[i]["kl"]) - 0.015) < 1e-8                                              # same averaged KL -> same LR step
        assert torch.equal(r[i]["w"], torch.zeros(5))                                            # rank 0's parameters everywhere
    assert torch.equal(r[0]["g_out"], r[1]["g_out"]) and torch.equal(r[0]["stats"], r[1]["stats"])
    # normalising each shard with the all-reduced statistics == normalising the concatenated batch (oracle)
    from oracle import ppo_oracle as P
    both = torch.cat([r[0]["adv"], r[1]["adv"]], dim=1)
    want = P.normalize_advantages(both)
    s = r[0]["stats"]
    n = s[2]
    mean = s[0] / n
    std = torch.sqrt((s[1] - s[0] * s[0] / n) / (n - 1))
    got = torch.cat([(r[i]["adv"] - mean.float()) / (std.float() + 1e-8) for i in range(2)], dim=1)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-5, atol=1e-6)


def _replica_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "humanoid-gym_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from humanoid.algo.ppo import dist_utils as D
    g = torch.Generator().manual_seed(5)
    params = torch.randn(926105, generator=g)
    res = {}
    res["same"] = D.check_replicas(params, lr=1e-3, what="identical replicas")
    for name, mutate, kw in (("one_ulp", lambda p: p.view(torch.int32).__setitem__(777, p.view(torch.int32)[777] + (1 if rank == 1 else 0)), {}),
                             ("swap", lambda p: None, {}), ("lr", lambda p: None, dict(lr=1e-3 if rank == 0 else 1.0000001e-3)),
                             ("expired", lambda p: None, dict(comm_expired=1 if rank == 1 else 0))):
        p = params.clone()
        if name == "swap" and rank == 1:                    # two elements exchanged: same plain sum, caught by the weighted one
            p[10], p[11] = params[11].clone(), params[10].clone()
        mutate(p)
        try:
            D.check_replicas(p, **dict(dict(lr=1e-3), **kw), what=name)
            res[name] = None
        except D.ReplicaMismatch as e:
            res[name] = str(e)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)                                      # every rank left every check in step
    res["total"] = float(t)
    torch.save(res, os.path.join(out_dir, "p%d.pt" % rank))
    dist.destroy_process_group()


def test_replica_digest_check_raises_on_every_rank(tmp_path):
    """dist_utils.check_replicas (OnPolicyRunner runs it every save_interval iterations and at the end of learn()): identical
    replicas pass with the same digest on both ranks; ONE parameter that differs in its last bit, two exchanged elements, a learning rate
    that differs in its last digit, or an expired wait of the direct exchange reported by one rank raise ReplicaMismatch on BOTH ranks --
    and the ranks are in step afterwards."""
    port = 29800 + (os.getpid() % 2000)
    mp.spawn(_replica_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(os.path.join(str(tmp_path), "p%d.pt" % i)) for i in range(2)]
    assert r[0]["same"] == r[1]["same"] and r[0]["same"] is not None
    for name in ("one_ulp", "swap", "lr", "expired"):
        assert r[0][name] and r[1][name], (name, r[0][name], r[1][name])
    assert "expired" in r[0]["expired"] and "differs" in r[0]["one_ulp"]
    assert r[0]["total"] == r[1]["total"] == 3.0
