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
