"""Data-parallel glue: envs shard across ranks with no data-path collective; the only exchanges are
  * per minibatch: ONE all-reduce of [flat fp32 gradient (926 105) | minibatch mean KL] -> averaged, so every rank
    clips the same gradient and takes the same adaptive-KL learning-rate decision;
  * per iteration: one all-reduce of (sum adv, sum adv^2, count) so advantages are normalised over the global batch.
Backend: torch.distributed "nccl" (= RCCL over xGMI) on the GPUs; the same functions run over "gloo" in the CPU tests."""
import torch
import torch.distributed as dist


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def average_grads_and_kl(grads, opt_state, ext):
    """grads (P,) fp32, opt_state (16,) fp64 with the minibatch KL at [8], ext (P+1,) fp32 scratch."""
    w = world_size()
    if w == 1:
        return
    P = grads.numel()
    ext[:P].copy_(grads)
    ext[P:].copy_(opt_state[8:9])
    dist.all_reduce(ext)
    ext.mul_(1.0 / w)
    grads.copy_(ext[:P])
    opt_state[8:9].copy_(ext[P:])


def allreduce_adv_stats(stats):
    """stats (3,) fp64 = [sum, sum of squares, count] of the raw advantages of this shard."""
    if world_size() > 1:
        dist.all_reduce(stats)
    return stats


def broadcast_parameters(params):
    if world_size() > 1:
        for p in params:
            dist.broadcast(p.data, src=0)
