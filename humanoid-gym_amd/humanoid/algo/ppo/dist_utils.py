"""Data-parallel glue: envs shard across ranks with no data-path collective; the only exchanges are
  * per minibatch: ONE in-place all-reduce of [flat fp32 gradient (926 105) | minibatch mean KL]; the means are formed
    inside hgym_ppo_apply, so every rank clips the same gradient and takes the same adaptive-KL learning-rate decision;
  * per iteration: one all-reduce of (sum adv, sum adv^2, count) so advantages are normalised over the global batch.
Backend: torch.distributed "nccl" (= RCCL over xGMI) on the GPUs; the same functions run over "gloo" in the CPU tests."""
import torch
import torch.distributed as dist


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def sum_grads_and_kl(grads_ext):
    """grads_ext (P+1,) fp32 = [flat gradient | minibatch mean KL] as hgym_ppo_grad leaves it: ONE all-reduce (SUM), in place,
    no staging copies; hgym_ppo_apply divides by world_size on the device (HgymPPOConfig.world_size)."""
    if world_size() > 1:
        dist.all_reduce(grads_ext)


def allreduce_adv_stats(stats):
    """stats (3,) fp64 = [sum, sum of squares, count] of the raw advantages of this shard."""
    if world_size() > 1:
        dist.all_reduce(stats)
    return stats


def broadcast_parameters(params):
    if world_size() > 1:
        for p in params:
            dist.broadcast(p.data, src=0)
