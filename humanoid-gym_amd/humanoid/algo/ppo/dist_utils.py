"""Data-parallel glue: envs shard across ranks with no data-path collective; the only exchanges are
  * per minibatch: ONE in-place all-reduce (SUM) of [flat fp32 gradient (926 105) | minibatch mean KL] between hgym_ppo_grad and
    hgym_ppo_apply -- the means are formed inside hgym_ppo_apply, so every rank clips the same gradient and takes the same
    adaptive-KL learning-rate decision.  Nothing of the minibatch can run under it (apply needs the result); rounds 1-2 bought an
    overlap by splitting the weight-gradient launch, which cost more than the exchange (ppo.py, update);
  * per iteration: one all-reduce of (sum adv, sum adv^2, count) so advantages are normalised over the global batch.
Backend: torch.distributed "nccl" (= RCCL over xGMI) on the GPUs; the same functions run over "gloo" in the CPU tests."""
import os

import torch
import torch.distributed as dist


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def active():
    """True when the update has to go through the collectives: more than one rank -- or ONE rank with HGYM_DIST_SINGLE=1, which
    issues every collective of the N > 1 path on a one-rank group (results must not change; this is how the RCCL stream
    ordering is exercised on a box with a single GPU, tests/test_dist_gpu.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("HGYM_DIST_SINGLE") == "1"


def sum_grads_and_kl(grads_ext):
    """grads_ext (P+1,) fp32 = [flat gradient | minibatch mean KL] as hgym_ppo_grad leaves it: ONE all-reduce (SUM), in place,
    no staging copies; hgym_ppo_apply divides by world_size on the device (HgymPPOConfig.world_size)."""
    if active():
        dist.all_reduce(grads_ext)


def start_sum(t):
    """Asynchronous in-place all-reduce (SUM) of the gradient vector (or a slice of it); returns a handle for `finish` (None on a single rank).
    With RCCL the collective runs on the process group's own stream, ordered after everything enqueued on the current
    stream so far -- kernels launched after this call overlap with it."""
    if active():
        return dist.all_reduce(t, async_op=True)
    return None


def finish(handle):
    """Order the current stream behind the collective (RCCL: a stream-side wait, the host does not block)."""
    if handle is not None:
        handle.wait()


def allreduce_adv_stats(stats):
    """stats (3,) fp64 = [sum, sum of squares, count] of the raw advantages of this shard."""
    if active():
        dist.all_reduce(stats)
    return stats


def broadcast_parameters(params):
    if active():
        for p in params:
            dist.broadcast(p.data, src=0)
