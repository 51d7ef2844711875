"""Data-parallel glue: envs shard across ranks with no data-path collective; the only exchanges are
  * per minibatch: ONE in-place all-reduce (SUM) of [flat fp32 gradient (926 105) | minibatch mean KL] between hgym_ppo_grad and
    hgym_ppo_apply -- the means are formed inside hgym_ppo_apply, so every rank clips the same gradient and takes the same
    adaptive-KL learning-rate decision.  Nothing of the minibatch can run under it (apply needs the result); rounds 1-2 bought an
    overlap by splitting the weight-gradient launch, which cost more than the exchange (ppo.py, update);
  * per iteration: one all-reduce of (sum adv, sum adv^2, count) so advantages are normalised over the global batch.
Backend: torch.distributed "nccl" (= RCCL over xGMI) on the GPUs; the same functions run over "gloo" in the CPU tests.
HGYM_COMM=p2p (opt-in) replaces the per-minibatch all-reduce by ONE direct kernel over peer mappings of the ranks' gradient buffers
(P2PComm below, csrc/hgym_comm.hip): a reduce-scatter + all-gather over the fully connected xGMI mesh instead of a ring."""
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


def comm_backend():
    """"p2p": the direct exchange (HGYM_COMM=p2p, at most 8 ranks); "rccl": torch.distributed's all-reduce (default)."""
    return "p2p" if (active() and os.environ.get("HGYM_COMM", "rccl").lower() == "p2p" and world_size() <= 8) else "rccl"


def make_comm(count, device):
    """The peer-mapped gradient buffer of the data-parallel update, or None.  Built only on request (HGYM_COMM=p2p, or HGYM_COMM=both:
    the exchange stays with the collective but the buffers exist, so that `bench.py --gpus N` can time the direct kernel as well in its
    profiling iterations) -- the default data-parallel path allocates nothing new.  If any rank cannot allocate / export / map (no IPC
    on this driver, ranks on different hosts), EVERY rank falls back to a plain tensor (HGYM_COMM=p2p: raises instead)."""
    if not active() or world_size() > 8 or os.environ.get("HGYM_COMM", "rccl").lower() not in ("p2p", "both"):
        return None
    comm, err = None, None
    try:
        comm = P2PComm(count, device)
    except Exception as e:          # noqa: BLE001 -- whatever went wrong, the collective path still works
        err = e
    ok = torch.tensor([0.0 if comm is None else 1.0], device=device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if float(ok) < 1.0:
        if comm_backend() == "p2p":
            raise RuntimeError("HGYM_COMM=p2p: the peer-mapped gradient buffers could not be set up on every rank (%r)" % (err,))
        return None
    return comm


class CommTimeout(RuntimeError):
    """A bounded wait of the direct exchange expired: some rank's kernel never ran beside this one's (P2PComm.check)."""


class _DevMem:
    """A raw device allocation presented to torch through __cuda_array_interface__ (the memory is owned by P2PComm)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = dict(shape=(int(n),), typestr=typestr, data=(int(ptr), False), version=3)


class P2PComm:
    """The ranks' [gradient | KL] vectors in fine-grained device memory, each mapped into every rank (hipIpcMemHandle exchanged once
    through torch.distributed), and hgym_comm_allreduce over them: ONE kernel per minibatch on the compute stream, between
    hgym_ppo_grad and hgym_ppo_apply.  `data` (count floats) IS the rank's gradient vector: NetBuffers is constructed on it.
    Sum in rank order, formed once per element and stored into every rank's buffer: bit-identical on all ranks."""

    def __init__(self, count, device):
        import ctypes as C
        from hgym import _lib as L
        self._L, self._C = L, C
        self.world, self.rank = world_size(), dist.get_rank()
        self.count = (int(count) + 3) // 4 * 4
        self.device = torch.device(device)
        flag_off = (self.count * 4 + 255) // 256 * 256
        stat_off = flag_off + 256
        self.nbytes = stat_off + 256
        base = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(L.lib.hgym_comm_alloc(self.nbytes, C.byref(base)), "hgym_comm_alloc")
            handle = (C.c_ubyte * 64)()
            L.check(L.lib.hgym_comm_ipc_export(base, handle), "hgym_comm_ipc_export")
            handles = [None] * self.world
            dist.all_gather_object(handles, (bytes(handle), os.getpid()))
            self._base, self._peer = base.value, [None] * self.world
            for q, (h, pid) in enumerate(handles):
                if q == self.rank:
                    self._peer[q] = self._base
                    continue
                if pid == os.getpid():
                    raise RuntimeError("P2PComm: two ranks in one process")
                ptr = C.c_void_p()
                hb = (C.c_ubyte * 64).from_buffer_copy(h)
                L.check(L.lib.hgym_comm_ipc_open(hb, C.byref(ptr)), "hgym_comm_ipc_open")
                self._peer[q] = ptr.value
        self.data = torch.as_tensor(_DevMem(self._base, self.count, "<f4"), device=self.device)
        self.status = torch.as_tensor(_DevMem(self._base + stat_off, 16, "<i8"), device=self.device)
        self._keep = (self.data, self.status)
        c = L.Comm()
        c.world, c.rank, c.count = self.world, self.rank, self.count
        for q in range(self.world):
            c.data[q] = C.cast(C.c_void_p(self._peer[q]), L.c_float_p)
            c.flags[q] = C.cast(C.c_void_p(self._peer[q] + flag_off), C.POINTER(C.c_uint32))
        c.status = C.cast(C.c_void_p(self._base + stat_off), L.c_i64_p)
        self.struct = c
        self.seq = 0
        # (make_comm's all-reduce of the success flag is the rendezvous: every rank has opened every buffer before anyone's first
        # kernel stores into them)

    def allreduce(self):
        """In-place SUM over the ranks of `data`, enqueued on the current stream (the same call sequence on every rank)."""
        self.seq += 1
        L, C = self._L, self._C
        L.check(L.lib.hgym_comm_allreduce(C.byref(self.struct), self.seq & 0xFFFFFFFF or 1,
                                          C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "hgym_comm_allreduce")

    def check(self):
        """Synchronises; raises if a bounded wait of any call so far expired (a rank never arrived).  Returns the last call's
        (wait for the slowest rank, exchange) in microseconds -- the kernel's own 100 MHz timestamps."""
        torch.cuda.synchronize(self.device)
        s = self.status.cpu()
        if int(s[0]) != 0:
            raise CommTimeout("hgym_comm_allreduce: a rank did not arrive within the kernel's bounded wait (rank %d, call %d)" % (self.rank, self.seq))
        return (int(s[9]) - int(s[8])) * 0.01, (int(s[10]) - int(s[9])) * 0.01

    def close(self):
        L, C = self._L, self._C
        torch.cuda.synchronize(self.device)
        dist.barrier()               # nobody unmaps a buffer a peer's kernel may still touch
        for q, p in enumerate(self._peer):
            if q != self.rank and p:
                L.lib.hgym_comm_ipc_close(C.c_void_p(p))
        self._peer = []
