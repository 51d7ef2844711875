"""Data-parallel glue: envs shard across ranks with no data-path collective; the only exchanges are
  * per minibatch: ONE in-place all-reduce (SUM) of [flat fp32 gradient (926 105) | minibatch mean KL] between hgym_ppo_grad and
    hgym_ppo_apply -- the means are formed inside hgym_ppo_apply, so every rank clips the same gradient and takes the same
    adaptive-KL learning-rate decision.  Nothing of the minibatch can run under it (apply needs the result); rounds 1-2 bought an
    overlap by splitting the weight-gradient launch, which cost more than the exchange (ppo.py, update);
  * per iteration: one all-reduce of (sum adv, sum adv^2, count) so advantages are normalised over the global batch.
Backend: torch.distributed "nccl" (= RCCL over xGMI) on the GPUs; the same functions run over "gloo" in the CPU tests.

The per-minibatch exchange has two implementations: the collective above, and ONE direct kernel over peer mappings of the ranks'
gradient buffers (P2PComm below, csrc/hgym_comm.hip: reduce-scatter + all-gather over the fully connected xGMI mesh instead of a
ring).  HGYM_COMM selects:
  auto (default)  2..8 ranks on one host: build the peer mappings, CHECK the direct kernel's sum on a known pattern, TIME both
                  exchanges (PROBE_CALLS calls each, every wait of the probe bounded by PROBE_WAIT_S), agree on the decision over all
                  ranks and use the faster one.  Any failure on any rank -- allocation, IPC export / open, a wrong sum, an expired
                  wait -- makes EVERY rank fall back to the collective; `comm_report()` says what was used and why.
  p2p             the direct kernel, unconditionally (a set-up failure raises)
  rccl            the collective, nothing else is allocated
  both            the collective in training, but the peer-mapped buffer exists (bench.py times the other exchange as well)
HGYM_COMM_FAIL_INJECT=alloc|map|timeout|sum[:rank] makes that step fail on one rank (default: the last) -- how the fallback is tested."""
import os
import socket

import torch
import torch.distributed as dist


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def active():
    """True when the update has to go through the collectives: more than one rank -- or ONE rank with HGYM_DIST_SINGLE=1, which
    issues every collective of the N > 1 path on a one-rank group (results must not change; this is how the RCCL stream
    ordering is exercised on a box with a single GPU, tests/test_dist_gpu.py)."""
    if not (dist.is_available() and dist.is_initialized()) or os.environ.get("HGYM_DIST_OFF") == "1":
        return False        # (HGYM_DIST_OFF=1: a rank of a running group trains ALONE -- bench.py's same-box N = 1 reference line)
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


class ReplicaMismatch(RuntimeError):
    """The data-parallel replicas no longer hold the same parameters (check_replicas) -- or some rank's direct exchange reported an
    expired wait: training must stop on EVERY rank, in the same iteration."""


def check_replicas(params, lr=None, comm_expired=0, what=""):
    """Collective (every rank, same program point): an exact digest of the flat parameter vector -- the int32 view summed in int64, the
    same with position weights, the learning rate's bits -- plus this rank's count of expired waits of the direct exchange, all-gathered
    and compared on every rank, so that all ranks take the same decision.  A stale cache line over xGMI, a time-out that one rank saw
    and another did not, a lost gradient: whatever made the replicas differ raises ReplicaMismatch everywhere instead of letting N
    different policies train on.  Cost: two reductions over 926 105 words, one 4-word all-gather, one host synchronisation -- called every
    save_interval iterations and at the end of learn() (OnPolicyRunner), never inside the timed step.  Returns the digest (list of ints)."""
    if not active():
        return None
    bits = params.detach().view(torch.int32).to(torch.int64)
    w = torch.arange(1, bits.numel() + 1, device=bits.device, dtype=torch.int64)
    lr_bits = int(torch.as_tensor(float(lr) if lr is not None else 0.0, dtype=torch.float64).view(torch.int64))
    mine = torch.stack([bits.sum(), (bits * (w & 0xffff)).sum(), torch.as_tensor(lr_bits, device=bits.device),
                        torch.as_tensor(int(comm_expired), device=bits.device, dtype=torch.int64)])
    got = [torch.empty_like(mine) for _ in range(world_size())]
    dist.all_gather(got, mine)
    rows = [g.tolist() for g in got]
    expired = [q for q, r in enumerate(rows) if r[3] != 0]
    if expired:
        raise ReplicaMismatch("%s: the direct gradient exchange reported expired waits on rank(s) %s: replicas can no longer be trusted" % (what, expired))
    diff = [q for q, r in enumerate(rows) if r[:3] != rows[0][:3]]
    if diff:
        raise ReplicaMismatch("%s: parameter digest of rank(s) %s differs from rank 0's (%s vs %s)" % (what, diff, rows[diff[0]][:3], rows[0][:3]))
    return rows[0][:3]


def broadcast_parameters(params):
    if active():
        for p in params:
            dist.broadcast(p.data, src=0)


PROBE_CALLS = 20
PROBE_ROUNDS = 6          # check rounds of probe_verify (the contribution changes every round)
PROBE_WAIT_S = 2.0
_REPORT = {}


def comm_mode():
    m = os.environ.get("HGYM_COMM", "auto").lower()
    return m if m in ("auto", "p2p", "rccl", "both") else "auto"


def comm_report():
    """What make_comm decided: mode, used ("p2p" / "collective"), fallback_reason (None when the direct exchange is in use or was
    never asked for), probe (per-backend microseconds per call, max over ranks) -- bench.py prints it under "comm"."""
    return dict(_REPORT)


def comm_backend():
    """"p2p": the update's per-minibatch exchange is the direct kernel; "rccl": torch.distributed's all-reduce."""
    return "p2p" if _REPORT.get("used") == "p2p" else "rccl"


def _inject(step):
    spec = os.environ.get("HGYM_COMM_FAIL_INJECT", "")
    if not spec:
        return False
    what, _, r = spec.partition(":")
    rank = int(r) if r else world_size() - 1
    return what == step and dist.get_rank() == rank


def _agree(ok, note=None):
    """Every rank contributes (ok, note); returns (all ok, first failure note).  An object gather: it also carries the reason."""
    got = [None] * world_size()
    dist.all_gather_object(got, (bool(ok), None if note is None else str(note)))
    bad = [(q, n) for q, (o, n) in enumerate(got) if not o]
    return (not bad), (None if not bad else "rank %d: %s" % bad[0])


def make_comm(count, device):
    """The peer-mapped gradient buffer of the data-parallel update when the direct exchange is going to be used (or, HGYM_COMM=both,
    timed), else None -- the caller then keeps its gradient in a plain tensor and exchanges it with the collective.  Every rank takes
    the same decision: each step that can fail locally is followed by an agreement over all ranks BEFORE the next collective call, so
    a failure on some ranks never leaves the others inside a different collective."""
    mode = comm_mode()
    _REPORT.clear()
    _REPORT.update(mode=mode, used="collective", fallback_reason=None, probe=None)
    if not active():
        return None
    W = world_size()
    if mode == "rccl":
        return None

    def give_up(reason, comm=None, torn=True):
        if torn:
            _teardown(comm)         # EVERY rank runs the same barriers, whether or not its own set-up got anywhere
        if mode == "p2p":
            raise RuntimeError("HGYM_COMM=p2p: the direct gradient exchange could not be set up on every rank (%s)" % reason)
        _REPORT["fallback_reason"] = reason
        return None

    if W > HGYM_COMM_MAX_RANKS:
        return give_up("%d ranks (the direct exchange serves at most %d)" % (W, HGYM_COMM_MAX_RANKS), torn=False)
    hosts = [None] * W
    dist.all_gather_object(hosts, socket.gethostname())
    if len(set(hosts)) > 1:
        return give_up("ranks on %d hosts (peer mappings are intra-node)" % len(set(hosts)), torn=False)
    # ---- 1: local allocation + export
    comm, err = None, None
    try:
        if _inject("alloc"):
            raise RuntimeError("injected allocation failure (HGYM_COMM_FAIL_INJECT)")
        comm = P2PComm(count, device)
    except Exception as e:          # noqa: BLE001 -- whatever went wrong, the collective path still works
        err = e
    ok, why = _agree(comm is not None, err)
    if not ok:
        return give_up("allocation / IPC export failed (%s)" % why, comm)
    # ---- 2: exchange the handles, map the peers
    handles = [None] * W
    dist.all_gather_object(handles, comm.handle())
    try:
        if _inject("map"):
            raise RuntimeError("injected mapping failure (HGYM_COMM_FAIL_INJECT)")
        comm.connect(handles)
    except Exception as e:          # noqa: BLE001
        err = e
    ok, why = _agree(comm.connected, err)
    if not ok:
        return give_up("hipIpcOpenMemHandle failed (%s)" % why, comm)
    # ---- 3: first contact: a known pattern through the direct kernel (also for "both": bench.py runs real updates through it) ...
    try:
        ok_local, note = comm.probe_verify(device)
    except Exception as e:          # noqa: BLE001
        ok_local, note = False, "raised %r" % (e,)
    ok, why = _agree(ok_local, note)
    _REPORT["first_contact_verified"] = bool(ok)
    if not ok:
        return give_up("first direct exchange failed (%s)" % why, comm)
    if mode == "both":
        return comm
    # ---- 4: ... then both exchanges timed (the same calls on every rank: nothing in here returns early)
    try:
        probe = comm.probe_time(device)
    except Exception as e:          # noqa: BLE001
        probe = dict(error="raised %r" % (e,))
    got = [None] * W
    dist.all_gather_object(got, probe)
    bad = [(q, g["error"]) for q, g in enumerate(got) if g.get("error")]
    if bad:
        return give_up("timing the direct exchange failed (rank %d: %s)" % bad[0], comm)
    p2p_us = max(g["p2p_us"] for g in got)
    coll_us = max(g["collective_us"] for g in got)
    _REPORT["probe"] = dict(p2p_us_per_call=p2p_us, collective_us_per_call=coll_us, calls=PROBE_CALLS,
                            note="isolated exchanges of the real payload at start-up, max over ranks of each rank's mean (HIP events)")
    if mode == "auto" and not (p2p_us < coll_us):
        return give_up("the collective was faster in the start-up probe (%.1f vs %.1f us per call)" % (coll_us, p2p_us), comm)
    _REPORT["used"] = "p2p"
    return comm


def _teardown(comm):
    """Collective: every rank calls it, with its own P2PComm or None.  Nobody unmaps a buffer a peer's kernel may still touch, nobody
    frees a buffer a peer still has mapped."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    dist.barrier()
    if comm is not None:
        comm._unmap()
    dist.barrier()
    if comm is not None:
        comm._free()


class CommTimeout(RuntimeError):
    """A bounded wait of the direct exchange expired: some rank's kernel never ran beside this one's (P2PComm.check)."""


class _DevMem:
    """A raw device allocation presented to torch through __cuda_array_interface__ (the memory is owned by P2PComm)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = dict(shape=(int(n),), typestr=typestr, data=(int(ptr), False), version=3)


HGYM_COMM_MAX_RANKS = 8


class P2PComm:
    """The ranks' [gradient | KL] vectors in fine-grained device memory, each mapped into every rank (hipIpcMemHandle exchanged once
    through torch.distributed), and hgym_comm_allreduce over them: ONE kernel per minibatch on the compute stream, between
    hgym_ppo_grad and hgym_ppo_apply.  `data` (count floats) IS the rank's gradient vector: NetBuffers is constructed on it.
    Sum in rank order, formed once per element and stored into every rank's buffer: bit-identical on all ranks.
    Construction is local (allocate + export); `connect(handles)` maps the peers; `close()` undoes whatever has been done."""

    def __init__(self, count, device):
        import ctypes as C
        from hgym import _lib as L
        self._L, self._C = L, C
        self.world, self.rank = world_size(), dist.get_rank()
        self.count = (int(count) + 3) // 4 * 4
        self.device = torch.device(device)
        self._flag_off = (self.count * 4 + 255) // 256 * 256
        self._stat_off = self._flag_off + 256
        self._aux_off = self._stat_off + 256            # hgym_comm_sum64's slots: HGYM_COMM_AUX_DOUBLES (2 x 8 x 4) doubles
        self.nbytes = self._aux_off + 512
        # header v9: the call numbers live on the device (status[2] / status[3]) and the advantage statistics travel through the same
        # mappings: an update that uses this communicator issues the same launches with the same arguments every iteration and has no
        # torch.distributed call in it -- PPO.update_capturable
        self.capturable = True
        self._base, self._peer, self.connected, self.seq = None, [None] * self.world, False, 0
        base = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(L.lib.hgym_comm_alloc(self.nbytes, C.byref(base)), "hgym_comm_alloc")
            self._base = base.value
            self._handle = (C.c_ubyte * 64)()
            try:
                L.check(L.lib.hgym_comm_ipc_export(base, self._handle), "hgym_comm_ipc_export")
            except Exception:
                L.lib.hgym_comm_free(base)        # (nobody else will: the constructor's caller never gets the object)
                self._base = None
                raise
        self.data = torch.as_tensor(_DevMem(self._base, self.count, "<f4"), device=self.device)
        self.status = torch.as_tensor(_DevMem(self._base + self._stat_off, 16, "<i8"), device=self.device)

    def handle(self):
        return (bytes(self._handle), os.getpid(), self.device.index if self.device.index is not None else torch.cuda.current_device())

    def connect(self, handles):
        """handles[q] = rank q's handle(): open every peer's buffer and fill the HgymComm.  (The agreement that follows in make_comm is
        the rendezvous: every rank has opened every buffer before anyone's first kernel stores into them.)"""
        L, C = self._L, self._C
        with torch.cuda.device(self.device):
            mine = handles[self.rank][2]
            for q, (h, pid, dev) in enumerate(handles):
                if q == self.rank:
                    self._peer[q] = self._base
                    continue
                if pid == os.getpid():
                    raise RuntimeError("P2PComm: two ranks in one process")
                # (ranks that see all devices of the node and differ by index: ask the runtime before touching peer memory -- a kernel
                # that dereferences an inaccessible mapping is a memory fault, not an error code)
                if dev != mine and max(dev, mine) < torch.cuda.device_count() and not torch.cuda.can_device_access_peer(mine, dev):
                    raise RuntimeError("device %d cannot access device %d (rank %d)" % (mine, dev, q))
                ptr = C.c_void_p()
                hb = (C.c_ubyte * 64).from_buffer_copy(h)
                L.check(L.lib.hgym_comm_ipc_open(hb, C.byref(ptr)), "hgym_comm_ipc_open")
                self._peer[q] = ptr.value
        c = L.Comm()
        c.world, c.rank, c.count = self.world, self.rank, self.count
        for q in range(self.world):
            c.data[q] = C.cast(C.c_void_p(self._peer[q]), L.c_float_p)
            c.flags[q] = C.cast(C.c_void_p(self._peer[q] + self._flag_off), C.POINTER(C.c_uint32))
            c.aux[q] = C.cast(C.c_void_p(self._peer[q] + self._aux_off), L.c_f64_p)
        c.status = C.cast(C.c_void_p(self._base + self._stat_off), L.c_i64_p)
        c.wait_ticks = 0
        self.struct = c
        self.connected = True

    def set_wait(self, seconds):
        """Bound of every wait inside the kernel (None: the library's default, 15 s)."""
        self.struct.wait_ticks = 0 if seconds is None else max(1, int(seconds * 1e8))

    def allreduce(self):
        """In-place SUM over the ranks of `data`, enqueued on the current stream (the same call sequence on every rank).  The call number
        is the device's (hgym_comm_allreduce(seq = 0), header v9: status[2], advanced by the kernel): the launch is the same every time,
        so a captured update replays it; `seq` here only counts the calls this object has enqueued (replays are added by the caller)."""
        self.seq += 1
        L, C = self._L, self._C
        L.check(L.lib.hgym_comm_allreduce(C.byref(self.struct), 0, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                "hgym_comm_allreduce")

    def skip_call(self):
        """This rank does NOT take part in the next exchange (the fault-injection path of the start-up check): the call number moves on
        as if it had, on the host and on the device."""
        self.seq += 1
        self.status[2] += 1

    def sum64(self, t):
        """In-place SUM over the ranks of t (<= 3 doubles on this device), enqueued on the current stream: the advantage statistics
        of the global batch (hgym_comm_sum64: rank-ordered fp64 sum, bit-identical on every rank; call number on the device)."""
        assert t.dtype == torch.float64 and t.is_cuda and t.is_contiguous() and 1 <= t.numel() <= 3
        L, C = self._L, self._C
        L.check(L.lib.hgym_comm_sum64(C.byref(self.struct), L.f64ptr(t), int(t.numel()), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                "hgym_comm_sum64")

    def read_status(self):
        """Synchronises the current stream; the 16 status words as a list of ints."""
        L, C = self._L, self._C
        host = (C.c_int64 * 16)()
        L.check(L.lib.hgym_comm_status(C.byref(self.struct), host, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                "hgym_comm_status")
        return list(host)

    def raise_if_expired(self, words):
        if int(words[0]) != 0:
            raise CommTimeout("hgym_comm_allreduce: a rank did not arrive within the kernel's bounded wait (rank %d, first at call %d of %d); "
                              "the gradient of that minibatch is garbage and the replicas have diverged" % (self.rank, int(words[1]), self.seq))

    def check(self):
        """Synchronises; raises if a bounded wait of any call so far expired (a rank never arrived).  Returns the last call's
        (wait for the slowest rank, exchange) in microseconds -- the kernel's own 100 MHz timestamps."""
        s = self.read_status()
        self.raise_if_expired(s)
        return (s[9] - s[8]) * 0.01, (s[10] - s[9]) * 0.01

    def probe_verify(self, device):
        """First contact, local part (make_comm agrees on the result over all ranks): (ok, note).  PROBE_ROUNDS known patterns through
        the direct kernel back to back, as training issues it: a kernel on this stream writes the contribution, the exchange follows
        with no host synchronisation in between, the result is copied aside, the next round overwrites the buffer.  Rank q contributes
        (q + 1) (k + 1) x [1 .. 7 repeating] in round k, the sum is W (W + 1) / 2 (k + 1) x the pattern, exact in fp32 -- a value that
        changes every round, so a peer's (or this device's) stale cache line of an earlier round shows up as a wrong sum here, not as
        a silently diverged replica later.  Every wait is bounded by PROBE_WAIT_S."""
        # The ONE collective call in here (the barrier that lines the ranks up in front of round 0) is made unconditionally: whatever fails
        # locally before it (an allocation, a launch) is recorded and reported afterwards, so no rank is ever left alone in the barrier
        # while its peer is already in make_comm's agreement.
        W, n = self.world, self.count
        err, pat, got = None, None, None
        try:
            self.set_wait(PROBE_WAIT_S)
            pat = (torch.arange(n, device=device, dtype=torch.float32) % 7.0) + 1.0
            got = torch.zeros(PROBE_ROUNDS, n, device=device, dtype=torch.float32)
            self.data.zero_()
            torch.cuda.synchronize(device)
        except Exception as e:          # noqa: BLE001
            err = e
        dist.barrier()
        if err is not None:
            return False, "set-up of the first exchange raised %r" % (err,)
        skip = _inject("timeout")
        for k in range(PROBE_ROUNDS):
            if skip:
                self.skip_call()              # this rank's kernel never runs: the peers' waits expire
                continue
            torch.mul(pat, float((self.rank + 1) * (k + 1)), out=self.data)
            self.allreduce()
            got[k].copy_(self.data)
            # round 0 is read back alone: a peer that never shows up costs one bounded wait, not PROBE_ROUNDS of them; the rest are
            # enqueued without a host synchronisation in between
            if (k == 0 or k == PROBE_ROUNDS - 1) and self.read_status()[0] != 0:
                return False, "a bounded wait (%.1f s) expired" % PROBE_WAIT_S
        # ... and the small fp64 exchange of the advantage statistics (hgym_comm_sum64), three rounds of changing values, exact in fp64
        if not skip:
            for k in range(3):
                st = torch.tensor([float(self.rank + 1 + k), 0.5 * (self.rank + 1) * (k + 1), 4096.0], dtype=torch.float64, device=device)
                self.sum64(st)
                want64 = [W * (W + 1) / 2.0 + W * k, 0.5 * (k + 1) * W * (W + 1) / 2.0, 4096.0 * W]
                if self.read_status()[0] != 0:
                    return False, "a bounded wait (%.1f s) of the statistics exchange expired" % PROBE_WAIT_S
                if st.tolist() != want64:
                    return False, "wrong sum in the statistics exchange (round %d: %s, expected %s)" % (k + 1, st.tolist(), want64)
        for k in range(PROBE_ROUNDS):
            want = pat * float(W * (W + 1) // 2 * (k + 1))
            if _inject("sum"):
                want = want + 1.0
            if not skip and not torch.equal(got[k], want):
                return False, "wrong sum in %d of %d elements (round %d of %d)" % (int((got[k] != want).sum()), n, k + 1, PROBE_ROUNDS)
        return True, None

    def probe_time(self, device):
        """PROBE_CALLS calls of each exchange on the real payload size between barriers, HIP-event timed: dict(p2p_us, collective_us),
        or dict(error).  The collective sequence -- one agreement, two barriers, 3 + PROBE_CALLS all-reduces -- is the same on every rank
        whatever happens locally: allocations come first and are agreed on (a rank that cannot allocate the scratch tensor makes EVERY
        rank skip the timing), local GPU work is wrapped and its error reported at the end, collective calls are never skipped."""
        err, scratch = None, None
        try:
            scratch = torch.zeros(self.count, device=device, dtype=torch.float32)
            self.data.zero_()
            self.set_wait(0.5)
            torch.cuda.synchronize(device)
        except Exception as e:          # noqa: BLE001
            err = e
        ok, why = _agree(err is None, err)
        if not ok:
            return dict(error="set-up of the timing probe failed (%s)" % why)
        local = []

        def guarded(fn):
            try:
                return fn()
            except Exception as e:      # noqa: BLE001 -- recorded; the collective sequence goes on
                local.append(e)
                return None

        def timed(fn, collective):
            run = fn if collective else (lambda: guarded(fn))       # a collective call is never wrapped away
            for _ in range(3):
                run()
            guarded(lambda: torch.cuda.synchronize(device))
            dist.barrier()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(PROBE_CALLS)]
            for a, b in ev:
                a.record()
                run()
                b.record()
            return guarded(lambda: (torch.cuda.synchronize(device), sum(a.elapsed_time(b) for a, b in ev) / len(ev) * 1e3)[1])

        t_p2p = timed(self.allreduce, False)
        s = guarded(self.read_status)
        t_coll = timed(lambda: finish(start_sum(scratch)), True)
        guarded(lambda: (self.data.zero_(), torch.cuda.synchronize(device), self.set_wait(None)))
        if local:
            return dict(error="timing the exchanges raised %r" % (local[0],))
        if s is None or s[0] != 0:
            return dict(error="a bounded wait expired while timing the direct exchange")
        return dict(p2p_us=t_p2p, collective_us=t_coll)

    def _unmap(self):
        L, C = self._L, self._C
        for q, p in enumerate(self._peer):
            if q != self.rank and p:
                L.lib.hgym_comm_ipc_close(C.c_void_p(p))
        self._peer = [None] * self.world
        self.connected = False

    def _free(self):
        self.data = self.status = None
        if self._base:
            self._L.lib.hgym_comm_free(self._C.c_void_p(self._base))
            self._base = None

    def close(self):
        """Unmap the peers and free the buffer.  Collective: every rank calls it (barriers inside)."""
        _teardown(self)
