// hgym_comm.hip -- the data-parallel update's gradient exchange as ONE direct kernel over peer mappings.  Round 5: HGYM_COMM=auto (the
// default for 2..8 ranks on one host) builds the mappings, checks and times this kernel against torch.distributed's all-reduce (= RCCL)
// once at start-up and uses the faster one; any set-up failure, wrong sum or expired wait falls back to the collective on every rank
// (humanoid/algo/ppo/dist_utils.py).
//
// Why.  The exchange is fully exposed by construction (the norm clip and the KL-adaptive learning rate need
// the whole averaged gradient before Adam can start, and the next minibatch's forward needs Adam's result), so at 6 ms per
// iteration a weak-scaling efficiency of 0.9 leaves <= 88 us per minibatch for collective + rank skew.  The payload is small --
// [flat gradient | KL] = 926 106 floats = 3.7 MB -- and MI355X's xGMI is a fully connected mesh of point-to-point links
// (7 x ~153 GB/s per GPU): a ring collective is bound by per-link latency x 2 (N - 1) hops, a DIRECT reduce-scatter + all-gather
// moves (N - 1) / N of the payload once each way over N - 1 links in parallel: 7 x 463 KB per phase at N = 8, ~3 us of wire time
// per link and phase, two flag round trips.
//
// Protocol (rank r of W, call number seq = 1, 2, ...; every rank's buffer is mapped into every rank's address space through
// hipIpcMemHandles exchanged once, fine-grained memory so that peer writes are visible without cache maintenance):
//   A  the previous kernel on the stream has written my gradient into MY buffer.  Workgroup 0 stores seq into slot A[r] of every
//      rank's flag block; every workgroup waits until its own flag block shows A[q] >= seq for all q: all gradients are complete.
//   RS my shard = elements [r S, (r + 1) S) (S = count / W rounded up to 4).  For each 16-byte piece of it: load the piece from all W
//      buffers, add them in RANK ORDER q = 0 .. W - 1 (fp32, the same association on every rank -- and each element is summed by
//      exactly one rank, so all ranks end with bit-identical vectors), store the sum into all W buffers.
//   B  every workgroup release-fences and bumps a local counter; the last one stores seq into slot B[r] of every rank's flag block.
//      Workgroup 0 waits for B[q] >= seq for all q before it exits: when my kernel completes, every shard of my buffer holds its
//      final sum and no peer still reads my gradient -- the next kernel on the stream (hgym_ppo_apply) may read and overwrite it.
// Visibility across devices (what the protocol relies on; between processes on ONE device all of it is trivially true, which is why
// dist_utils.P2PComm.probe_verify checks several rounds of CHANGING data at first contact and falls back to the collective on a wrong sum):
//   * my gradient, written by the previous kernel, is in memory when phase A's flag goes out: the kernel boundary is at least an
//     agent-scope release, which on a device with one L2 per XCD writes every L2 back (no flag-side fence could: it would reach one XCD);
//   * a peer's loads of it are not served from a line of an earlier call: the system-scope acquire fence behind phase A's wait
//     invalidates the reader's L1 / L2 (the mapping of a peer's fine-grained buffer is uncached anyway);
//   * my sums are in every peer's memory before flag B: every workgroup's system-scope release fence (L2 write-back + vmcnt(0): remote
//     stores are acknowledged by the remote memory), then the agent-scope counter, then the last workgroup's flag stores;
//   * the next kernel on my stream reads what the peers stored into MY buffer: its kernel-start acquire invalidates my caches.
// Every wait is BOUNDED (HgymComm.wait_ticks of the 100 MHz wall clock; 0 = 15 s): on expiry the kernel writes status[0] = 1 (sticky) and
// status[1] = seq, skips the sum and STILL runs phase B's book-keeping -- the local done counter counts one call and is reset by the call's
// last workgroup, so a communicator that has seen a time-out stays usable (the payload of that call is garbage; the caller decides:
// dist_utils falls back to the collective during its start-up probe and raises CommTimeout in training).  The host reads the status
// word where it synchronises anyway (hgym_comm_status).
//
// Timestamps (100 MHz wall clock) of the last call are left in status[8 ..]: kernel start, phase A complete, phase B complete --
// "A - start" is how long this rank waited for the slowest rank (the skew), "B - A" the exchange itself.
#include "hgym_common.hpp"

namespace hgym {

constexpr int COMM_THREADS = 256;
constexpr int COMM_BLOCKS = 64;
constexpr long long COMM_WAIT_TICKS = 1500000000ll;      // bounded wait: 15 s of the 100 MHz wall clock (s_memrealtime)

struct CommArgs {
    float* data[HGYM_COMM_MAX_RANKS];
    uint32_t* flags[HGYM_COMM_MAX_RANKS];        // per rank: [0, 8) phase A arrivals, [8, 16) phase B arrivals, [16] local done counter
    long long* status;                           // MY status block: [0] error flag, [8 ..] timestamps
    int world, rank;
    int64_t count;                               // floats (a multiple of 4)
    int64_t shard;                               // floats per shard (a multiple of 4)
    uint32_t seq;
    long long wait_ticks;                        // bound of every wait, 100 MHz ticks
};

__device__ __forceinline__ uint32_t ld_sys(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// lanes 0 .. world - 1 of the calling wave each watch one slot; returns false on time-out (wave-uniform result)
__device__ __forceinline__ bool wait_slots(const uint32_t* slots, int world, uint32_t seq, int lane, long long ticks) {
    bool ok = true;
    if (lane < world) {
        uint32_t spins = 0;
        const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
        // (seq - flag) as a signed distance: correct across the 32-bit wrap of the call counter
        while ((int32_t)(ld_sys(slots + lane) - seq) < 0) {
            // back off after the first ~thousand polls: a rank that is late by milliseconds (another process's kernels on a shared
            // GPU, a host hiccup) should not be polled at full rate
            if (++spins < 1024u) __builtin_amdgcn_s_sleep(8);
            else __builtin_amdgcn_s_sleep(127);
            if ((spins & 255u) == 0u && (long long)__builtin_amdgcn_s_memrealtime() - t0 > ticks) { ok = false; break; }
        }
    }
    return __all(ok);
}

// seq == 0 in the arguments (header v9): the call number lives on the device -- status[2] holds the number of the last call, every workgroup
// reads it on entry, the call's last workgroup (which all of them have counted on by then) writes it back.  The launch's arguments are then
// the same for every call: what a captured HIP graph replays.
__device__ __forceinline__ uint32_t call_number(uint32_t seq_arg, const long long* last) {
    if (seq_arg != 0u) return seq_arg;
    uint32_t s = (uint32_t)__hip_atomic_load(last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    return s ? s : 1u;      // 1 .. 2^32 - 1, never 0 (a zero-filled flag block means "call 0 has arrived")
}

__global__ __launch_bounds__(COMM_THREADS) void p2p_allreduce_kernel(CommArgs c) {
    const int t = threadIdx.x, lane = t & 63;
    __shared__ int s_ok;
    uint32_t* myf = c.flags[c.rank];
    const bool dev_seq = c.seq == 0u;
    c.seq = call_number(c.seq, c.status + 2);
    if (blockIdx.x == 0 && t == 0) c.status[8] = (long long)__builtin_amdgcn_s_memrealtime();
    // ---- A: my gradient is complete (written by the previous kernel on this stream); tell everyone, wait for everyone
    if (blockIdx.x == 0 && t < c.world) st_sys(c.flags[t] + c.rank, c.seq);
    if (t < 64) {
        const bool ok = wait_slots(myf, c.world, c.seq, lane, c.wait_ticks);
        if (t == 0) s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    const bool arrived = s_ok != 0;            // (workgroup-uniform)
    if (!arrived && t == 0) {
        c.status[0] = 1;
        c.status[1] = (long long)c.seq;
        // ... and remember it for this call's last workgroup, which tells the peers (below): a rank whose wait expired has NOT summed its
        // shard, so nobody's vector is good -- the verdict must be the same on every rank (a late rank used to see both flags
        // of the rank that gave up on it and finish with status 0)
        __hip_atomic_store(myf + 17, c.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (arrived) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        if (blockIdx.x == 0 && t == 0) c.status[9] = (long long)__builtin_amdgcn_s_memrealtime();
        // ---- RS + AG: my shard, summed in rank order, written to every buffer
        const int64_t lo = (int64_t)c.rank * c.shard;
        int64_t hi = lo + c.shard;
        hi = hi < c.count ? hi : c.count;
        typedef __attribute__((ext_vector_type(4))) float f4;
        for (int64_t i = lo + ((int64_t)blockIdx.x * COMM_THREADS + t) * 4; i < hi; i += (int64_t)gridDim.x * COMM_THREADS * 4) {
            f4 v[HGYM_COMM_MAX_RANKS];
#pragma unroll
            for (int q = 0; q < HGYM_COMM_MAX_RANKS; ++q)
                if (q < c.world) v[q] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(c.data[q] + i));
            f4 s = v[0];
#pragma unroll
            for (int q = 1; q < HGYM_COMM_MAX_RANKS; ++q)
                if (q < c.world) s += v[q];
#pragma unroll
            for (int q = 0; q < HGYM_COMM_MAX_RANKS; ++q)
                if (q < c.world) __builtin_nontemporal_store(s, reinterpret_cast<f4*>(c.data[q] + i));
        }
    }
    // ---- B: all my stores are out (release, system scope) -> the last workgroup tells everyone; workgroup 0 waits for everyone.
    // A workgroup whose phase-A wait expired takes part too (it has nothing to release): the done counter counts THIS call's workgroups
    // and the last one resets it, so the next call elects its last workgroup whatever happened in this one.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    __shared__ int s_last;
    if (t == 0) {
        const uint32_t done = __hip_atomic_fetch_add(myf + 16, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (done == gridDim.x - 1u) ? 1 : 0;
        if (s_last) {
            __hip_atomic_store(myf + 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every workgroup of this call has counted
            if (dev_seq) __hip_atomic_store(c.status + 2, (long long)c.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ... and has read the call number
        }
    }
    __syncthreads();
    if (s_last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "");
        // slot 24 + rank of every flag block: "rank gave up in call seq" -- stored (and released) in front of the completion flag, so whoever
        // sees this rank's completion flag of the call also sees its verdict
        const bool gave_up = __hip_atomic_load(myf + 17, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == c.seq;
        if (gave_up && t < c.world) st_sys(c.flags[t] + 24 + c.rank, c.seq);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        if (t < c.world) st_sys(c.flags[t] + 8 + c.rank, c.seq);
    }
    if (blockIdx.x == 0) {
        if (t < 64) {
            // (after an expired phase-A wait the peers' completion flags cannot be expected either: no second bounded wait)
            bool ok = arrived ? wait_slots(myf + 8, c.world, c.seq, lane, c.wait_ticks) : false;
            if (ok) {       // every rank completed the call: did any of them give up on somebody?
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
                const bool bad = lane < c.world && ld_sys(myf + 24 + lane) == c.seq;
                ok = !__any(bad);
            }
            if (t == 0) {
                if (!ok) {
                    c.status[0] = 1;
                    c.status[1] = (long long)c.seq;
                }
                c.status[10] = (long long)__builtin_amdgcn_s_memrealtime();
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
}

// hgym_comm_sum64 (header v9): the per-iteration all-reduce of the advantage statistics -- n <= 3 doubles -- over the same peer mappings, so that
// a data-parallel update has NO torch.distributed call left in it and can be captured.  One wavefront.  Rank r stores {v0, v1, v2} and then the
// call number as a tag into slot [parity][r] of EVERY rank's aux block; lane q waits for the tag of slot [parity][q] in its own block, then the
// wave adds the slots in rank order q = 0 .. W - 1 (fp64, the same association on every rank: bit-identical results).  Two slot sets, used
// alternately: a rank can be one call ahead of a peer that has not read the previous call's slots yet, never two (it needs that peer's
// arrival of the call in between).  The call number is status[3]; waits are bounded like the gradient exchange's (status[0] = 1 on expiry).
struct Sum64Args {
    double* aux[HGYM_COMM_MAX_RANKS];
    long long* status;
    double* vals;
    int world, rank, n;
    long long wait_ticks;
};

__global__ __launch_bounds__(64) void p2p_sum64_kernel(const Sum64Args c) {
    const int lane = threadIdx.x;
    const uint32_t seq = call_number(0u, c.status + 3);
    const int par = (int)(seq & 1u);
    __shared__ double s_v[HGYM_COMM_MAX_RANKS][4];
    double mine[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (i < c.n) mine[i] = c.vals[i];
    if (lane < c.world) {
        double* slot = c.aux[lane] + ((int64_t)par * HGYM_COMM_MAX_RANKS + c.rank) * 4;
#pragma unroll
        for (int i = 0; i < 3; ++i) __hip_atomic_store(slot + 1 + i, mine[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (lane < c.world) {
        unsigned long long* tag = reinterpret_cast<unsigned long long*>(c.aux[lane] + ((int64_t)par * HGYM_COMM_MAX_RANKS + c.rank) * 4);
        __hip_atomic_store(tag, (unsigned long long)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    bool ok = true;
    if (lane < c.world) {
        const unsigned long long* tag = reinterpret_cast<const unsigned long long*>(c.aux[c.rank] + ((int64_t)par * HGYM_COMM_MAX_RANKS + lane) * 4);
        uint32_t spins = 0;
        const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
        while ((uint32_t)__hip_atomic_load(tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if (++spins < 1024u) __builtin_amdgcn_s_sleep(8);
            else __builtin_amdgcn_s_sleep(127);
            if ((spins & 255u) == 0u && (long long)__builtin_amdgcn_s_memrealtime() - t0 > c.wait_ticks) { ok = false; break; }
        }
    }
    ok = __all(ok);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    if (ok && lane < c.world) {
        const double* slot = c.aux[c.rank] + ((int64_t)par * HGYM_COMM_MAX_RANKS + lane) * 4;
#pragma unroll
        for (int i = 0; i < 3; ++i) s_v[lane][i] = __hip_atomic_load(slot + 1 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (lane == 0) {
        if (ok) {
            for (int i = 0; i < c.n; ++i) {
                double t = s_v[0][i];
                for (int q = 1; q < c.world; ++q) t += s_v[q][i];
                c.vals[i] = t;
            }
        } else {
            c.status[0] = 1;
            c.status[1] = (long long)seq;
        }
        __hip_atomic_store(c.status + 3, (long long)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace hgym

using namespace hgym;

extern "C" {

int32_t hgym_comm_alloc(int64_t bytes, void** dev_ptr) {
    HG_REQUIRE(bytes > 0 && dev_ptr, HGYM_E_BADARG, "bytes=%lld", (long long)bytes);
    void* p = nullptr;
    // fine-grained: device-coherent, not cached in the writer's L2 -- what a peer stores is what the next local kernel loads
    if (hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        HG_FAIL(HGYM_E_LAUNCH, "hipExtMallocWithFlags(%lld bytes, fine-grained) failed", (long long)bytes);
    }
    if (hipMemset(p, 0, (size_t)bytes) != hipSuccess) HG_FAIL(HGYM_E_LAUNCH, "hipMemset failed");
    if (hipDeviceSynchronize() != hipSuccess) HG_FAIL(HGYM_E_LAUNCH, "hipDeviceSynchronize failed");
    *dev_ptr = p;
    return HGYM_OK;
}

int32_t hgym_comm_free(void* dev_ptr) {
    if (dev_ptr && hipFree(dev_ptr) != hipSuccess) HG_FAIL(HGYM_E_LAUNCH, "hipFree failed");
    return HGYM_OK;
}

int32_t hgym_comm_ipc_export(void* dev_ptr, void* handle_out) {
    HG_REQUIRE(dev_ptr && handle_out, HGYM_E_BADARG, "null pointer");
    static_assert(sizeof(hipIpcMemHandle_t) <= HGYM_IPC_HANDLE_BYTES, "handle does not fit");
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, dev_ptr) != hipSuccess) {
        (void)hipGetLastError();
        HG_FAIL(HGYM_E_LAUNCH, "hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 is required on this driver)");
    }
    memset(handle_out, 0, HGYM_IPC_HANDLE_BYTES);
    memcpy(handle_out, &h, sizeof(h));
    return HGYM_OK;
}

int32_t hgym_comm_ipc_open(const void* handle, void** dev_ptr) {
    HG_REQUIRE(handle && dev_ptr, HGYM_E_BADARG, "null pointer");
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        (void)hipGetLastError();
        HG_FAIL(HGYM_E_LAUNCH, "hipIpcOpenMemHandle failed");
    }
    *dev_ptr = p;
    return HGYM_OK;
}

int32_t hgym_comm_ipc_close(void* dev_ptr) {
    if (dev_ptr && hipIpcCloseMemHandle(dev_ptr) != hipSuccess) {
        (void)hipGetLastError();
        HG_FAIL(HGYM_E_LAUNCH, "hipIpcCloseMemHandle failed");
    }
    return HGYM_OK;
}

int32_t hgym_comm_allreduce(const HgymComm* c, uint32_t seq, void* stream) {
    HG_REQUIRE(c && c->world >= 1 && c->world <= HGYM_COMM_MAX_RANKS && c->rank >= 0 && c->rank < c->world, HGYM_E_BADARG, "bad communicator");
    HG_REQUIRE(c->count > 0 && c->count % 4 == 0 && c->status, HGYM_E_BADARG, "count=%lld (a positive multiple of 4)", (long long)c->count);
    CommArgs a;
    memset(&a, 0, sizeof(a));
    for (int q = 0; q < c->world; ++q) {
        HG_REQUIRE(c->data[q] && c->flags[q] && ((uintptr_t)c->data[q] & 15) == 0, HGYM_E_BADARG, "rank %d: null / unaligned buffer", q);
        a.data[q] = c->data[q];
        a.flags[q] = c->flags[q];
    }
    a.status = (long long*)c->status;
    a.world = c->world;
    a.rank = c->rank;
    a.count = c->count;
    a.shard = round_up(ceil_div(c->count, c->world), 4);
    a.seq = seq;
    a.wait_ticks = c->wait_ticks > 0 ? (long long)c->wait_ticks : COMM_WAIT_TICKS;
    prof_begin(HGYM_PROF_COMM, (hipStream_t)stream);
    hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(COMM_BLOCKS), dim3(COMM_THREADS), 0, (hipStream_t)stream, a);
    prof_end(HGYM_PROF_COMM, (hipStream_t)stream, (double)c->count * 4.0 * 2.0 * (c->world - 1) / c->world);
    HG_CHECK_LAUNCH("p2p_allreduce_kernel");
    return HGYM_OK;
}

int32_t hgym_comm_sum64(const HgymComm* c, double* vals, int32_t n, void* stream) {
    HG_REQUIRE(c && c->world >= 1 && c->world <= HGYM_COMM_MAX_RANKS && c->rank >= 0 && c->rank < c->world, HGYM_E_BADARG, "bad communicator");
    HG_REQUIRE(vals && n >= 1 && n <= 3 && c->status, HGYM_E_BADARG, "n=%d (1 .. 3)", n);
    Sum64Args a;
    memset(&a, 0, sizeof(a));
    for (int q = 0; q < c->world; ++q) {
        HG_REQUIRE(c->aux[q] && ((uintptr_t)c->aux[q] & 7) == 0, HGYM_E_BADARG, "rank %d: null / unaligned aux block", q);
        a.aux[q] = c->aux[q];
    }
    a.status = (long long*)c->status;
    a.vals = vals;
    a.world = c->world;
    a.rank = c->rank;
    a.n = n;
    a.wait_ticks = c->wait_ticks > 0 ? (long long)c->wait_ticks : COMM_WAIT_TICKS;
    hipLaunchKernelGGL(p2p_sum64_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    HG_CHECK_LAUNCH("p2p_sum64_kernel");
    return HGYM_OK;
}

int32_t hgym_comm_status(const HgymComm* c, int64_t* host16, void* stream) {
    HG_REQUIRE(c && c->status && host16, HGYM_E_BADARG, "null pointer");
    // waits for `stream` (the caller synchronises here anyway: the end of an update, a log read-back) and hands back the 16 status words:
    // [0] != 0: a bounded wait of some call expired (sticky), [1] the call number it was, [8 .. 10] the last call's timestamps
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
        (void)hipGetLastError();
        HG_FAIL(HGYM_E_LAUNCH, "hipStreamSynchronize failed");
    }
    if (hipMemcpy(host16, c->status, 16 * sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        HG_FAIL(HGYM_E_LAUNCH, "hipMemcpy of the status block failed");
    }
    return HGYM_OK;
}

}  // extern "C"
