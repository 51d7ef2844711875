"""GPU probe: how the two rollout kernels scale with the number of envs (latency floor vs throughput), plus the
env-step ablation breakdown at N=4096."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "humanoid-gym_amd"))
import torch
from hgym import EnvBuffers, default_env_config, NetBuffers, make_net_config, _lib as L


def timeit(fn, reps=100, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def env_time(N, graph=True):
    cfg = default_env_config(N)
    buf = EnvBuffers(cfg, "cuda")
    sim, st, out = buf.sim_struct(), buf.state_struct(), buf.out_struct()
    nz = buf.noise_struct()
    s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(L.lib.hgym_env_prime(C.byref(cfg), C.byref(sim), C.byref(st), C.byref(out), C.byref(nz), s()))
    a = torch.randn(N, 12, device="cuda")
    f = lambda: L.check(L.lib.hgym_env_step_synth(C.byref(cfg), C.byref(sim), C.byref(st), C.byref(out), L.fptr(a), s()))
    for _ in range(5): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): f()
    return timeit(g.replay, 10, 2) / 20


def policy_time(M):
    cfg = make_net_config(705, 219, 12, [512, 256, 128], [768, 256, 128], "bf16", max(M, 512))
    net = NetBuffers(cfg, "cuda")
    for k, v in net.views.items():
        v.copy_(torch.randn(v.shape, device="cuda") * (0.05 if v.dim() > 1 else 0.01))
    net.views["std"].fill_(1.0)
    net.sync_shadow()
    o, p = torch.randn(M, 705, device="cuda"), torch.randn(M, 219, device="cuda")
    sc = torch.zeros(1, dtype=torch.int64, device="cuda")
    out = net.act(o, p, seed=1, step_counter=sc)
    f = lambda: net.act(o, p, seed=1, step_counter=sc, out=out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): f()
    return timeit(g.replay, 10, 2) / 20


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "env"):
        for N in (256, 1024, 2048, 4096, 8192, 16384):
            print("env_step(+finalize) N=%5d: %.1f us" % (N, env_time(N)), flush=True)
    if what in ("all", "policy"):
        for M in (256, 1024, 2048, 4096, 8192, 16384):
            print("policy_act M=%5d: %.1f us" % (M, policy_time(M)), flush=True)
    if what == "env4096":
        print("env_step(+finalize) N=4096: %.1f us" % env_time(4096), flush=True)
