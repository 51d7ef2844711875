"""GPU probe: times the fused env step and GAE at the BASELINE size (run on the GPU box)."""
import ctypes as C
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "humanoid-gym_amd"))
import torch
from hgym import EnvBuffers, default_env_config, _lib as L

for N in (4096, 8192, 32768):
    cfg = default_env_config(N)
    buf = EnvBuffers(cfg, "cuda")
    sim, st, out = buf.sim_struct(), buf.state_struct(), buf.out_struct()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    nz = buf.noise_struct()
    L.check(L.lib.hgym_env_prime(C.byref(cfg), C.byref(sim), C.byref(st), C.byref(out), C.byref(nz), s))
    a = torch.randn(N, 12, device="cuda")
    for _ in range(20):
        L.check(L.lib.hgym_env_step_synth(C.byref(cfg), C.byref(sim), C.byref(st), C.byref(out), L.fptr(a), s))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 200
    e0.record()
    for _ in range(K):
        L.check(L.lib.hgym_env_step_synth(C.byref(cfg), C.byref(sim), C.byref(st), C.byref(out), L.fptr(a), s))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / K
    byt = N * 7898
    print("env_step_synth N=%d: %.1f us/step  %.1f GB/s algorithmic  resets/step=%.1f" % (N, us, byt / us / 1e3, float(buf.reset.sum())))
    T = 60
    r, v = torch.rand(T, N, device="cuda"), torch.randn(T, N, device="cuda")
    d = (torch.rand(T, N, device="cuda") < 0.01).to(torch.uint8)
    lv = torch.randn(N, device="cuda")
    ret, adv = torch.zeros(T, N, device="cuda"), torch.zeros(T, N, device="cuda")
    stats = torch.zeros(3, dtype=torch.float64, device="cuda")
    for _ in range(5):
        L.check(L.lib.hgym_gae(T, N, L.fptr(r), L.fptr(v), L.u8ptr(d), L.fptr(lv), 0.994, 0.9, L.fptr(ret), L.fptr(adv), L.f64ptr(stats), s))
    torch.cuda.synchronize()
    e0.record()
    for _ in range(100):
        L.check(L.lib.hgym_gae(T, N, L.fptr(r), L.fptr(v), L.u8ptr(d), L.fptr(lv), 0.994, 0.9, L.fptr(ret), L.fptr(adv), L.f64ptr(stats), s))
        L.check(L.lib.hgym_adv_normalize(T * N, L.fptr(adv), L.f64ptr(stats), s))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 100
    print("gae+norm T=60 N=%d: %.1f us  %.1f GB/s algorithmic" % (N, us, T * N * 25 / us / 1e3))
