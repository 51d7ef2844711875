"""GPU probe: does a HIP graph with k independent [policy_act -> env_step] chains (N/k envs each, one capture stream per chain)
overlap them?  Prints us per vec-step of the whole N-env rollout for k = 1, 2, 4, with and without a per-step cross-chain join."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "humanoid-gym_amd"))
import torch
from hgym import EnvBuffers, default_env_config, NetBuffers, make_net_config, _lib as L

N, STEPS = 4096, 20


def make_net():
    cfg = make_net_config(705, 219, 12, [512, 256, 128], [768, 256, 128], "bf16", N)
    net = NetBuffers(cfg, "cuda")
    for k, v in net.views.items():
        v.copy_(torch.randn(v.shape, device="cuda") * (0.05 if v.dim() > 1 else 0.01))
    net.views["std"].fill_(1.0)
    net.sync_shadow()
    return net


class Chain:
    def __init__(self, n, net):
        self.n, self.net = n, net
        self.cfg = default_env_config(n)
        self.buf = EnvBuffers(self.cfg, "cuda")
        self.sim, self.st, self.out = self.buf.sim_struct(), self.buf.state_struct(), self.buf.out_struct()
        nz = self.buf.noise_struct()
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(L.lib.hgym_env_prime(C.byref(self.cfg), C.byref(self.sim), C.byref(self.st), C.byref(self.out), C.byref(nz), s))
        self.sc = torch.zeros(1, dtype=torch.int64, device="cuda")
        self.o = net.act(self.buf.obs, self.buf.priv_obs, seed=1, step_counter=self.sc)

    def step(self):
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        self.net.act(self.buf.obs, self.buf.priv_obs, seed=1, step_counter=self.sc, out=self.o)
        L.check(L.lib.hgym_env_step_synth(C.byref(self.cfg), C.byref(self.sim), C.byref(self.st), C.byref(self.out),
                                          L.fptr(self.o["actions"]), s))


def run(k, join):
    net = make_net()
    chains = [Chain(N // k, net) for _ in range(k)]
    for c in chains:
        c.step()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(k)]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(main)
        for t in range(STEPS):
            for c, s in zip(chains, streams):
                with torch.cuda.stream(s):
                    c.step()
            if join and k > 1:
                for s in streams[1:]:
                    streams[0].wait_stream(s)
                for s in streams[1:]:
                    s.wait_stream(streams[0])
        for s in streams:
            main.wait_stream(s)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 10 / STEPS


# (the per-step cross-chain join variant aborts inside the HIP graph capture on ROCm 7.2; HGYM_CHAINS_JOIN=1 runs it anyway)
JOINS = (False, True) if os.environ.get("HGYM_CHAINS_JOIN") else (False,)
for k in [int(x) for x in os.environ.get("HGYM_CHAINS", "1,2,4,8").split(",")]:
    for join in JOINS:
        if k == 1 and join:
            continue
        print("chains=%d join=%d: %.1f us per vec-step of %d envs" % (k, join, run(k, join), N), flush=True)
