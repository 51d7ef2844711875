"""GPU probe: WHERE the slow workgroups of the fused rollout launch run.  Library variant built with -DHGYM_HWID_PROBE=1 (HGYM_LIB): every actor +
env workgroup records XCC_ID and HW_ID (shader engine, shader array, CU, SIMD of wavefront 0) beside its phase clock.  K launches back to back
(HGYM_PROBE_BURST, default 20), the last one analysed: duration of the actor + env workgroup by XCD, by shader engine, by CU parity (CUs 2k / 2k + 1
share an instruction cache and a scalar cache), and whether the workgroup on the neighbouring CU is an actor or a critic one is NOT known here (the
critic row has no free slot) -- the by-CU table shows which CU ids were taken by actor workgroups."""
import ctypes as C
import os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "humanoid-gym_amd"))
import torch
from hgym import _lib as L
from humanoid.algo import PPO
PPO.precision = "bf16"
from humanoid.envs import task_registry
from humanoid.utils import get_args

N = 4096
os.environ["HGYM_GRAPH"] = "0"
a = get_args(["--task=humanoid_ppo", "--headless", "--num_envs", str(N)])
env, _ = task_registry.make_env(name=a.task, args=a)
runner, _ = task_registry.make_alg_runner(env=env, name=a.task, args=a, log_root=None)
runner.learn(num_learning_iterations=1, init_at_random_ep_len=True)
torch.cuda.synchronize()
nb = N // 32
buf = torch.zeros(nb * 3 * 8, dtype=torch.int64, device="cuda")
L.check(L.lib.hgym_prof_phase_buffer(C.c_void_p(buf.data_ptr()), buf.numel()))
alg, st = runner.alg, runner.alg.storage
obs_all, priv_all = st._obs_all, st._priv_all
alg.env_stores_transitions = True
K = int(os.environ.get("HGYM_PROBE_BURST", "20"))
for rep in range(3):
    env.rollout_begin(alg._sample_step, K)
    for i in range(K):
        alg.fused_rollout_step(env, i, obs_all[i], priv_all[i], obs_all[i + 1], priv_all[i + 1], (obs_all[i + 2], priv_all[i + 2]))
    torch.cuda.synchronize()
    env.rollout_end()
    st.step = 0
    t = buf.view(3, nb, 8).cpu()
    odd = (torch.arange(nb) & 1).bool()
    a0 = t[0][:, 0].clone()
    a0[odd] = t[1][odd, 0]                         # interleaved build: tile b's actor sits in grid row b & 1
    dur = (t[2][:, 5] - a0).double() * 0.01        # actor start -> env end, us
    hw = t[2][:, 1]
    xcc, hwid = (hw >> 32) & 15, hw & 0xFFFFFFFF
    cu, sh, se, simd = (hwid >> 8) & 15, (hwid >> 12) & 1, (hwid >> 13) & 7, (hwid >> 4) & 3
    print("burst %d: actor + env workgroups: mean %.1f us, p90 %.1f, max %.1f" % (rep, dur.mean(), dur.quantile(0.9), dur.max()))
    for name, key in (("XCD", xcc), ("shader engine", se), ("shader array", sh), ("CU id", cu), ("CU parity", cu & 1)):
        g = defaultdict(list)
        for k, d in zip(key.tolist(), dur.tolist()):
            g[k].append(d)
        print("  by %-14s" % name, "  ".join("%d: n=%d mean %.1f max %.1f" % (k, len(v), sum(v) / len(v), max(v)) for k, v in sorted(g.items())))
    slow = dur >= dur.quantile(0.9)
    print("  slowest 10 %% (block: xcc/se/sh/cu -> us):", ", ".join("%d: %d/%d/%d/%d -> %.1f" % (b, xcc[b], se[b], sh[b], cu[b], dur[b]) for b in torch.nonzero(slow).flatten().tolist()))
L.check(L.lib.hgym_prof_phase_buffer(None, 0))
