"""GPU probe: where the per-env chain's time goes when it runs on three wavefronts (library variant built with -DHGYM_W3_PROBE=1, with or
without -DHGYM_ENV_WAVES3=0): end times of the roles' wavefronts relative to the start of the phase, and the SIMD each wavefront of the
workgroup was placed on (slots 4-7: wavefronts 0-3 = the roles main / A / B / frames of the default build, slot 0: wavefront 4, which
fetches the rows after next; with -DHGYM_ENV_WAVES3=0 wavefront 0 is the whole chain and 1-4 fetch).  HGYM_LIB selects the variant."""
import ctypes as C
import os, sys
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
env.rollout_begin(alg._sample_step, 6)
for i in range(6):
    buf.zero_()
    alg.fused_rollout_step(env, i, obs_all[i], priv_all[i], obs_all[i + 1], priv_all[i + 1], (obs_all[i + 2], priv_all[i + 2]))
    torch.cuda.synchronize()
    if i < 3:
        continue
    t = buf.view(3, nb, 8)[2].cpu()
    simd = t[:, 1]
    d = t.double() * 0.01
    rel = lambda k: (d[:, k] - d[:, 2])
    print("step %d chain phase (start -> barrier behind phase F) mean %.2f us | wavefront ends after the phase start: w0 %.2f  w1 %.2f  w2 %.2f  w3 %.2f  w4 %.2f us (max over blocks %.2f %.2f %.2f %.2f %.2f)" % (
        i, rel(3).mean(), rel(4).mean(), rel(5).mean(), rel(6).mean(), rel(7).mean(), rel(0).mean(), rel(4).max(), rel(5).max(), rel(6).max(), rel(7).max(), rel(0).max()))
    from collections import Counter
    c = Counter(tuple((int(v) >> (4 * w)) & 15 for w in range(8)) for v in simd.tolist())
    print("step %d SIMD of wavefronts 0..7, by workgroup count: %s" % (i, c.most_common(4)))
env.rollout_end()
L.check(L.lib.hgym_prof_phase_buffer(None, 0))
