"""GPU probe: phase clock of the fused rollout step (hgym_rollout_step) -- per-phase mean durations per workgroup of the actor
tile, the critic tile and the env part behind the actor tile (hgym_prof_phase_buffer; 100 MHz stamps)."""
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

N = int(os.environ.get("HGYM_N", "4096"))
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
# HGYM_PROBE_BURST=K: K launches back to back without a host synchronisation in between (as the captured graph runs them: every
# launch finds the caches the way the previous one left them), the phase clock of the LAST one; default: 4 launches, a
# synchronisation behind each
BURST = int(os.environ.get("HGYM_PROBE_BURST", "0"))
NS = BURST if BURST > 0 else 4
env.rollout_begin(alg._sample_step, NS)
acc = []
for i in range(NS):
    if BURST == 0:
        buf.zero_()
    alg.fused_rollout_step(env, i, obs_all[i], priv_all[i], obs_all[i + 1], priv_all[i + 1], (obs_all[i + 2], priv_all[i + 2]))
    if BURST == 0:
        torch.cuda.synchronize()
        acc.append(buf.clone())
torch.cuda.synchronize()
if BURST > 0:
    acc = [None, buf.clone()]
env.rollout_end()
st.step = 0
L.check(L.lib.hgym_prof_phase_buffer(None, 0))
FWD = ["input0+draws", "layer0 k-loop", "epilogue0+sync", "layer1+sync", "layer2+sync", "head"]
ENV = ["hist stores", "joints+sync", "per-env chain+sync", "stage-out", "phase B"]
for i, b in enumerate(acc[1:], NS - 1 if BURST > 0 else 1):
    t = b.view(3, nb, 8).cpu().double() * 0.01
    if os.environ.get("HGYM_RO_INTERLEAVE", "1") != "0":     # the default build: tile b's actor sits in grid row b & 1, its critic in the other
        odd = (torch.arange(nb) & 1).bool()
        a_, c_ = t[0].clone(), t[1].clone()
        a_[odd], c_[odd] = t[1][odd], t[0][odd]
        t = torch.stack([a_, c_, t[2]])
    for row, tag, names in ((0, "actor", FWD), (1, "critic", FWD), (2, "env (behind the actor tile)", ENV)):
        d = t[row]
        segs = [(d[:, k + 1] - d[:, k]).mean().item() for k in range(len(names))]
        print("step %d %-28s block mean %.1f us: " % (i, tag, (d[:, len(names)] - d[:, 0]).mean().item()) +
              ", ".join("%s %.1f" % (n, s) for n, s in zip(names, segs)))
    print("step %d critic workgroup start -> its end (tile + next step's draws + the actor's first-layer partial sums ahead): mean %.1f us, max %.1f us" % (
        i, (t[1][:, 7] - t[1][:, 0]).mean().item(), (t[1][:, 7] - t[1][:, 0]).max().item()))
    print("step %d actor end -> env start (barrier behind the head: the slowest wavefront's head / draws): mean %.1f us" % (i, (t[2][:, 0] - t[0][:, 6]).mean().item()))
    print("step %d actor start -> env end: mean %.1f us, grid span %.1f us" % (
        i, (t[2][:, 5] - t[0][:, 0]).mean().item(), (t[2][:, 5].max() - t[0][:, 0].min()).item()))
    t0 = t[0][:, 0].min()
    r2 = t[2][:, 6] - t0
    print("step %d   third grid row (the finaliser's + 127 empty workgroups) start offsets: min %.1f p50 %.1f max %.1f | finaliser start %.1f end %.1f | last env end %.1f | last critic end %.1f" % (
        i, r2.min().item(), r2.median().item(), r2.max().item(), r2[0].item(), (t[2][0, 7] - t0).item(), (t[2][:, 5] - t0).max().item(), (t[1][:, 7] - t0).max().item()))
    q = lambda x: "min %.1f p50 %.1f p90 %.1f max %.1f" % (x.min().item(), x.median().item(), x.quantile(0.9).item(), x.max().item())
    print("step %d   actor start offsets: %s | critic start offsets: %s" % (i, q(t[0][:, 0] - t0), q(t[1][:, 0] - t0)))
    print("step %d   actor+env durations: %s | env end offsets: %s" % (i, q(t[2][:, 5] - t[0][:, 0]), q(t[2][:, 5] - t0)))
    slow = (t[2][:, 5] - t[0][:, 0]) >= (t[2][:, 5] - t[0][:, 0]).quantile(0.9)
    for row, tag, names in ((0, "actor", FWD), (2, "env", ENV)):
        d = t[row]
        print("step %d   %s phases, slowest 10%% of workgroups vs the rest: " % (i, tag) + ", ".join(
            "%s %.1f/%.1f" % (n, (d[slow, k + 1] - d[slow, k]).mean().item(), (d[~slow, k + 1] - d[~slow, k]).mean().item()) for k, n in enumerate(names)))
    print("step %d   slow workgroups (block ids): %s" % (i, torch.nonzero(slow).flatten().tolist()))
