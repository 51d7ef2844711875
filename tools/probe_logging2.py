"""bench.py's logging leg step by step (fresh process, HGYM_ASYNC_SAVE as given): wall clock of every phase of the timed learn(6)."""
import contextlib, io, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "humanoid-gym_amd"))
import torch
from humanoid.algo import PPO
PPO.precision = "bf16"
from humanoid.envs import task_registry
from humanoid.utils import get_args
a = get_args(["--task=humanoid_ppo", "--headless", "--num_envs", "4096", "--seed", "5"])
out = []
T0 = time.perf_counter()
def stamp(what):
    out.append("%9.2f ms  %s" % ((time.perf_counter() - T0) * 1e3, what))
with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
    env, _ = task_registry.make_env(name=a.task, args=a)
    runner, _ = task_registry.make_alg_runner(env=env, name=a.task, args=a, log_root=tmp)
    alg = runner.alg
    for name, obj in (("update", alg), ("compute_returns", alg), ("save", runner), ("_log_flush", runner), ("_log_snapshot", runner), ("log", runner)):
        f = getattr(obj, name)
        def wrap(f=f, name=name):
            def g(*k, **kw):
                t = time.perf_counter(); r = f(*k, **kw); dt = (time.perf_counter() - t) * 1e3
                if dt > 0.5: stamp("%s took %.2f ms" % (name, dt))
                return r
            return g
        setattr(obj, name, wrap())
    runner.learn(num_learning_iterations=2, init_at_random_ep_len=True)
    if os.environ.get("PROBE_WAIT", "1") == "1":
        runner.wait_for_saves()
    torch.cuda.synchronize()
    gaps = []
    if os.environ.get("PROBE_GIL"):
        import threading
        def gil_probe():
            last = time.perf_counter()
            while True:
                time.sleep(0.0005)
                t = time.perf_counter()
                if t - last > 0.003:
                    gaps.append((t - T0, t - last))
                last = time.perf_counter()
        threading.Thread(target=gil_probe, daemon=True).start()
    for rep in range(int(os.environ.get("PROBE_REPS", "3"))):
        T0 = time.perf_counter(); out.append("-- learn(6) #%d" % rep)
        runner.learn(num_learning_iterations=6, init_at_random_ep_len=False)
        stamp("learn returned")
        torch.cuda.synchronize()
        stamp("device idle")
        runner.wait_for_saves()
        stamp("writer idle")
        if gaps:
            out.append("   GIL probe thread (sleep 0.5 ms, loop): waits > 3 ms: " + ", ".join("%.1f ms at %.1f" % (g * 1e3, a * 1e3) for a, g in gaps[:40]))
            del gaps[:]
print("\n".join(out))
