"""Where does a logging run's wall time go?  learn(K) with log_dir set, repeated under HGYM_ASYNC_SAVE=0 / 1 and (async) several
interpreter switch intervals; per call: wall, device phases, checkpoint host time on the training thread, wait for the background writer
AFTER the call.  PROBE_LONG=N adds learn(N) with save_interval 50 per mode (what a checkpoint in the middle of a run costs)."""
import contextlib, io, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "humanoid-gym_amd"))
import torch
from humanoid.algo import PPO
PPO.precision = "bf16"
from humanoid.envs import task_registry
from humanoid.utils import get_args

a = get_args(["--task=humanoid_ppo", "--headless", "--num_envs", "4096", "--seed", "5"])
LONG = int(os.environ.get("PROBE_LONG", "0"))
out = []
with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
    env, _ = task_registry.make_env(name=a.task, args=a)
    runner, _ = task_registry.make_alg_runner(env=env, name=a.task, args=a, log_root=tmp)
    runner.learn(num_learning_iterations=2, init_at_random_ep_len=True)
    runner.wait_for_saves()
    torch.cuda.synchronize()
    modes = [("sync", "0", None), ("async", "1", None), ("async sw=0.5ms", "1", 5e-4), ("async sw=0.05ms", "1", 5e-5)]
    for name, flag, sw in modes:
        os.environ["HGYM_ASYNC_SAVE"] = flag
        sys.setswitchinterval(sw if sw else 5e-3)
        for rep in range(4 + (1 if LONG else 0)):
            k = LONG if rep == 4 else 6
            runner.save_interval = 50 if rep == 4 else 100
            s0 = getattr(runner, "save_time_s", 0.0)
            t0 = time.perf_counter()
            runner.learn(num_learning_iterations=k, init_at_random_ep_len=False)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            s1 = getattr(runner, "save_time_s", 0.0)
            runner.wait_for_saves()
            t3 = time.perf_counter()
            out.append("%-18s learn(%d): %.1f ms wall = %.2f ms/iter (+%.1f ms to drain, +%.1f ms for the writer), device %.2f + %.2f ms/iter, "
                       "training thread in save() %.1f ms" % (name, k, (t1 - t0) * 1e3, (t1 - t0) * 1e3 / k, (t2 - t1) * 1e3, (t3 - t2) * 1e3,
                                                                runner.last_collection_time * 1e3, runner.last_learn_time * 1e3, (s1 - s0) * 1e3))
print("\n".join(out))
