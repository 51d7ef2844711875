"""Where does a logging run's wall time go?  learn(6) with log_dir set, repeated; per call: wall, device phases, checkpoint host time."""
import contextlib, io, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "humanoid-gym_amd"))
import torch
from humanoid.algo import PPO
PPO.precision = "bf16"
from humanoid.envs import task_registry
from humanoid.utils import get_args
from humanoid.algo.ppo import on_policy_runner as R

a = get_args(["--task=humanoid_ppo", "--headless", "--num_envs", "4096", "--seed", "5"])
with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
    env, _ = task_registry.make_env(name=a.task, args=a)
    runner, _ = task_registry.make_alg_runner(env=env, name=a.task, args=a, log_root=tmp)
    out = []
    orig_save, orig_wait = runner.save, runner.wait_for_saves
    marks = {}
    def save(*k, **kw):
        t = time.perf_counter(); orig_save(*k, **kw); marks["save"] = marks.get("save", 0.0) + time.perf_counter() - t
    def wait():
        t = time.perf_counter(); orig_wait(); marks["wait"] = marks.get("wait", 0.0) + time.perf_counter() - t
    runner.save, runner.wait_for_saves = save, wait
    runner.learn(num_learning_iterations=2, init_at_random_ep_len=True)
    torch.cuda.synchronize()
    for rep in range(6):
        marks.clear()
        t0 = time.perf_counter()
        runner.learn(num_learning_iterations=6, init_at_random_ep_len=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out.append("learn(6): %.1f ms wall (+%.1f ms to drain), device %.2f + %.2f ms/iter, save %.1f ms, wait %.1f ms" % (
            (t1 - t0) * 1e3, (t2 - t1) * 1e3, runner.last_collection_time * 1e3, runner.last_learn_time * 1e3,
            marks.get("save", 0) * 1e3, marks.get("wait", 0) * 1e3))
print("\n".join(out))
