"""GPU: sha256 of the flat parameter / Adam moment / gradient vectors after N learning iterations from the task's seed -- run under two builds of
the library (HGYM_LIB=...) to show that a kernel change left every bit of the training state alone.
    python tools/param_digest.py [iters] [num_envs]"""
import hashlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "humanoid-gym_amd"))
import torch
from humanoid.envs import task_registry
from humanoid.utils import get_args
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
envs = sys.argv[2] if len(sys.argv) > 2 else "1024"
a = get_args(["--task=humanoid_ppo", "--headless", "--num_envs", envs])
env, _ = task_registry.make_env(name=a.task, args=a)
runner, _ = task_registry.make_alg_runner(env=env, name=a.task, args=a, log_root=None)
runner.learn(num_learning_iterations=iters, init_at_random_ep_len=True)
torch.cuda.synchronize()
net = runner.alg.net
h = lambda t: hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]
print("digest lib=%s iters=%d envs=%s params %s m %s v %s grads %s lr %.6e steps %d" % (
    os.path.basename(os.path.dirname(os.environ.get("HGYM_LIB", "base/x"))), iters, envs, h(net.params), h(net.adam_m), h(net.adam_v), h(net.grads),
    float(net.opt_state[0]), int(net.opt_state[1])))
