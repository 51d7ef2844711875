import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "humanoid-gym_amd"))
import numpy as np, torch
from humanoid.envs import task_registry
from humanoid.utils import get_args
from humanoid.algo import PPO
PPO.precision = "bf16"
for k in range(3):
    torch.manual_seed(1234); np.random.seed(1234)
    args = get_args(["--task=humanoid_ppo", "--headless", "--num_envs", "256", "--seed", "77"])
    env, _ = task_registry.make_env(name=args.task, args=args)
    print("after make_env torch rng", torch.rand(1).item())
    r, _ = task_registry.make_alg_runner(env=env, name=args.task, args=args, log_root=None)
    torch.cuda.synchronize()
    print(k, "params sum %.6f" % float(r.alg.net.params.double().sum()), "friction %.6f" % float(env._buf.f["friction"].double().sum()),
          "obs0 %.6f" % float(r.alg.storage._obs_all[0].double().sum()), "eplen", int(env.episode_length_buf.sum()))
