"""python humanoid-gym_amd/humanoid/scripts/train.py --task=humanoid_ppo --headless  (reference scripts/train.py:36-43)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from humanoid.envs import *  # noqa: F401,F403,E402
from humanoid.utils import get_args, task_registry  # noqa: E402


def train(args):
    env, env_cfg = task_registry.make_env(name=args.task, args=args)
    ppo_runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args)
    ppo_runner.learn(num_learning_iterations=train_cfg.runner.max_iterations, init_at_random_ep_len=True)


if __name__ == "__main__":
    train(get_args())
