"""Roll out a trained policy in a few envs and export it for sim2sim (reference scripts/play.py:48-169 minus the
camera / video capture, which is graphics)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: E402

from humanoid import LEGGED_GYM_ROOT_DIR  # noqa: E402
from humanoid.envs import *  # noqa: F401,F403,E402
from humanoid.utils import get_args, export_policy_as_jit, task_registry, Logger  # noqa: E402

EXPORT_POLICY = True


def play(args, steps=1200):
    env_cfg, train_cfg = task_registry.get_cfgs(name=args.task)
    env_cfg.env.num_envs = min(env_cfg.env.num_envs, 4)
    env_cfg.noise.add_noise = True
    env_cfg.domain_rand.push_robots = False
    env, _ = task_registry.make_env(name=args.task, args=args, env_cfg=env_cfg)
    obs = env.get_observations()
    train_cfg.runner.resume = True
    ppo_runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args, train_cfg=train_cfg)
    policy = ppo_runner.get_inference_policy(device=env.device)
    if EXPORT_POLICY:
        path = os.path.join(LEGGED_GYM_ROOT_DIR, "logs", train_cfg.runner.experiment_name, "exported", "policies")
        export_policy_as_jit(ppo_runner.alg.actor_critic, path)
        print("Exported policy as jit script to: ", path)
    logger = Logger(env.dt)
    for i in range(steps):
        actions = policy(obs.detach())
        env.commands[:, 0] = 0.5
        env.commands[:, 1] = 0.0
        env.commands[:, 2] = 0.0
        env.commands[:, 3] = 0.0
        obs, critic_obs, rews, dones, infos = env.step(actions.detach())
        logger.log_states({"command_x": env.commands[0, 0].item(), "base_vel_x": env.base_lin_vel[0, 0].item(),
                           "dof_pos": env.dof_pos[0, 0].item(), "dof_torque": env.torques[0, 0].item()})
    logger.print_rewards()


if __name__ == "__main__":
    play(get_args())
