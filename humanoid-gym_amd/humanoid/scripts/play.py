"""Roll out a trained policy and export it for sim2sim: the reference's scripts/play.py:48-169 minus the camera / video capture
(graphics; SURVEY.md section 2 out of scope).  Same overrides of the test configuration (:50-66), same policy export (:79-83), the same
twelve logged states of one robot / one joint and the per-episode reward log (:136-158), print_rewards + plot_states at the end."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: E402

from humanoid import LEGGED_GYM_ROOT_DIR  # noqa: E402
from humanoid.envs import *  # noqa: F401,F403,E402
from humanoid.utils import get_args, export_policy_as_jit, task_registry, Logger  # noqa: E402

EXPORT_POLICY = True
FIX_COMMAND = True


def play(args, steps=1200, logger=None):
    env_cfg, train_cfg = task_registry.get_cfgs(name=args.task)
    # override some parameters for testing (play.py:50-66)
    env_cfg.env.num_envs = min(env_cfg.env.num_envs, 1)
    env_cfg.terrain.mesh_type = "plane"
    env_cfg.terrain.num_rows = 5
    env_cfg.terrain.num_cols = 5
    env_cfg.terrain.curriculum = False
    env_cfg.terrain.max_init_terrain_level = 5
    env_cfg.noise.add_noise = True
    env_cfg.domain_rand.push_robots = False
    env_cfg.domain_rand.joint_angle_noise = 0.0
    env_cfg.noise.curriculum = False
    env_cfg.noise.noise_level = 0.5
    train_cfg.seed = 123145
    print("train_cfg.runner_class_name:", train_cfg.runner_class_name)

    env, _ = task_registry.make_env(name=args.task, args=args, env_cfg=env_cfg)
    obs = env.get_observations()
    train_cfg.runner.resume = True
    ppo_runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args, train_cfg=train_cfg)
    policy = ppo_runner.get_inference_policy(device=env.device)
    if EXPORT_POLICY:
        path = os.path.join(LEGGED_GYM_ROOT_DIR, "logs", train_cfg.runner.experiment_name, "exported", "policies")
        export_policy_as_jit(ppo_runner.alg.actor_critic, path)
        print("Exported policy as jit script to: ", path)

    logger = logger if logger is not None else Logger(env.dt)
    robot_index = 0      # which robot is used for logging
    joint_index = 1      # which joint is used for logging
    for _ in range(steps):
        actions = policy(obs.detach())
        if FIX_COMMAND:
            env.commands[:, 0] = 0.5
            env.commands[:, 1] = 0.0
            env.commands[:, 2] = 0.0
            env.commands[:, 3] = 0.0
        obs, critic_obs, rews, dones, infos = env.step(actions.detach())
        logger.log_states({
            "dof_pos_target": actions[robot_index, joint_index].item() * env.cfg.control.action_scale,
            "dof_pos": env.dof_pos[robot_index, joint_index].item(),
            "dof_vel": env.dof_vel[robot_index, joint_index].item(),
            "dof_torque": env.torques[robot_index, joint_index].item(),
            "command_x": env.commands[robot_index, 0].item(),
            "command_y": env.commands[robot_index, 1].item(),
            "command_yaw": env.commands[robot_index, 2].item(),
            "base_vel_x": env.base_lin_vel[robot_index, 0].item(),
            "base_vel_y": env.base_lin_vel[robot_index, 1].item(),
            "base_vel_z": env.base_lin_vel[robot_index, 2].item(),
            "base_vel_yaw": env.base_ang_vel[robot_index, 2].item(),
            "contact_forces_z": env.contact_forces[robot_index, env.feet_indices, 2].cpu().numpy(),
        })
        if infos["episode"]:
            num_episodes = torch.sum(env.reset_buf).item()
            if num_episodes > 0:
                logger.log_rewards(infos["episode"], num_episodes)
    logger.print_rewards()
    logger.plot_states()
    return logger


if __name__ == "__main__":
    play(get_args())
