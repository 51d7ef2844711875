"""XBot-L task configuration (values of reference envs/custom/humanoid_config.py:34-261)."""
from humanoid.envs.base.base_config import namespace as ns
from humanoid.envs.base.legged_robot_config import LeggedRobotCfg as _B, LeggedRobotCfgPPO as _P

_LEG = ("leg_roll", "leg_yaw", "leg_pitch", "knee", "ankle_pitch", "ankle_roll")
_FRAME_STACK, _C_FRAME_STACK, _SINGLE_OBS, _SINGLE_PRIV = 15, 3, 47, 73


class XBotLCfg(_B):
    env = ns("env", _B.env, frame_stack=_FRAME_STACK, c_frame_stack=_C_FRAME_STACK, num_single_obs=_SINGLE_OBS,
             num_observations=_FRAME_STACK * _SINGLE_OBS, single_num_privileged_obs=_SINGLE_PRIV,
             num_privileged_obs=_C_FRAME_STACK * _SINGLE_PRIV, num_actions=12, num_envs=4096, episode_length_s=24,
             use_ref_actions=False)
    safety = ns("safety", pos_limit=1.0, vel_limit=1.0, torque_limit=0.85)
    asset = ns("asset", _B.asset, file="{LEGGED_GYM_ROOT_DIR}/resources/robots/XBot/urdf/XBot-L.urdf", name="XBot-L",
               foot_name="ankle_roll", knee_name="knee", terminate_after_contacts_on=["base_link"],
               penalize_contacts_on=["base_link"], self_collisions=0, flip_visual_attachments=False,
               replace_cylinder_with_capsule=False, fix_base_link=False)
    terrain = ns("terrain", _B.terrain, mesh_type="plane", curriculum=False, measure_heights=False, static_friction=0.6,
                 dynamic_friction=0.6, terrain_length=8.0, terrain_width=8.0, num_rows=20, num_cols=20, max_init_terrain_level=10,
                 terrain_proportions=[0.2, 0.2, 0.4, 0.1, 0.1, 0, 0], restitution=0.0)
    noise = ns("noise", add_noise=True, noise_level=0.6,
               noise_scales=ns("noise_scales", dof_pos=0.05, dof_vel=0.5, ang_vel=0.1, lin_vel=0.05, quat=0.03, height_measurements=0.1))
    init_state = ns("init_state", _B.init_state, pos=[0.0, 0.0, 0.95], default_joint_angles={
        "%s_%s_joint" % (side, j) if j != "knee" else "%s_knee_joint" % side: 0.0
        for side in ("left", "right") for j in _LEG})
    control = ns("control", _B.control, stiffness={"leg_roll": 200.0, "leg_pitch": 350.0, "leg_yaw": 200.0, "knee": 350.0, "ankle": 15},
                 damping={"leg_roll": 10, "leg_pitch": 10, "leg_yaw": 10, "knee": 10, "ankle": 10}, action_scale=0.25, decimation=10)
    sim = ns("sim", _B.sim, dt=0.001, substeps=1, up_axis=1,
             physx=ns("physx", _B.sim.physx, num_threads=10, solver_type=1, num_position_iterations=4, num_velocity_iterations=1,
                      contact_offset=0.01, rest_offset=0.0, bounce_threshold_velocity=0.1, max_depenetration_velocity=1.0,
                      max_gpu_contact_pairs=2 ** 23, default_buffer_size_multiplier=5, contact_collection=2))
    domain_rand = ns("domain_rand", randomize_friction=True, friction_range=[0.1, 2.0], randomize_base_mass=True,
                     added_mass_range=[-5.0, 5.0], push_robots=True, push_interval_s=4, max_push_vel_xy=0.2, max_push_ang_vel=0.4,
                     action_delay=0.5, action_noise=0.02)
    commands = ns("commands", _B.commands, num_commands=4, resampling_time=8.0, heading_command=True,
                  ranges=ns("ranges", lin_vel_x=[-0.3, 0.6], lin_vel_y=[-0.3, 0.3], ang_vel_yaw=[-0.3, 0.3], heading=[-3.14, 3.14]))
    rewards = ns("rewards", base_height_target=0.89, min_dist=0.2, max_dist=0.5, target_joint_pos_scale=0.17, target_feet_height=0.06,
                 cycle_time=0.64, only_positive_rewards=True, tracking_sigma=5, max_contact_force=700,
                 scales=ns("scales", joint_pos=1.6, feet_clearance=1.0, feet_contact_number=1.2, feet_air_time=1.0, foot_slip=-0.05,
                           feet_distance=0.2, knee_distance=0.2, feet_contact_forces=-0.01, tracking_lin_vel=1.2, tracking_ang_vel=1.1,
                           vel_mismatch_exp=0.5, low_speed=0.2, track_vel_hard=0.5, default_joint_pos=0.5, orientation=1.0,
                           base_height=0.2, base_acc=0.2, action_smoothness=-0.002, torques=-1e-5, dof_vel=-5e-4, dof_acc=-1e-7,
                           collision=-1.0))
    normalization = ns("normalization", clip_observations=18.0, clip_actions=18.0,
                       obs_scales=ns("obs_scales", lin_vel=2.0, ang_vel=1.0, dof_pos=1.0, dof_vel=0.05, quat=1.0, height_measurements=5.0))


class XBotLCfgPPO(_P):
    seed = 5
    runner_class_name = "OnPolicyRunner"
    policy = ns("policy", init_noise_std=1.0, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[768, 256, 128])
    algorithm = ns("algorithm", _P.algorithm, entropy_coef=0.001, learning_rate=1e-5, num_learning_epochs=2, gamma=0.994, lam=0.9,
                   num_mini_batches=4)
    runner = ns("runner", policy_class_name="ActorCritic", algorithm_class_name="PPO", num_steps_per_env=60, max_iterations=3001,
                save_interval=100, experiment_name="XBot_ppo", run_name="", resume=False, load_run=-1, checkpoint=-1, resume_path=None)


class XBotLDWLCfgPPO(XBotLCfgPPO):
    """XBot-L PPO with the denoising auxiliary head trained jointly (BASELINE.json configs[4]; the reference only announces it,
    README.md:113 -- hyper-parameters are this repo's: head as wide as the actor, unit loss weight, 73 = one privileged frame)."""
    policy = ns("policy", XBotLCfgPPO.policy, denoiser_hidden_dims=[512, 256, 128], denoiser_targets=_SINGLE_PRIV)
    algorithm = ns("algorithm", XBotLCfgPPO.algorithm, denoise_coef=1.0)
    runner = ns("runner", XBotLCfgPPO.runner, experiment_name="XBot_dwl_ppo")
