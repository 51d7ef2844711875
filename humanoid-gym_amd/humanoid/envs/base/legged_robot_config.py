"""Default sections of a legged-robot task (values of reference envs/base/legged_robot_config.py:34-237).

Only the constants are shared with the reference; sections the MI355X hot path does not consume (PhysX solver
settings, viewer, terrain generator parameters) are still present so that user configs that override them load."""
from .base_config import BaseConfig, namespace as ns

_grid_x = [round(-0.8 + 0.1 * i, 1) for i in range(17)]
_grid_y = [round(-0.5 + 0.1 * i, 1) for i in range(11)]


class LeggedRobotCfg(BaseConfig):
    env = ns("env", num_envs=4096, num_observations=235, num_privileged_obs=None, num_actions=12, env_spacing=3.0,
             send_timeouts=True, episode_length_s=20)
    terrain = ns("terrain", mesh_type="trimesh", horizontal_scale=0.1, vertical_scale=0.005, border_size=25, curriculum=True,
                 static_friction=1.0, dynamic_friction=1.0, restitution=0.0, measure_heights=True, measured_points_x=_grid_x,
                 measured_points_y=_grid_y, selected=False, terrain_kwargs=None, max_init_terrain_level=5, terrain_length=8.0,
                 terrain_width=8.0, num_rows=10, num_cols=20, terrain_proportions=[0.1, 0.1, 0.35, 0.25, 0.2],
                 slope_treshold=0.75)
    commands = ns("commands", curriculum=False, max_curriculum=1.0, num_commands=4, resampling_time=10.0, heading_command=True,
                  ranges=ns("ranges", lin_vel_x=[-1.0, 1.0], lin_vel_y=[-1.0, 1.0], ang_vel_yaw=[-1, 1], heading=[-3.14, 3.14]))
    init_state = ns("init_state", pos=[0.0, 0.0, 1.0], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0],
                    default_joint_angles={"joint_a": 0.0, "joint_b": 0.0})
    control = ns("control", stiffness={"joint_a": 10.0, "joint_b": 15.0}, damping={"joint_a": 1.0, "joint_b": 1.5},
                 action_scale=0.5, decimation=4)
    asset = ns("asset", file="", name="legged_robot", foot_name="None", penalize_contacts_on=[], terminate_after_contacts_on=[],
               disable_gravity=False, collapse_fixed_joints=True, fix_base_link=False, default_dof_drive_mode=3, self_collisions=0,
               replace_cylinder_with_capsule=True, flip_visual_attachments=True, density=0.001, angular_damping=0.0,
               linear_damping=0.0, max_angular_velocity=1000.0, max_linear_velocity=1000.0, armature=0.0, thickness=0.01)
    domain_rand = ns("domain_rand", randomize_friction=True, friction_range=[0.5, 1.25], randomize_base_mass=False,
                     added_mass_range=[-1.0, 1.0], push_robots=True, push_interval_s=15, max_push_vel_xy=1.0)
    rewards = ns("rewards", only_positive_rewards=True, tracking_sigma=0.25, max_contact_force=100.0,
                 scales=ns("scales", termination=-0.0, tracking_lin_vel=1.0, tracking_ang_vel=0.5, lin_vel_z=-2.0, ang_vel_xy=-0.05,
                           orientation=-0.0, torques=-0.00001, dof_vel=-0.0, dof_acc=-2.5e-7, base_height=-0.0, feet_air_time=1.0,
                           collision=-1.0, feet_stumble=-0.0, action_rate=-0.0, stand_still=-0.0))
    normalization = ns("normalization", clip_observations=100.0, clip_actions=100.0,
                       obs_scales=ns("obs_scales", lin_vel=2.0, ang_vel=0.25, dof_pos=1.0, dof_vel=0.05, height_measurements=5.0))
    noise = ns("noise", add_noise=True, noise_level=1.0,
               noise_scales=ns("noise_scales", dof_pos=0.01, dof_vel=1.5, lin_vel=0.1, ang_vel=0.2, gravity=0.05,
                               height_measurements=0.1))
    viewer = ns("viewer", ref_env=0, pos=[10, 0, 6], lookat=[11.0, 5, 3.0])
    sim = ns("sim", dt=0.005, substeps=1, gravity=[0.0, 0.0, -9.81], up_axis=1,
             physx=ns("physx", num_threads=10, solver_type=1, num_position_iterations=4, num_velocity_iterations=0,
                      contact_offset=0.01, rest_offset=0.0, bounce_threshold_velocity=0.5, max_depenetration_velocity=1.0,
                      max_gpu_contact_pairs=2 ** 23, default_buffer_size_multiplier=5, contact_collection=2))


class LeggedRobotCfgPPO(BaseConfig):
    seed = 1
    runner_class_name = "OnPolicyRunner"
    policy = ns("policy", init_noise_std=1.0, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128])
    algorithm = ns("algorithm", value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2, entropy_coef=0.01,
                   num_learning_epochs=5, num_mini_batches=4, learning_rate=1.0e-3, schedule="adaptive", gamma=0.99, lam=0.95,
                   desired_kl=0.01, max_grad_norm=1.0)
    runner = ns("runner", policy_class_name="ActorCritic", algorithm_class_name="PPO", num_steps_per_env=24, max_iterations=1500,
                save_interval=100, experiment_name="test", run_name="", resume=False, load_run=-1, checkpoint=-1, resume_path=None)
