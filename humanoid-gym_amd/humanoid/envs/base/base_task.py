"""VecEnv base: device, sizes, the buffers the runner reads (reference envs/base/base_task.py:41-174 minus the
viewer / camera, which are graphics and outside the hot path)."""
import torch


class BaseTask:
    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless):
        self.sim_params = sim_params
        self.physics_engine = physics_engine
        self.sim_device = sim_device
        self.headless = headless
        self.device = sim_device
        self.num_envs = cfg.env.num_envs
        self.num_obs = cfg.env.num_observations
        self.num_privileged_obs = cfg.env.num_privileged_obs
        self.num_actions = cfg.env.num_actions
        self.extras = {}
        self.gym = None       # attributes play.py pokes at; there is no PhysX handle behind them
        self.sim = None
        self.viewer = None
        self.envs = []
        self.enable_viewer_sync = False
        self.create_sim()

    def create_sim(self):
        raise NotImplementedError

    def get_observations(self):
        return self.obs_buf

    def get_privileged_observations(self):
        return self.privileged_obs_buf

    def reset_idx(self, env_ids):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def step(self, actions):
        raise NotImplementedError

    def render(self, sync_frame_time=True):
        return None

    def set_camera(self, position, lookat):
        return None
