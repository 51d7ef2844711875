"""name -> (env class, env cfg, train cfg); builds the env and the runner (reference utils/task_registry.py:44-160)."""
import os
from datetime import datetime

from humanoid import LEGGED_GYM_ROOT_DIR
from humanoid.algo import OnPolicyRunner  # noqa: F401  (resolved by name from the train config)
from .helpers import get_args, update_cfg_from_args, class_to_dict, get_load_path, set_seed, shard_seed, parse_sim_params


class TaskRegistry:
    def __init__(self):
        self.task_classes, self.env_cfgs, self.train_cfgs = {}, {}, {}

    def register(self, name, task_class, env_cfg, train_cfg):
        self.task_classes[name], self.env_cfgs[name], self.train_cfgs[name] = task_class, env_cfg, train_cfg

    def get_task_class(self, name):
        return self.task_classes[name]

    def get_cfgs(self, name):
        env_cfg, train_cfg = self.env_cfgs[name], self.train_cfgs[name]
        env_cfg.seed = train_cfg.seed
        return env_cfg, train_cfg

    def make_env(self, name, args=None, env_cfg=None):
        if args is None:
            args = get_args()
        if name not in self.task_classes:
            raise ValueError(f"Task with name: {name} was not registered")
        if env_cfg is None:
            env_cfg, _ = self.get_cfgs(name)
        env_cfg, _ = update_cfg_from_args(env_cfg, None, args)
        # as in the reference, --seed reaches the train cfg only (helpers.py:159-160); the env keeps cfg.seed.  With one process
        # per GPU the rank is folded in (helpers.shard_seed), here for the torch / numpy draws of create_sim (friction, base mass,
        # terrain levels) and in LeggedRobot._native_config for the device generator's key
        set_seed(shard_seed(env_cfg.seed))
        sim_params = parse_sim_params(args, {"sim": class_to_dict(env_cfg.sim)})
        env = self.task_classes[name](cfg=env_cfg, sim_params=sim_params, physics_engine=args.physics_engine,
                                      sim_device=args.sim_device, headless=args.headless)
        self.env_cfg_for_wandb = env_cfg
        return env, env_cfg

    def make_alg_runner(self, env, name=None, args=None, train_cfg=None, log_root="default"):
        if args is None:
            args = get_args()
        if train_cfg is None:
            if name is None:
                raise ValueError("Either 'name' or 'train_cfg' must be not None")
            _, train_cfg = self.get_cfgs(name)
        elif name is not None:
            print(f"'train_cfg' provided -> Ignoring 'name={name}'")
        _, train_cfg = update_cfg_from_args(None, train_cfg, args)
        stamp = datetime.now().strftime("%b%d_%H-%M-%S") + "_" + train_cfg.runner.run_name
        if log_root == "default":
            log_root = os.path.join(LEGGED_GYM_ROOT_DIR, "logs", train_cfg.runner.experiment_name)
            log_dir = os.path.join(log_root, stamp)
        elif log_root is None:
            log_dir = None
        else:
            log_dir = os.path.join(log_root, stamp)
        all_cfg = {**class_to_dict(train_cfg), **class_to_dict(getattr(self, "env_cfg_for_wandb", env.cfg))}
        runner_class = eval(all_cfg["runner_class_name"])
        runner = runner_class(env, all_cfg, log_dir, device=args.rl_device)
        if train_cfg.runner.resume:
            # (log_root=None -- a rank > 0 of a multi-GPU run, which does not log -- still resumes from the experiment's default directory)
            load_root = log_root if log_root is not None else os.path.join(LEGGED_GYM_ROOT_DIR, "logs", train_cfg.runner.experiment_name)
            resume_path = get_load_path(load_root, load_run=train_cfg.runner.load_run, checkpoint=train_cfg.runner.checkpoint)
            print(f"Loading model from: {resume_path}")
            runner.load(resume_path, load_optimizer=False)
        return runner, train_cfg


task_registry = TaskRegistry()
