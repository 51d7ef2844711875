"""Config / CLI / checkpoint plumbing with the reference's names (utils/helpers.py:44-253), free of isaacgym."""
import argparse
import copy
import datetime
import os
import random
import types

import numpy as np
import torch


def class_to_dict(obj) -> dict:
    """Flatten a config object; iterates dir(obj), so keys come out ALPHABETICALLY -- this fixes the order in
    which reward terms are accumulated."""
    if not hasattr(obj, "__dict__"):
        return obj
    out = {}
    for key in dir(obj):
        if key.startswith("_"):
            continue
        val = getattr(obj, key)
        out[key] = [class_to_dict(v) for v in val] if isinstance(val, list) else class_to_dict(val)
    return out


def update_class_from_dict(obj, values):
    for key, val in values.items():
        if isinstance(getattr(obj, key, None), type):
            update_class_from_dict(getattr(obj, key), val)
        else:
            setattr(obj, key, val)


def set_seed(seed):
    if seed == -1:
        seed = np.random.randint(0, 10000)
    print("Setting seed: {}".format(seed))
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    return seed


def shard_seed(seed):
    """The seed of THIS rank's env shard.  The reference is single-process (its `--horovod` flag is dead, helpers.py:207-212);
    with one process per GPU every rank must own an independent stream of command resamples, pushes, observation noise,
    reset offsets, frictions and base masses -- the same `cfg.seed` on every rank would make the data-parallel shards copies
    of each other up to the policy noise.  Rank 0 keeps the configured seed (single-GPU runs are unchanged)."""
    rank = 0
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        rank = torch.distributed.get_rank()
    return seed if seed == -1 else int(seed) + 100003 * rank


def init_distributed(args):
    """One process per GPU under `python -m torch.distributed.run` (WORLD_SIZE > 1 in the environment): join the process group (backend
    "nccl" = RCCL over xGMI; HGYM_DIST_BACKEND=gloo lets ranks share a GPU in tests), bind this rank to ITS GPU and point --sim_device /
    --rl_device at it.  Returns (rank, world_size); in a single process (0, 1) and nothing is touched.  The reference is single-process (its
    `--horovod` flag is dead, helpers.py:207-212): this is the N > 1 entry of the native stack -- envs shard across the ranks (every rank
    draws its own stream: shard_seed), PPO.update exchanges [gradient | KL] once per minibatch (algo/ppo/dist_utils.py)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    import torch.distributed as dist
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("init_distributed: no GPU visible (the hot path has no CPU fallback)")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if not dist.is_initialized():
        backend = os.environ.get("HGYM_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    dev = "cuda:%d" % local
    args.sim_device = args.rl_device = dev
    args.compute_device_id = args.sim_device_id = local
    return rank, world


def parse_sim_params(args, cfg):
    """The reference builds gymapi.SimParams for PhysX here; the synthetic physics backend only needs dt."""
    sim = cfg.get("sim", {}) if isinstance(cfg, dict) else {}
    return types.SimpleNamespace(dt=sim.get("dt", 0.005), substeps=sim.get("substeps", 1),
                                 use_gpu_pipeline=getattr(args, "use_gpu_pipeline", True), physx=sim.get("physx", {}))


def get_load_path(root, load_run=-1, checkpoint=-1):
    """Newest run directory (names start with e.g. 'Sep22_10-31-05_') and highest model_<it>.pt inside it."""
    def run_key(name):
        return (datetime.datetime.strptime(name[:3], "%b").month, int(name[3:5]), name[6:])
    try:
        runs = [r for r in os.listdir(root) if r != "exported"]
        try:
            runs.sort(key=run_key)
        except ValueError as e:
            print("WARNING - Could not sort runs by month: " + str(e))
            runs.sort()
        last_run = os.path.join(root, runs[-1])
    except Exception:
        raise ValueError("No runs in this directory: " + root)
    run_dir = last_run if load_run == -1 else os.path.join(root, load_run)
    if checkpoint == -1:
        models = sorted((f for f in os.listdir(run_dir) if "model" in f), key=lambda m: "{0:0>15}".format(m))
        model = models[-1]
    else:
        model = "model_{}.pt".format(checkpoint)
    return os.path.join(run_dir, model)


def update_cfg_from_args(env_cfg, cfg_train, args):
    if env_cfg is not None and args.num_envs is not None:
        env_cfg.env.num_envs = args.num_envs
    if cfg_train is not None:
        if args.seed is not None:
            cfg_train.seed = args.seed
        r = cfg_train.runner
        for name in ("max_iterations", "experiment_name", "run_name", "load_run", "checkpoint"):
            if getattr(args, name, None) is not None:
                setattr(r, name, getattr(args, name))
        if args.resume:
            r.resume = args.resume
    return env_cfg, cfg_train


def get_args(argv=None):
    """The reference's flags (helpers.py:167-245) plus the ones gymutil.parse_arguments used to add."""
    p = argparse.ArgumentParser(description="RL Policy")
    p.add_argument("--task", type=str, default="XBotL_free")
    p.add_argument("--resume", action="store_true", default=False)
    p.add_argument("--experiment_name", type=str)
    p.add_argument("--run_name", type=str)
    p.add_argument("--load_run", type=str)
    p.add_argument("--checkpoint", type=int)
    p.add_argument("--headless", action="store_true", default=False)
    p.add_argument("--horovod", action="store_true", default=False)
    p.add_argument("--rl_device", type=str, default="cuda:0")
    p.add_argument("--num_envs", type=int)
    p.add_argument("--seed", type=int)
    p.add_argument("--max_iterations", type=int)
    # gymutil's own
    p.add_argument("--sim_device", type=str, default="cuda:0")
    p.add_argument("--pipeline", type=str, default="gpu")
    p.add_argument("--graphics_device_id", type=int, default=0)
    p.add_argument("--physx", action="store_true", default=True)
    p.add_argument("--flex", action="store_true", default=False)
    p.add_argument("--num_threads", type=int, default=0)
    p.add_argument("--subscenes", type=int, default=0)
    p.add_argument("--slices", type=int, default=0)
    args = p.parse_args(argv)
    dev = args.sim_device
    args.sim_device_type, _, idx = dev.partition(":")
    args.compute_device_id = int(idx) if idx else 0
    args.sim_device_id = args.compute_device_id
    args.use_gpu = args.sim_device_type == "cuda"
    args.use_gpu_pipeline = args.pipeline in ("gpu", "GPU")
    args.physics_engine = "physx"
    args.device = args.sim_device_type
    return args


def export_policy_as_jit(actor_critic, path):
    os.makedirs(path, exist_ok=True)
    path = os.path.join(path, "policy_1.pt")
    model = copy.deepcopy(actor_critic.actor).to("cpu")
    torch.jit.script(model).save(path)
