"""CPU-only: the `humanoid` import surface mirrors the reference's -- module paths, exported
names, config values (compared with the reference's own class_to_dict dump), registry and error conventions."""
import inspect
import json
import os

import pytest
import torch


def test_import_paths_and_exports():
    import humanoid
    from humanoid.envs import LEGGED_GYM_ROOT_DIR, LeggedRobot, XBotLCfg, XBotLCfgPPO, XBotLFreeEnv, task_registry  # noqa: F401
    from humanoid.utils import get_args, task_registry as tr2, export_policy_as_jit, Logger, class_to_dict, set_seed  # noqa: F401
    from humanoid.utils import get_load_path, update_class_from_dict, wrap_to_pi, quat_apply_yaw, Terrain  # noqa: F401
    from humanoid.algo import VecEnv, PPO, OnPolicyRunner, ActorCritic, RolloutStorage  # noqa: F401
    assert tr2 is task_registry and "humanoid_ppo" in task_registry.task_classes
    assert issubclass(XBotLFreeEnv, LeggedRobot)
    assert list(inspect.signature(XBotLFreeEnv.__init__).parameters)[1:] == ["cfg", "sim_params", "physics_engine", "sim_device", "headless"]
    assert list(inspect.signature(OnPolicyRunner.__init__).parameters)[1:] == ["env", "train_cfg", "log_dir", "device"]
    assert list(inspect.signature(RolloutStorage.__init__).parameters)[1:] == [
        "num_envs", "num_transitions_per_env", "obs_shape", "privileged_obs_shape", "actions_shape", "device"]
    # the reference's positional/keyword order, then this repo's keyword-only-in-practice extension (denoising head weight)
    assert list(inspect.signature(PPO.__init__).parameters)[1:] == [
        "actor_critic", "num_learning_epochs", "num_mini_batches", "clip_param", "gamma", "lam", "value_loss_coef", "entropy_coef",
        "learning_rate", "max_grad_norm", "use_clipped_value_loss", "schedule", "desired_kl", "device", "denoise_coef"]
    for m in ("step", "reset", "get_observations", "get_privileged_observations"):
        assert hasattr(LeggedRobot, m)
    for m in ("act", "process_env_step", "compute_returns", "update", "init_storage", "test_mode", "train_mode"):
        assert hasattr(PPO, m)
    for m in ("learn", "save", "load", "get_inference_policy", "get_inference_critic", "log"):
        assert hasattr(OnPolicyRunner, m)


def test_configs_equal_reference_dump(golden_dir):
    from humanoid.envs import XBotLCfg, XBotLCfgPPO
    from humanoid.utils import class_to_dict
    ref = json.load(open(os.path.join(golden_dir, "config_dump.json")))
    mine = json.loads(json.dumps(dict(env=class_to_dict(XBotLCfg()), train=class_to_dict(XBotLCfgPPO()))))
    assert mine["train"] == ref["train"]
    assert mine["env"] == ref["env"]
    # alphabetical flattening fixes the reward accumulation order
    assert list(class_to_dict(XBotLCfg().rewards.scales)) == sorted(class_to_dict(XBotLCfg().rewards.scales))


def test_actor_critic_state_dict_and_jit_export(tmp_path, golden_dir):
    from humanoid.algo import ActorCritic
    from humanoid.utils import export_policy_as_jit
    K = json.load(open(os.path.join(golden_dir, "constants.json")))
    ac = ActorCritic(705, 219, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[768, 256, 128])
    assert list(ac.state_dict().keys()) == K["state_dict_keys"]
    assert sum(p.numel() for p in ac.parameters()) == 926105
    import numpy as np
    G = np.load(os.path.join(golden_dir, "policy_example.npz"))
    sd = ac.state_dict()
    for i in (0, 2, 4, 6):
        sd["actor.%d.weight" % i] = torch.from_numpy(G["w_%d_weight" % i])
        sd["actor.%d.bias" % i] = torch.from_numpy(G["w_%d_bias" % i])
    ac.load_state_dict(sd)
    export_policy_as_jit(ac, str(tmp_path))
    pol = torch.jit.load(str(tmp_path / "policy_1.pt"))
    y = pol(torch.zeros(1, 705))                     # sim2sim.py:192,147 usage
    np.testing.assert_allclose(y.detach().numpy(), G["y_zeros"], rtol=1e-5, atol=1e-6)
    with pytest.raises(RuntimeError):
        ac.act(torch.zeros(2, 705))                  # un-bound: no silent CPU training path


def test_error_conventions():
    from humanoid.utils import task_registry, get_load_path
    from humanoid.utils.helpers import get_args
    with pytest.raises(ValueError):
        task_registry.make_env("nope", args=get_args([]))
    with pytest.raises(ValueError):
        task_registry.make_alg_runner(env=None, name=None, args=get_args([]), train_cfg=None)
    with pytest.raises(ValueError):
        get_load_path("/nonexistent_dir_for_runs")
    from humanoid.algo import PPO, ActorCritic
    with pytest.raises(RuntimeError):
        PPO(ActorCritic(4, 4, 2, [8], [8]), device="cpu")     # fails loudly: no CPU fallback


def test_get_args_flags():
    from humanoid.utils import get_args
    a = get_args(["--task=humanoid_ppo", "--headless", "--num_envs", "64", "--seed", "3", "--max_iterations", "2",
                  "--rl_device", "cuda:0", "--sim_device", "cuda:0", "--run_name", "v1", "--resume", "--load_run", "x", "--checkpoint", "5"])
    assert (a.task, a.headless, a.num_envs, a.seed, a.max_iterations, a.run_name, a.resume, a.load_run, a.checkpoint) == \
           ("humanoid_ppo", True, 64, 3, 2, "v1", True, "x", 5)
    assert get_args([]).task == "XBotL_free"          # the reference's (unregistered) default, App. A item 15


def test_get_load_path_ordering(tmp_path):
    from humanoid.utils import get_load_path
    for run in ("Dec30_10-00-00_a", "Jan02_09-00-00_b", "exported"):
        os.makedirs(tmp_path / run)
    for m in ("model_0.pt", "model_100.pt", "model_20.pt"):
        (tmp_path / "Dec30_10-00-00_a" / m).write_text("x")
    (tmp_path / "Jan02_09-00-00_b" / "model_5.pt").write_text("x")
    assert get_load_path(str(tmp_path)).endswith(os.path.join("Dec30_10-00-00_a", "model_100.pt"))   # month-aware sort
    assert get_load_path(str(tmp_path), load_run="Jan02_09-00-00_b").endswith("model_5.pt")
    assert get_load_path(str(tmp_path), checkpoint=20).endswith("model_20.pt")
