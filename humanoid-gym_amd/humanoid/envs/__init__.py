"""Task registration (reference envs/__init__.py:33-42)."""
from humanoid import LEGGED_GYM_ROOT_DIR, LEGGED_GYM_ENVS_DIR
from .base.legged_robot import LeggedRobot

from .custom.humanoid_config import XBotLCfg, XBotLCfgPPO, XBotLDWLCfgPPO
from .custom.humanoid_env import XBotLFreeEnv

from humanoid.utils.task_registry import task_registry

task_registry.register("humanoid_ppo", XBotLFreeEnv, XBotLCfg(), XBotLCfgPPO())
task_registry.register("humanoid_dwl_ppo", XBotLFreeEnv, XBotLCfg(), XBotLDWLCfgPPO())     # + denoising head (native extension)
