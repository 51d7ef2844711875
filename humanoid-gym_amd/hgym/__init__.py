"""hgym: thin Python layer over libhgym_hip.so (ctypes).  Importing it requires the built HIP library."""
from . import _lib
from ._lib import lib, check, HgymError
from .env_buffers import EnvBuffers, default_env_config
from .net import NetBuffers, make_net_config, make_ppo_config, make_batch
