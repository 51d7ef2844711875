from .vec_env import VecEnv
from .ppo import *
