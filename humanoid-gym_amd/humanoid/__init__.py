"""`humanoid` -- the reference's import surface (humanoid.envs / humanoid.algo / humanoid.utils) over the
MI355X hot path in libhgym_hip.so.  Same module paths, class names and signatures as roboterax/humanoid-gym
(reference humanoid/__init__.py:33-37) so that scripts/train.py and scripts/play.py run unchanged."""
import os

LEGGED_GYM_ROOT_DIR = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
LEGGED_GYM_ENVS_DIR = os.path.join(LEGGED_GYM_ROOT_DIR, "humanoid", "envs")
