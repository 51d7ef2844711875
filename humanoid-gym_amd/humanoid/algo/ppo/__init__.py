from .ppo import PPO
from .on_policy_runner import OnPolicyRunner
from .actor_critic import ActorCritic
from .rollout_storage import RolloutStorage
