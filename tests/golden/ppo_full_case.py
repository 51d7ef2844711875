"""The full-width PPO-update case (XBot-L layer widths 705-512-256-128-12 / 219-768-256-128-1, humanoid_config.py:234-237):
its INPUTS, as a function of one seed.

Shared by the recorder (tests/golden/gen_fixtures.py::gen_ppo_update_full, which runs the reference's PPO on them) and by the
replaying tests (tests/test_net_gpu.py, tests/test_oracle_algo_golden.py).  926 105 initial parameters and 656 x (705 + 219)
observation values are not stored in the fixture (they would be 6 MB of incompressible floats): both sides regenerate them
from the seed with oracle/philox.py's platform-independent fills.  The fixture holds what the reference COMPUTED.

Shape: N = 82 envs x T = 8 steps = 656 samples, 4 minibatches of 164 rows = two full 64-row tiles + a ragged 36-row tail
(one full 32-row tile + 4 rows for the rollout kernel's 32-row tiling)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import philox as X  # noqa: E402

SEED = 20240922
N, T = 82, 8
ACTOR_HIDDEN, CRITIC_HIDDEN = [512, 256, 128], [768, 256, 128]
NAMES = ["std"] + ["actor.%d.%s" % (i, k) for i in (0, 2, 4, 6) for k in ("weight", "bias")] + \
        ["critic.%d.%s" % (i, k) for i in (0, 2, 4, 6) for k in ("weight", "bias")]
HYPER = dict(num_learning_epochs=2, num_mini_batches=4, clip_param=0.2, gamma=0.994, lam=0.9, value_loss_coef=1.0,
             entropy_coef=0.001, learning_rate=1e-3, max_grad_norm=1.0, use_clipped_value_loss=True, schedule="adaptive",
             desired_kl=0.01)                                   # humanoid_config.py:240-256 (XBotLCfgPPO.algorithm)


def initial_parameters(seed=SEED):
    """state_dict of the ActorCritic: nn.Linear's default init law (weights and biases U(-1/sqrt(fan_in), 1/sqrt(fan_in))),
    std = init_noise_std = 1 -- values from Philox, so that they are the same numbers on every host."""
    sd = {"std": np.ones(12, dtype=np.float32)}
    tag = 100
    for net, dims in (("actor", [705] + ACTOR_HIDDEN + [12]), ("critic", [219] + CRITIC_HIDDEN + [1])):
        for l in range(4):
            k, n = dims[l], dims[l + 1]
            b = 1.0 / np.sqrt(k)
            sd["%s.%d.weight" % (net, 2 * l)] = X.fill_uniform(seed, tag, (n, k), -b, b)
            sd["%s.%d.bias" % (net, 2 * l)] = X.fill_uniform(seed, tag + 1, (n,), -b, b)
            tag += 2
    return sd


def rollout_inputs(seed=SEED):
    """What the env hands the algorithm over T steps: clipped observations, the policy's standard-normal draws, rewards, dones,
    time-outs; plus the last privileged observation for the bootstrap value."""
    obs = np.clip(X.fill_normal(seed, 1, (T, N, 705), 1.2), -18, 18)
    priv = np.clip(X.fill_normal(seed, 2, (T, N, 219), 1.2), -18, 18)
    z = X.fill_normal(seed, 3, (T, N, 12))
    rew = X.fill_uniform(seed, 4, (T, N), 0.0, 0.2)
    done = X.fill_uniform(seed, 5, (T, N)) < 0.15
    tout = done & (X.fill_uniform(seed, 6, (T, N)) < 0.5)
    last_priv = np.clip(X.fill_normal(seed, 7, (N, 219), 1.2), -18, 18)
    perm = np.argsort(X.fill_uniform(seed, 8, (T * N,)), kind="stable").astype(np.int64)      # the minibatch permutation
    return dict(obs=obs, priv=priv, z=z, rew_in=rew, done=done, time_outs=tout, last_priv=last_priv, perm=perm)


def sample_index(name, numel, seed=SEED, k=4096):
    """Flat indices of the fp32-exact sample the fixture keeps of a large tensor (all of it if numel <= k)."""
    if numel <= k:
        return np.arange(numel)
    tag = 1000 + NAMES.index(name)
    return np.sort(np.unique((X.fill_uniform(seed, tag, (k,)) * numel).astype(np.int64)))
