"""ActorCritic with the reference's interface and state_dict keys (algo/ppo/actor_critic.py:36-128):
`std`, `actor.{0,2,4,6}.{weight,bias}`, `critic.{0,2,4,6}.{weight,bias}`.

The module is an ordinary nn.Module so checkpoints, `copy.deepcopy(actor_critic.actor)` + TorchScript export
(utils/helpers.py export_policy_as_jit) and CPU inference of the exported actor keep working.  For training,
`bind()` re-points every parameter at its slice of the flat fp32 master vector of a hgym.NetBuffers, after which
act / evaluate / log-prob run through libhgym_hip.so (MFMA forward, fused Gaussian epilogue) and the optimiser
kernels update the very memory the module's parameters alias.
"""
import math

import torch
import torch.nn as nn

_HALF_LOG_2PI = 0.5 * math.log(2 * math.pi)


def _mlp(sizes, activation):
    layers = []
    for i in range(len(sizes) - 1):
        layers.append(nn.Linear(sizes[i], sizes[i + 1]))
        if i < len(sizes) - 2:
            layers.append(activation)
    return nn.Sequential(*layers)


class ActorCritic(nn.Module):
    def __init__(self, num_actor_obs, num_critic_obs, num_actions, actor_hidden_dims=[256, 256, 256],
                 critic_hidden_dims=[256, 256, 256], init_noise_std=1.0, activation=nn.ELU(), denoiser_hidden_dims=None,
                 denoiser_targets=0, **kwargs):
        """denoiser_hidden_dims / denoiser_targets (native extension, BASELINE configs[4]): an auxiliary head
        obs -> denoiser_hidden_dims -> denoiser_targets that regresses the newest `denoiser_targets` columns of the
        privileged observation (the clean single-frame privileged state) from the noisy observation history; trained jointly
        with PPO (PPO(denoise_coef=...)).  The reference has no code for it (README.md:113); off by default."""
        if kwargs:
            print("ActorCritic.__init__ got unexpected arguments, which will be ignored: " + str(list(kwargs.keys())))
        super().__init__()
        if not isinstance(activation, nn.ELU):
            raise NotImplementedError("the MFMA epilogue implements ELU (the reference default); got %r" % (activation,))
        self.num_actor_obs, self.num_critic_obs, self.num_actions = num_actor_obs, num_critic_obs, num_actions
        self.actor_hidden_dims, self.critic_hidden_dims = list(actor_hidden_dims), list(critic_hidden_dims)
        self.actor = _mlp([num_actor_obs] + self.actor_hidden_dims + [num_actions], activation)
        self.critic = _mlp([num_critic_obs] + self.critic_hidden_dims + [1], activation)
        self.denoiser_hidden_dims = list(denoiser_hidden_dims) if denoiser_hidden_dims else None
        self.denoiser_targets = int(denoiser_targets) if self.denoiser_hidden_dims else 0
        if self.denoiser_hidden_dims:
            self.denoiser = _mlp([num_actor_obs] + self.denoiser_hidden_dims + [self.denoiser_targets], activation)
        print(f"Actor MLP: {self.actor}")
        print(f"Critic MLP: {self.critic}")
        self.std = nn.Parameter(init_noise_std * torch.ones(num_actions))
        self._net = None          # hgym.NetBuffers once bound
        self._last = None         # outputs of the last act(): mu, sigma, logp, values
        self._sample_seed = 0
        self._sample_step = None

    # ------------------------------------------------------------------ binding to the HIP path
    def bind(self, net):
        """Alias every parameter to its slice of net.params (state_dict order) and keep them in sync."""
        with torch.no_grad():
            for name, p in self.named_parameters():
                view = net.views[name]
                view.copy_(p.detach().to(view.device))
                p.data = view
        self._net = net
        net.sync_shadow()
        return self

    @property
    def bound(self):
        return self._net is not None

    def load_state_dict(self, state_dict, strict=True):
        out = super().load_state_dict(state_dict, strict=strict)
        if self._net is not None:
            self._net.sync_shadow()
        return out

    def _need_net(self):
        if self._net is None:
            raise RuntimeError("ActorCritic is not bound to the HIP network (PPO binds it); there is no CPU training path")
        return self._net

    # ------------------------------------------------------------------ reference API
    @staticmethod
    def init_weights(sequential, scales):
        [torch.nn.init.orthogonal_(m.weight, gain=scales[i]) for i, m in
         enumerate(mod for mod in sequential if isinstance(mod, nn.Linear))]

    def reset(self, dones=None):
        pass

    def forward(self):
        raise NotImplementedError

    @property
    def action_mean(self):
        return self._last["mu"]

    @property
    def action_std(self):
        return self._last["sigma"]

    @property
    def entropy(self):
        return (0.5 + _HALF_LOG_2PI + torch.log(self._last["sigma"])).sum(dim=-1)

    def update_distribution(self, observations):
        net = self._need_net()
        mu = net.forward(0, observations.contiguous())
        self._last = dict(mu=mu, sigma=mu * 0.0 + self.std.detach())

    def act(self, observations, critic_observations=None, out=None, env_fin=None, shadow=None, **kwargs):
        """Sample actions.  With critic_observations the critic runs in the same call (what PPO.act needs)."""
        net = self._need_net()
        if critic_observations is None:
            self.update_distribution(observations)
            z = torch.randn_like(self._last["mu"])
            a = self._last["mu"] + self._last["sigma"] * z
            self._last["actions"] = a
            return a
        self._last = net.act(observations.contiguous(), critic_observations.contiguous(), seed=self._sample_seed,
                             step_counter=self._sample_step, out=out, env_fin=env_fin, shadow=shadow)
        return self._last["actions"]

    def get_actions_log_prob(self, actions):
        if self._last is not None and self._last.get("actions") is actions and "logp" in self._last:
            return self._last["logp"]
        mu, sg = self._last["mu"], self._last["sigma"]
        return (-((actions - mu) ** 2) / (2 * sg ** 2) - torch.log(sg) - _HALF_LOG_2PI).sum(dim=-1)

    def act_inference(self, observations):
        if self._net is not None and observations.is_cuda:
            return self._net.forward(0, observations.contiguous())
        return self.actor(observations)        # exported-policy / CPU evaluation plumbing (BASELINE config #1)

    def evaluate(self, critic_observations, **kwargs):
        return self._need_net().forward(1, critic_observations.contiguous())

    def denoise(self, observations):
        """The auxiliary head's estimate of the clean privileged frame from the (noisy) observation history."""
        if not self.denoiser_hidden_dims:
            raise RuntimeError("this ActorCritic was built without a denoiser head (denoiser_hidden_dims)")
        if self._net is not None and observations.is_cuda:
            return self._net.forward(2, observations.contiguous())
        return self.denoiser(observations)
