"""Gaussian MLP actor + MLP critic (hyper-parameters: reference
``legged_robot_config.py:204-213``, flat override ``anymal_c_flat_config.py:62-65``)."""
import torch
import torch.nn as nn
from torch.distributions import Normal

_ACT = {"elu": nn.ELU, "relu": nn.ReLU, "selu": nn.SELU, "lrelu": nn.LeakyReLU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid}


def _mlp(n_in, hidden, n_out, act):
    layers, last = [], n_in
    for h in hidden:
        layers += [nn.Linear(last, h), _ACT[act]()]
        last = h
    layers.append(nn.Linear(last, n_out))
    return nn.Sequential(*layers)


class ActorCritic(nn.Module):
    is_recurrent = False

    def __init__(self, num_actor_obs, num_critic_obs, num_actions, actor_hidden_dims=(256, 256, 256),
                 critic_hidden_dims=(256, 256, 256), activation="elu", init_noise_std=1.0, **kwargs):
        super().__init__()
        if kwargs:
            print("ActorCritic.__init__ got unexpected arguments, which will be ignored: " + str(list(kwargs.keys())))
        self.actor = _mlp(num_actor_obs, list(actor_hidden_dims), num_actions, activation)
        self.critic = _mlp(num_critic_obs, list(critic_hidden_dims), 1, activation)
        self.std = nn.Parameter(init_noise_std * torch.ones(num_actions))
        self.distribution = None
        Normal.set_default_validate_args(False)     # no host-syncing argument checks (also keeps act() graph-capturable)

    def reset(self, dones=None):
        pass

    @property
    def action_mean(self):
        return self.distribution.mean

    @property
    def action_std(self):
        return self.distribution.stddev

    @property
    def entropy(self):
        return self.distribution.entropy().sum(dim=-1)

    def update_distribution(self, observations):
        mean = self.actor(observations)
        self.distribution = Normal(mean, mean * 0.0 + self.std)

    def act(self, observations, **kwargs):
        self.update_distribution(observations)
        # == self.distribution.sample(), written without torch.normal(tensor, tensor), whose std >= 0 check
        # synchronises with the host (and therefore cannot be captured into a HIP graph)
        mean = self.distribution.mean
        return mean + self.distribution.stddev * torch.randn_like(mean)

    def get_actions_log_prob(self, actions):
        return self.distribution.log_prob(actions).sum(dim=-1)

    def act_inference(self, observations):
        return self.actor(observations)

    def evaluate(self, critic_observations, **kwargs):
        return self.critic(critic_observations)
