"""PPO with GAE, clipped surrogate / value losses and the adaptive-KL learning rate
(hyper-parameters: reference ``legged_robot_config.py:215-228``)."""
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.optim as optim


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class RolloutStorage:
    class Transition:
        def __init__(self):
            self.clear()

        def clear(self):
            self.observations = self.critic_observations = self.actions = self.rewards = self.dones = None
            self.values = self.actions_log_prob = self.action_mean = self.action_sigma = None

    def __init__(self, num_envs, num_transitions_per_env, obs_shape, privileged_obs_shape, actions_shape, device="cpu"):
        T, N, self.device = num_transitions_per_env, num_envs, device
        z = lambda *s: torch.zeros(T, N, *s, device=device)
        self.observations = z(*obs_shape)
        self.privileged_observations = z(*privileged_obs_shape) if privileged_obs_shape[0] is not None else None
        self.rewards, self.dones = z(1), z(1).byte()
        self.actions, self.mu, self.sigma = z(*actions_shape), z(*actions_shape), z(*actions_shape)
        self.actions_log_prob, self.values, self.returns, self.advantages = z(1), z(1), z(1), z(1)
        self.num_transitions_per_env, self.num_envs, self.step = T, N, 0

    def add_transitions(self, t):
        if self.step >= self.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        i = self.step
        self.observations[i].copy_(t.observations)
        if self.privileged_observations is not None:
            self.privileged_observations[i].copy_(t.critic_observations)
        self.actions[i].copy_(t.actions)
        self.rewards[i].copy_(t.rewards.view(-1, 1))
        self.dones[i].copy_(t.dones.view(-1, 1))
        self.values[i].copy_(t.values)
        self.actions_log_prob[i].copy_(t.actions_log_prob.view(-1, 1))
        self.mu[i].copy_(t.action_mean)
        self.sigma[i].copy_(t.action_sigma)
        self.step += 1

    def clear(self):
        self.step = 0

    def compute_returns(self, last_values, gamma, lam):
        """GAE(gamma, lambda); advantages normalised over the GLOBAL batch: with several ranks
        the fused [returns || advantages] buffer is all-gathered once (RCCL over xGMI)."""
        adv = 0
        for step in reversed(range(self.num_transitions_per_env)):
            nxt = last_values if step == self.num_transitions_per_env - 1 else self.values[step + 1]
            not_done = 1.0 - self.dones[step].float()
            delta = self.rewards[step] + not_done * gamma * nxt - self.values[step]
            adv = delta + not_done * gamma * lam * adv
            self.returns[step] = adv + self.values[step]
        self.advantages = self.returns - self.values
        if _world() > 1:
            fused = torch.cat((self.returns.flatten(), self.advantages.flatten()))
            gathered = torch.empty(_world() * fused.numel(), device=fused.device, dtype=fused.dtype)
            dist.all_gather_into_tensor(gathered, fused)
            all_adv = gathered.view(_world(), 2, -1)[:, 1, :]
            mean, std = all_adv.mean(), all_adv.std()
        else:
            mean, std = self.advantages.mean(), self.advantages.std()
        self.advantages = (self.advantages - mean) / (std + 1e-8)

    def get_statistics(self):
        done = self.dones.clone()
        done[-1] = 1
        flat = done.permute(1, 0, 2).reshape(-1, 1)
        idx = torch.cat((flat.new_tensor([-1], dtype=torch.int64), flat.nonzero(as_tuple=False)[:, 0]))
        lens = idx[1:] - idx[:-1]
        return lens.float().mean(), self.rewards.mean()

    def mini_batch_generator(self, num_mini_batches, num_epochs=8):
        B = self.num_envs * self.num_transitions_per_env
        mb = B // num_mini_batches
        perm = torch.randperm(num_mini_batches * mb, requires_grad=False, device=self.device)
        obs = self.observations.flatten(0, 1)
        cobs = self.privileged_observations.flatten(0, 1) if self.privileged_observations is not None else obs
        act, val, ret = self.actions.flatten(0, 1), self.values.flatten(0, 1), self.returns.flatten(0, 1)
        lp, adv = self.actions_log_prob.flatten(0, 1), self.advantages.flatten(0, 1)
        mu, sig = self.mu.flatten(0, 1), self.sigma.flatten(0, 1)
        for _ in range(num_epochs):
            for i in range(num_mini_batches):
                ix = perm[i * mb:(i + 1) * mb]
                yield obs[ix], cobs[ix], act[ix], val[ix], adv[ix], ret[ix], lp[ix], mu[ix], sig[ix], (None, None), None


class PPO:
    def __init__(self, actor_critic, num_learning_epochs=1, num_mini_batches=1, clip_param=0.2, gamma=0.998, lam=0.95,
                 value_loss_coef=1.0, entropy_coef=0.0, learning_rate=1e-3, max_grad_norm=1.0,
                 use_clipped_value_loss=True, schedule="fixed", desired_kl=0.01, device="cpu"):
        self.device = device
        self.desired_kl, self.schedule, self.learning_rate = desired_kl, schedule, learning_rate
        self.actor_critic = actor_critic.to(device)
        self.storage = None
        self.optimizer = optim.Adam(self.actor_critic.parameters(), lr=learning_rate)
        self.transition = RolloutStorage.Transition()
        self.clip_param, self.num_learning_epochs, self.num_mini_batches = clip_param, num_learning_epochs, num_mini_batches
        self.value_loss_coef, self.entropy_coef = value_loss_coef, entropy_coef
        self.gamma, self.lam, self.max_grad_norm = gamma, lam, max_grad_norm
        self.use_clipped_value_loss = use_clipped_value_loss

    def init_storage(self, num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape):
        self.storage = RolloutStorage(num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape, self.device)

    def test_mode(self):
        self.actor_critic.eval()

    def train_mode(self):
        self.actor_critic.train()

    def act(self, obs, critic_obs):
        t = self.transition
        t.actions = self.actor_critic.act(obs).detach()
        t.values = self.actor_critic.evaluate(critic_obs).detach()
        t.actions_log_prob = self.actor_critic.get_actions_log_prob(t.actions).detach()
        t.action_mean = self.actor_critic.action_mean.detach()
        t.action_sigma = self.actor_critic.action_std.detach()
        t.observations, t.critic_observations = obs, critic_obs
        return t.actions

    def process_env_step(self, rewards, dones, infos):
        t = self.transition
        t.rewards = rewards.clone()
        t.dones = dones
        if "time_outs" in infos:      # bootstrap on time-outs (the env sends them: legged_robot.py:190-191)
            t.rewards += self.gamma * torch.squeeze(t.values * infos["time_outs"].unsqueeze(1).to(self.device), 1)
        self.storage.add_transitions(t)
        t.clear()
        self.actor_critic.reset(dones)

    def compute_returns(self, last_critic_obs):
        last_values = self.actor_critic.evaluate(last_critic_obs).detach()
        self.storage.compute_returns(last_values, self.gamma, self.lam)

    def _allreduce_grads(self):
        if _world() == 1:
            return
        grads = [p.grad for p in self.actor_critic.parameters() if p.grad is not None]
        flat = torch.cat([g.flatten() for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= _world()
        o = 0
        for g in grads:
            g.copy_(flat[o:o + g.numel()].view_as(g))
            o += g.numel()

    def update(self):
        mean_v, mean_s = 0.0, 0.0
        gen = self.storage.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs)
        for obs, cobs, act, tval, adv, ret, old_lp, old_mu, old_sig, _, _ in gen:
            self.actor_critic.act(obs)
            lp = self.actor_critic.get_actions_log_prob(act)
            val = self.actor_critic.evaluate(cobs)
            mu, sig, ent = self.actor_critic.action_mean, self.actor_critic.action_std, self.actor_critic.entropy
            if self.desired_kl is not None and self.schedule == "adaptive":
                with torch.inference_mode():
                    kl = torch.sum(torch.log(sig / old_sig + 1.0e-5) + (old_sig.square() + (old_mu - mu).square()) / (2.0 * sig.square()) - 0.5, dim=-1)
                    kl_mean = kl.mean()
                    if _world() > 1:
                        dist.all_reduce(kl_mean, op=dist.ReduceOp.SUM)
                        kl_mean /= _world()
                    if kl_mean > self.desired_kl * 2.0:
                        self.learning_rate = max(1e-5, self.learning_rate / 1.5)
                    elif 0.0 < kl_mean < self.desired_kl / 2.0:
                        self.learning_rate = min(1e-2, self.learning_rate * 1.5)
                    for g in self.optimizer.param_groups:
                        g["lr"] = self.learning_rate
            ratio = torch.exp(lp - torch.squeeze(old_lp))
            a = torch.squeeze(adv)
            surrogate = torch.max(-a * ratio, -a * torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param)).mean()
            if self.use_clipped_value_loss:
                vclip = tval + (val - tval).clamp(-self.clip_param, self.clip_param)
                vloss = torch.max((val - ret).pow(2), (vclip - ret).pow(2)).mean()
            else:
                vloss = (ret - val).pow(2).mean()
            loss = surrogate + self.value_loss_coef * vloss - self.entropy_coef * ent.mean()
            self.optimizer.zero_grad()
            loss.backward()
            self._allreduce_grads()
            nn.utils.clip_grad_norm_(self.actor_critic.parameters(), self.max_grad_norm)
            self.optimizer.step()
            mean_v += vloss.item()
            mean_s += surrogate.item()
        n = self.num_learning_epochs * self.num_mini_batches
        self.storage.clear()
        return mean_v / n, mean_s / n
