"""PPO with GAE, clipped surrogate / value losses and the adaptive-KL learning rate
(hyper-parameters: reference ``legged_robot_config.py:215-228``)."""
import os

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
        if self._gae_kernel(last_values, gamma, lam):
            pass
        else:
            self._gae_torch(last_values, gamma, lam)
        self._normalise_advantages()

    def _gae_kernel(self, last_values, gamma, lam):
        """One launch of the HIP scan (``lg_gae_returns``) instead of ~6 torch kernels per rollout step; GPU only."""
        if not self.values.is_cuda:
            return False
        if getattr(self, "_lib", None) is None:
            from .. import capi
            self._lib = capi.load_library()        # on a GPU the extension is the product path: a missing build fails loudly here
        T, N = self.num_transitions_per_env, self.num_envs
        lv = last_values.reshape(-1).contiguous().float()
        rc = self._lib.lg_gae_returns(self.rewards.data_ptr(), self.values.data_ptr(), self.dones.data_ptr(), lv.data_ptr(), float(gamma), float(lam),
                                      self.returns.data_ptr(), self.advantages.data_ptr(), T, N, torch.cuda.current_stream(self.values.device).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"lg_gae_returns failed ({rc}): {self._lib.lg_last_error().decode()}")
        return True

    def _gae_torch(self, last_values, gamma, lam):
        adv = 0
        for step in reversed(range(self.num_transitions_per_env)):
            nxt = last_values if step == self.num_transitions_per_env - 1 else self.values[step + 1]
            not_done = 1.0 - self.dones[step].float()
            delta = self.rewards[step] + not_done * gamma * nxt - self.values[step]
            adv = delta + not_done * gamma * lam * adv
            self.returns[step] = adv + self.values[step]
        torch.sub(self.returns, self.values, out=self.advantages)        # in place: a captured update graph reads this buffer

    def _normalise_advantages(self):
        if _world() > 1:
            fused = torch.cat((self.returns.flatten(), self.advantages.flatten()))
            gathered = torch.empty(_world() * fused.numel(), device=fused.device, dtype=fused.dtype)
            dist.all_gather_into_tensor(gathered, fused)
            all_adv = gathered.view(_world(), 2, -1)[:, 1, :]
            mean, std = all_adv.mean(), all_adv.std()
        else:
            mean, std = self.advantages.mean(), self.advantages.std()
        self.advantages.sub_(mean).div_(std + 1e-8)

    def get_statistics(self):
        done = self.dones.clone()
        done[-1] = 1
        flat = done.permute(1, 0, 2).reshape(-1, 1)
        idx = torch.cat((flat.new_tensor([-1], dtype=torch.int64), flat.nonzero(as_tuple=False)[:, 0]))
        lens = idx[1:] - idx[:-1]
        return lens.float().mean(), self.rewards.mean()

    def mini_batch_generator(self, num_mini_batches, num_epochs=8, perm=None):
        B = self.num_envs * self.num_transitions_per_env
        mb = B // num_mini_batches
        if perm is None:
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


class _LinearSplitK(torch.autograd.Function):
    """``F.linear`` whose weight gradient is computed as a batched GEMM over row chunks.  dW = g^T x has the whole mini-batch
    (24 576 rows) as its K dimension and only out x in <= 512 x 512 outputs: hipBLASLt runs it on 16 workgroups (324 us for
    512 x 235); cutting the rows into S chunks gives S times the tiles (torch.bmm) and a cheap [S, out, in] sum."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.addmm(b, x, w.t())

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        M = x.shape[0]
        S = next((s for s in (16, 12, 8, 6, 4, 3, 2) if M % s == 0 and M // s >= 256), 1)
        g = g.contiguous()
        if S > 1:
            gw = torch.bmm(g.view(S, M // S, -1).transpose(1, 2), x.view(S, M // S, -1)).sum(0)
        else:
            gw = g.t() @ x
        gx = g @ w if ctx.needs_input_grad[0] else None
        return gx, gw, g.sum(0)


def _mlp_split_k(seq, x):
    """Apply an ``nn.Sequential`` of Linear / activation modules with ``_LinearSplitK`` in place of ``nn.Linear``."""
    for m in seq:
        x = _LinearSplitK.apply(x, m.weight, m.bias) if isinstance(m, nn.Linear) else m(x)
    return x


class PPO:
    def __init__(self, actor_critic, num_learning_epochs=1, num_mini_batches=1, clip_param=0.2, gamma=0.998, lam=0.95,
                 value_loss_coef=1.0, entropy_coef=0.0, learning_rate=1e-3, max_grad_norm=1.0,
                 use_clipped_value_loss=True, schedule="fixed", desired_kl=0.01, device="cpu", graphed_update=True, fused_loss=True):
        self.device = device
        # Single-GPU runs replay one captured HIP graph per mini-batch step (forward, losses, backward, grad clip, Adam and
        # the adaptive-KL learning rate all on the device): the flat networks' update is launch-bound (~150 tiny kernels).
        self._graph_ok = bool(graphed_update) and str(device).startswith("cuda")
        self._graph, self._graph_key, self._graph_whole = None, None, False
        self._updates_done = 0
        # ... and the surrogate / value / entropy losses with their gradients w.r.t. the network outputs come from ONE HIP kernel
        # (``lg_ppo_loss``) instead of ~100 small torch kernels; autograd only runs through the two MLPs.
        import os as _os
        self._fused_loss = bool(fused_loss) and self._graph_ok and _os.environ.get("LG_PPO_FUSED_LOSS", "1") != "0"
        # ... and for the MLP shape of the flat tasks the two networks' forward and backward run in the MFMA learner kernels
        # (lg_mlp_forward / lg_mlp_backward, rl/mlp_kernels.py): autograd is not involved at all.  LG_PPO_MLP_KERNELS=0 disables.
        self._mlp_kernels = self._fused_loss and _os.environ.get("LG_PPO_MLP_KERNELS", "1") != "0"
        self._mlp = None
        # ... and gradient clipping, the adaptive-KL learning-rate rule and Adam are two launches (lg_adam_step) instead of ~60
        # small torch kernels, updating torch.optim.Adam's own state tensors in place.  LG_PPO_ADAM_KERNEL=0 disables.
        self._adam_kernel = self._fused_loss and _os.environ.get("LG_PPO_ADAM_KERNEL", "1") != "0"
        self._fused_minibatch = _os.environ.get("LG_PPO_FUSED_MINIBATCH", "1") != "0"   # lg_ppo_minibatch instead of forward / loss / backward
        self._wide_kernels = _os.environ.get("LG_PPO_WIDE_KERNELS", "1") != "0"         # lg_mlp_wide_* for widths the LDS-resident kernels do not cover
        self._split_k = _os.environ.get("LG_PPO_SPLIT_K", "1") != "0"       # torch MLP path (wide networks): see _LinearSplitK
        self._lib = None
        self.desired_kl, self.schedule, self.learning_rate = desired_kl, schedule, learning_rate
        self.actor_critic = actor_critic.to(device)
        self.storage = None
        if self._graph_ok:
            # The CUDA generator allocates its graph-safe state tensors at the first capture in the process; if that happens
            # under torch.inference_mode (the rollout graph, rl/runner.py) they become inference tensors and a later capture
            # with autograd enabled cannot touch them.  Prime them here, in normal mode.
            # (kept alive: the generator drops the tensors again when its last graph goes away)
            self._prime = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._prime, capture_error_mode="thread_local"):
                torch.zeros(1, device=device).add_(1.0)
            self._lr = torch.tensor(float(learning_rate), device=device)
            self.optimizer = optim.Adam(self.actor_critic.parameters(), lr=self._lr, capturable=True)
        else:
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

    def _flat_grad_views(self):
        """Data-parallel kernel path: every parameter's ``.grad`` (and the policy-std gradient the loss kernel writes) becomes a
        view of ONE persistent flat buffer, with one extra slot for the mean KL.  The learner kernels then write their gradients
        straight into that buffer and a mini-batch step needs a single all-reduce over it (RCCL: one collective of
        n_params + 1 floats instead of a concatenation, an all-reduce, a scalar all-reduce and 17 copies back)."""
        params = list(self.actor_critic.parameters())
        n = sum(p.numel() for p in params)
        buf = getattr(self, "_gflat", None)
        if buf is None or buf.numel() != n + 1 or any(p.grad is None or p.grad.data_ptr() != buf[o].data_ptr() for p, o in zip(params, self._goffs)):
            buf = torch.zeros(n + 1, device=self.device)
            offs, o = [], 0
            for p in params:
                if p.grad is not None:
                    buf[o:o + p.numel()].copy_(p.grad.reshape(-1))
                p.grad = buf[o:o + p.numel()].view_as(p)
                offs.append(o)
                o += p.numel()
            self._gflat, self._goffs = buf, offs
            std = self.actor_critic.std
            self._d_std = std.grad                      # the loss kernels write d loss / d std here
            self._mlp = None                            # descriptors hold the old .grad addresses
        return self._gflat

    def _allreduce_flat(self, adaptive):
        """One collective per mini-batch step: mean gradient and (adaptive schedule) mean KL over the ranks."""
        flat = self._gflat
        if adaptive:
            flat[-1:].copy_(self._stats[2:3])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / _world())
        if adaptive:
            self._stats[2:3].copy_(flat[-1:])

    def after_optimizer_load(self):
        """The optimiser's tensors were replaced (checkpoint resume): re-link the device learning rate, drop the captured graph."""
        if self._graph_ok:
            for g in self.optimizer.param_groups:
                self._lr.copy_(torch.as_tensor(g["lr"], device=self.device).float().reshape(()))
                g["lr"] = self._lr
            self._graph, self._updates_done = None, 0

    # ------------------------------------------------------------------ captured mini-batch step
    def _flat(self):
        st = self.storage
        obs = st.observations.flatten(0, 1)
        cobs = st.privileged_observations.flatten(0, 1) if st.privileged_observations is not None else obs
        return (obs, cobs, st.actions.flatten(0, 1), st.values.flatten(0, 1), st.advantages.flatten(0, 1), st.returns.flatten(0, 1),
                st.actions_log_prob.flatten(0, 1), st.mu.flatten(0, 1), st.sigma.flatten(0, 1))

    def _fused_ready(self):
        if not self._fused_loss:
            return False
        if self._lib is None:
            from .. import capi
            self._lib = capi.load_library()        # cuda device: no silent torch substitute for a missing extension
        ac = self.actor_critic
        return bool(self._lib) and hasattr(ac, "actor") and hasattr(ac, "critic") and hasattr(ac, "std") and ac.std.numel() <= 16

    def _mlp_trainer(self):
        """MlpTrainer for the current storage / mini-batch size, or None when the networks are not the kernels' shape."""
        if not self._mlp_kernels:
            return None
        st, ac = self.storage, self.actor_critic
        obs = st.observations.flatten(0, 1)
        cobs = st.privileged_observations.flatten(0, 1) if st.privileged_observations is not None else obs
        mb = self._ix.numel()
        key = (obs.data_ptr(), cobs.data_ptr(), mb)
        if self._mlp is None or self._mlp_key != key:
            from .mlp_kernels import MlpTrainer
            if not (isinstance(ac.actor, nn.Sequential) and isinstance(ac.critic, nn.Sequential)):
                self._mlp_kernels = False
                return None
            self._mlp, self._mlp_key = MlpTrainer([ac.actor, ac.critic], [obs, cobs], mb), key
            if not self._mlp.supported:                  # not the LDS-resident shape: the layer-wise GEMM kernels take any widths
                from .mlp_kernels import WideMlpTrainer
                self._mlp = WideMlpTrainer([ac.actor, ac.critic], [obs, cobs], mb) if self._wide_kernels else None
            if self._mlp is None or not self._mlp.supported:
                self._mlp_kernels, self._mlp = False, None
                return None
        return self._mlp

    def _adam_table(self):
        """lg_adam_tensor[] over torch.optim.Adam's parameters and state, or None while the state does not exist yet (the very
        first step is torch's: it creates exp_avg / exp_avg_sq / step) or the optimiser is not plain capturable Adam."""
        from .. import capi
        opt = self.optimizer
        if len(opt.param_groups) != 1:
            return None
        g = opt.param_groups[0]
        if g.get("amsgrad") or g.get("weight_decay") or g.get("maximize") or not g.get("capturable") or g["lr"] is not self._lr:
            return None
        params = [q for q in g["params"] if q.grad is not None]
        if not params or len(params) > 32:
            return None
        table = (capi.lg_adam_tensor * len(params))()
        for i, q in enumerate(params):
            stt = opt.state.get(q)
            if not stt or "exp_avg" not in stt or not torch.is_tensor(stt["step"]) or stt["step"].dtype != torch.float32 \
                    or not stt["step"].is_cuda or q.dtype != torch.float32 or not q.is_contiguous() or not q.grad.is_contiguous():
                return None
            t = table[i]
            t.param, t.grad, t.exp_avg, t.exp_avg_sq = q.data_ptr(), q.grad.data_ptr(), stt["exp_avg"].data_ptr(), stt["exp_avg_sq"].data_ptr()
            t.step, t.numel = stt["step"].data_ptr(), q.numel()
        if getattr(self, "_adam_scratch", None) is None:
            self._adam_scratch = torch.zeros(capi.LG_ADAM_SCRATCH_FLOATS, device=self.device)
        return table

    def _mb_forward_loss_backward(self, tr):
        """MLP forward -> lg_ppo_loss -> MLP backward as separate launches (learner kernels when ``tr`` is given, torch otherwise)."""
        st, ac, ix = self.storage, self.actor_critic, self._ix
        if tr is not None:
            tr.refresh()                             # parameter / .grad addresses (stable in steady state)
            mu, val = tr.forward(ix)
        else:
            obs_all = st.observations.flatten(0, 1)
            obs = obs_all[ix]
            cobs = st.privileged_observations.flatten(0, 1)[ix] if st.privileged_observations is not None else obs
            if self._split_k and isinstance(ac.actor, nn.Sequential) and isinstance(ac.critic, nn.Sequential):
                mu, val = _mlp_split_k(ac.actor, obs), _mlp_split_k(ac.critic, cobs)
            else:
                mu, val = ac.actor(obs), ac.critic(cobs)
        mb, A = mu.shape
        if getattr(self, "_d_mu", None) is None or self._d_mu.shape != mu.shape:
            self._d_mu, self._d_val = torch.empty_like(mu), torch.empty(mb, 1, device=mu.device)
            if getattr(self, "_d_std", None) is None or self._d_std.numel() != A:
                self._d_std = torch.zeros(A, device=mu.device)
            if getattr(self, "_stats", None) is None:
                self._stats = torch.zeros(4, device=mu.device)
        d_mu, d_val = (tr.grad_outputs if tr is not None else (self._d_mu, self._d_val))
        p = lambda t: t.data_ptr()
        rc = self._lib.lg_ppo_loss(p(mu), p(ac.std), p(val), p(ix), p(st.actions), p(st.actions_log_prob), p(st.mu), p(st.sigma), p(st.advantages),
                                   p(st.values), p(st.returns), float(self.clip_param), float(self.value_loss_coef), float(self.entropy_coef),
                                   int(bool(self.use_clipped_value_loss)), p(d_mu), p(self._d_std), p(d_val), p(self._stats), int(mb), int(A),
                                   torch.cuda.current_stream(mu.device).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"lg_ppo_loss failed ({rc}): {self._lib.lg_last_error().decode()}")
        if tr is not None:
            tr.backward(ix)
        else:
            torch.autograd.backward([mu, val], [d_mu, d_val])

    def _mb_step_fused(self, phase=None):
        """Mini-batch step on the device kernels: [MLP forward -> PPO loss -> MLP backward] -> gradient clip + KL rule + Adam.
        The bracket is ONE kernel (lg_ppo_minibatch) when the networks have the learner kernels' shape; otherwise torch MLP
        passes (autograd) around lg_ppo_loss.  ``phase`` (data-parallel kernel path): "pre" = everything in front of the flat
        all-reduce, "post" = everything behind it -- the two halves are captured as HIP graphs around the eager collective."""
        st, ac, ix = self.storage, self.actor_critic, self._ix
        if phase == "post":
            return self._mb_step_fused_post(self.desired_kl is not None and self.schedule == "adaptive", self._dp_acc_done, True)
        flat_dp = _world() > 1 and self._mlp_kernels
        if flat_dp:
            self._flat_grad_views()                  # .grad tensors = views of one buffer (before the descriptors read their addresses)
        tr = self._mlp_trainer()
        flat_dp = flat_dp and tr is not None         # (torch MLP path: autograd allocates its own .grad tensors)
        p = lambda t: t.data_ptr()
        if tr is not None and self._fused_minibatch and tr.has_fused_minibatch:
            # forward + loss + backward of both networks in ONE kernel (lg_ppo_minibatch): mu / value never reach HBM
            mb, A = ix.numel(), ac.std.numel()
            if getattr(self, "_d_std", None) is None or self._d_std.numel() != A:
                self._d_std = torch.zeros(A, device=self.device)
            if getattr(self, "_stats", None) is None:
                self._stats = torch.zeros(4, device=self.device)
            tr.refresh()
            from .. import capi
            b = capi.lg_ppo_batch()
            b.actions, b.old_log_prob, b.old_mu, b.old_sigma = p(st.actions), p(st.actions_log_prob), p(st.mu), p(st.sigma)
            b.advantages, b.old_values, b.returns, b.std = p(st.advantages), p(st.values), p(st.returns), p(ac.std)
            b.clip, b.value_coef, b.entropy_coef = float(self.clip_param), float(self.value_loss_coef), float(self.entropy_coef)
            b.use_clipped_value, b.d_std, b.stats = int(bool(self.use_clipped_value_loss)), p(self._d_std), p(self._stats)
            b.loss_acc = p(self._acc)                # the kernel also keeps the update's running loss sums
            tr.ppo_minibatch(ix, b)
            acc_done = True
        else:
            self._mb_forward_loss_backward(tr)
            acc_done = False
        ac.std.grad = self._d_std
        adaptive = self.desired_kl is not None and self.schedule == "adaptive"
        if phase == "pre":
            if not flat_dp:
                raise RuntimeError("the two-graph data-parallel step needs the flat-gradient kernel path")
            self._dp_acc_done = acc_done
            if adaptive:
                self._gflat[-1:].copy_(self._stats[2:3])
            return
        if flat_dp:                                  # data-parallel ranks: mean gradient and mean KL (rollout shards are equal-sized), ONE collective
            self._allreduce_flat(adaptive)
        elif _world() > 1:
            self._allreduce_grads()
            if adaptive:
                dist.all_reduce(self._stats[2:3], op=dist.ReduceOp.SUM)
                self._stats[2:3] /= _world()
        self._mb_step_fused_post(adaptive, acc_done, False)

    def _mb_step_fused_post(self, adaptive, acc_done, after_collective):
        """Behind the gradients (and the collective): [mean over ranks] -> gradient clip + KL rule + Adam."""
        ac = self.actor_critic
        p = lambda t: t.data_ptr()
        if after_collective:                         # the eager all-reduce summed the flat buffer (gradients + KL slot)
            self._gflat.mul_(1.0 / _world())
            if adaptive:
                self._stats[2:3].copy_(self._gflat[-1:])
        table = self._adam_table() if self._adam_kernel else None
        if table is not None:
            g = self.optimizer.param_groups[0]
            rc = self._lib.lg_adam_step(table, len(table), p(self._lr), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                        float(self.max_grad_norm), p(self._stats[2:]) if adaptive else None,
                                        float(self.desired_kl) if adaptive else 0.0, p(self._adam_scratch),
                                        torch.cuda.current_stream(self.device).cuda_stream)
            if rc != 0:
                raise RuntimeError(f"lg_adam_step failed ({rc}): {self._lib.lg_last_error().decode()}")
        else:
            if adaptive:
                with torch.no_grad():
                    kl, lr = self._stats[2], self._lr
                    down, up = torch.clamp(lr / 1.5, min=1e-5), torch.clamp(lr * 1.5, max=1e-2)
                    self._lr.copy_(torch.where(kl > self.desired_kl * 2.0, down, torch.where((kl < self.desired_kl / 2.0) & (kl > 0.0), up, lr)))
            nn.utils.clip_grad_norm_(ac.parameters(), self.max_grad_norm)
            self.optimizer.step()
        if not acc_done:
            with torch.no_grad():
                self._acc[0] += self._stats[1]
                self._acc[1] += self._stats[0]

    def _zero_grad(self):
        # the learner kernels overwrite persistent .grad tensors (their addresses are baked into the captured graph)
        if not (self._mlp_kernels and self._mlp is not None):
            self.optimizer.zero_grad(set_to_none=True)

    def _mb_step(self):
        """One mini-batch step on the rows listed in ``self._ix``; everything stays on the device (no host decisions)."""
        if self._fused_ready():
            return self._mb_step_fused()
        ix = self._ix
        obs, cobs, act, tval, adv, ret, old_lp, old_mu, old_sig = (t[ix] for t in self._flat())
        ac = self.actor_critic
        ac.update_distribution(obs)
        lp = ac.get_actions_log_prob(act)
        val = ac.evaluate(cobs)
        mu, sig, ent = ac.action_mean, ac.action_std, ac.entropy
        if self.desired_kl is not None and self.schedule == "adaptive":
            with torch.no_grad():
                kl = torch.sum(torch.log(sig / old_sig + 1.0e-5) + (old_sig.square() + (old_mu - mu).square()) / (2.0 * sig.square()) - 0.5, dim=-1).mean()
                lr = self._lr
                down, up = torch.clamp(lr / 1.5, min=1e-5), torch.clamp(lr * 1.5, max=1e-2)
                self._lr.copy_(torch.where(kl > self.desired_kl * 2.0, down, torch.where((kl < self.desired_kl / 2.0) & (kl > 0.0), up, lr)))
        ratio = torch.exp(lp - torch.squeeze(old_lp))
        a = torch.squeeze(adv)
        surrogate = torch.max(-a * ratio, -a * torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param)).mean()
        if self.use_clipped_value_loss:
            vclip = tval + (val - tval).clamp(-self.clip_param, self.clip_param)
            vloss = torch.max((val - ret).pow(2), (vclip - ret).pow(2)).mean()
        else:
            vloss = (ret - val).pow(2).mean()
        loss = surrogate + self.value_loss_coef * vloss - self.entropy_coef * ent.mean()
        loss.backward()
        nn.utils.clip_grad_norm_(self.actor_critic.parameters(), self.max_grad_norm)
        self.optimizer.step()
        with torch.no_grad():
            self._acc[0] += vloss.detach()
            self._acc[1] += surrogate.detach()

    def _update_graphed(self, perm=None):
        st = self.storage
        B = st.num_envs * st.num_transitions_per_env
        mb = B // self.num_mini_batches
        key = (st.observations.data_ptr(), st.advantages.data_ptr(), st.returns.data_ptr(), mb)
        if self._graph is not None and key != self._graph_key:
            # the storage was re-allocated: the captured graph reads the old buffers; warm up eagerly again before re-capturing
            self._graph, self._mlp, self._updates_done = None, None, 0
        self._graph_key = key
        if self._graph is None and (getattr(self, "_perm_buf", None) is None or self._perm_buf.numel() != self.num_mini_batches * mb):
            # mini-batch i reads rows _perm_buf[i*mb : (i+1)*mb]: fixed addresses, so captured kernels need no index copies
            self._perm_buf = torch.zeros(self.num_mini_batches * mb, dtype=torch.int64, device=self.device)
            self._ix_one = torch.zeros(mb, dtype=torch.int64, device=self.device)      # row list of the one-step graph
            self._ix = self._ix_one
            self._acc = torch.zeros(2, device=self.device)
        self._acc.zero_()
        # The first update runs eagerly ON THE CAPTURE STREAM: it is the warm-up torch asks for before capturing autograd +
        # optimiser work (library handles / workspaces and the optimiser state get created on that stream, outside capture).
        if self._graph is None and not hasattr(self, "_gstream"):
            self._gstream = torch.cuda.Stream(device=self.device)
        cur = torch.cuda.current_stream(self.device)
        if perm is None:                             # one permutation per update, re-used by every epoch (rsl_rl)
            perm = torch.randperm(self.num_mini_batches * mb, device=self.device)
        self._perm_buf.copy_(perm[:self._perm_buf.numel()])
        views = [self._perm_buf[i * mb:(i + 1) * mb] for i in range(self.num_mini_batches)]
        if self._graph is not None and self._graph_whole:
            self._graph.replay()                     # the whole update (epochs x mini-batches) is one graph
        elif self._graph is not None:
            for _ in range(self.num_learning_epochs):
                for i in range(self.num_mini_batches):
                    self._ix_one.copy_(views[i])     # one-step graph (torch MLP path): reads _ix_one
                    self._graph.replay()
        elif self._updates_done >= 1:
            # Capture.  On the kernel path a mini-batch step is 5 launches, so all epochs x mini-batches go into ONE graph (each
            # step reading its own slice of _perm_buf); the torch MLP path (~150 launches and fresh activations per step) keeps a
            # one-step graph replayed 20 times.
            self._graph_whole = self._mlp is not None and self._adam_kernel and (not self._mlp.has_fused_minibatch or self._fused_minibatch)
            self._gstream.wait_stream(cur)
            self._zero_grad()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self._gstream, capture_error_mode="thread_local"):
                if self._graph_whole:
                    for _ in range(self.num_learning_epochs):
                        for i in range(self.num_mini_batches):
                            self._ix = views[i]
                            self._mb_step()
                else:
                    self._ix = self._ix_one
                    self._mb_step()
            self._graph = g
            self._ix = self._ix_one
            if self._graph_whole:
                g.replay()                           # the capture did not execute
            else:
                for _ in range(self.num_learning_epochs):
                    for i in range(self.num_mini_batches):
                        self._ix_one.copy_(views[i])
                        g.replay()
        else:
            for _ in range(self.num_learning_epochs):
                for i in range(self.num_mini_batches):
                    self._gstream.wait_stream(cur)
                    with torch.cuda.stream(self._gstream):
                        self._ix = views[i]
                        self._zero_grad()
                        self._mb_step()
                    cur.wait_stream(self._gstream)
            self._ix = self._ix_one
        n = self.num_learning_epochs * self.num_mini_batches
        mean_v, mean_s, lr = (float(x) for x in torch.cat((self._acc / n, self._lr.view(1))).cpu())
        self.learning_rate = lr
        self._updates_done += 1
        st.clear()
        return mean_v, mean_s

    def _update_fused_eager(self, perm=None):
        """world_size > 1, kernel path.  A mini-batch step is [backward graph] -> flat all-reduce (RCCL, eager: one collective of all
        gradients + the KL slot) -> [optimiser graph]: after an eager first update the two halves are captured once -- one "pre" graph
        per mini-batch slot (each reads its own slice of the permutation buffer), one "post" graph (mean over ranks, clip, KL rule,
        Adam) -- so an update is 2 x epochs x mini-batches graph replays + epochs x mini-batches collectives from the host, no kernel
        launch of its own.  ``LG_DP_GRAPHS=0`` keeps the eager launches (A/B, and the fallback when capture is refused)."""
        st = self.storage
        B = st.num_envs * st.num_transitions_per_env
        mb = B // self.num_mini_batches
        nmb = self.num_mini_batches
        if getattr(self, "_perm_buf", None) is None or self._perm_buf.numel() != nmb * mb:
            self._perm_buf = torch.zeros(nmb * mb, dtype=torch.int64, device=self.device)
            self._acc = torch.zeros(2, device=self.device)
            self._dp_graphs = None
        key = (st.observations.data_ptr(), st.advantages.data_ptr(), st.returns.data_ptr(), mb)
        if getattr(self, "_dp_key", None) != key:
            self._dp_key, self._dp_graphs = key, None
        self._acc.zero_()
        if perm is None:
            perm = torch.randperm(nmb * mb, device=self.device)
        self._perm_buf.copy_(perm[:nmb * mb])
        views = [self._perm_buf[i * mb:(i + 1) * mb] for i in range(nmb)]
        want_graphs = os.environ.get("LG_DP_GRAPHS", "1") != "0" and self._mlp_kernels and self._adam_kernel
        self.dp_launches = None
        if want_graphs and self._updates_done >= 1 and getattr(self, "_dp_graphs", None) is None:
            try:
                self._ix = views[0]
                if self._mlp_trainer() is None:
                    raise RuntimeError("no learner kernels for these networks")
                if not hasattr(self, "_gstream"):
                    self._gstream = torch.cuda.Stream(device=self.device)
                self._gstream.wait_stream(torch.cuda.current_stream(self.device))
                pre = []
                for i in range(nmb):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=self._gstream, capture_error_mode="thread_local"):
                        self._ix = views[i]
                        self._zero_grad()
                        self._mb_step_fused("pre")
                    pre.append(g)
                post = torch.cuda.CUDAGraph()
                with torch.cuda.graph(post, stream=self._gstream, capture_error_mode="thread_local"):
                    self._mb_step_fused("post")
                self._dp_graphs = (pre, post)
            except Exception as exc:                  # capture refused (e.g. the torch MLP path): stay eager, say so once
                self._dp_graphs = False
                self._dp_graph_error = f"{type(exc).__name__}: {exc}"
        if getattr(self, "_dp_graphs", None):
            pre, post = self._dp_graphs
            for _ in range(self.num_learning_epochs):
                for i in range(nmb):
                    pre[i].replay()
                    dist.all_reduce(self._gflat, op=dist.ReduceOp.SUM)
                    post.replay()
            n = self.num_learning_epochs * nmb
            self.dp_launches = {"graph_replays": 2 * n, "collectives": n, "kernel_launches_from_host": 0}
        else:
            for _ in range(self.num_learning_epochs):
                for i in range(nmb):
                    self._ix = views[i]
                    self._zero_grad()
                    self._mb_step()
        n = self.num_learning_epochs * nmb
        mean_v, mean_s, lr = (float(x) for x in torch.cat((self._acc / n, self._lr.view(1))).cpu())
        self.learning_rate = lr
        self._updates_done += 1
        st.clear()
        return mean_v, mean_s

    def update(self, perm=None):
        """One PPO update over the stored rollout.  ``perm`` (optional) fixes the mini-batch permutation (tests)."""
        if self._graph_ok and _world() == 1:
            return self._update_graphed(perm)
        if self._graph_ok and self._fused_ready():
            return self._update_fused_eager(perm)
        mean_v, mean_s = 0.0, 0.0
        gen = self.storage.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs, perm)
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
