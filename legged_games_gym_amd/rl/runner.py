"""``OnPolicyRunner`` with the constructor / learn / save / load / get_inference_policy surface the
reference uses (``task_registry.py:154-161``, ``scripts/train.py:43``, ``scripts/play.py:59``);
checkpoints are ``model_<it>.pt`` in ``log_dir`` so ``get_load_path`` (helpers.py:103-125) finds them."""
import os
import statistics
import time
from collections import deque

import torch

from .. import capi
from .actor_critic import ActorCritic
from .ppo import PPO


class OnPolicyRunner:
    def __init__(self, env, train_cfg, log_dir=None, device="cpu"):
        self.cfg, self.alg_cfg, self.policy_cfg = train_cfg["runner"], train_cfg["algorithm"], train_cfg["policy"]
        self.device, self.env = device, env
        num_critic_obs = env.num_privileged_obs if env.num_privileged_obs is not None else env.num_obs
        if self.cfg.get("policy_class_name", "ActorCritic") != "ActorCritic":
            raise NotImplementedError("only the feed-forward ActorCritic is bundled")
        actor_critic = ActorCritic(env.num_obs, num_critic_obs, env.num_actions, **self.policy_cfg).to(device)
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            for prm in actor_critic.parameters():       # replicas start from rank 0's weights (env seeds differ per rank)
                dist.broadcast(prm.data, src=0)
        self.alg = PPO(actor_critic, device=device, **self.alg_cfg)
        self.num_steps_per_env = self.cfg["num_steps_per_env"]
        self.save_interval = self.cfg["save_interval"]
        self.alg.init_storage(env.num_envs, self.num_steps_per_env, [env.num_obs], [env.num_privileged_obs], [env.num_actions])
        self.log_dir, self.writer = log_dir, None
        self.tot_timesteps, self.tot_time, self.current_learning_iteration = 0, 0.0, 0
        _, _ = self.env.reset()
        self._fused = self._make_fused_actor()

    def _make_fused_actor(self):
        """Rollouts with the MFMA actor (and, for the flat task, the actor fused into the step kernel) instead of torch ops;
        the critic, log-probs and the time-out bootstrap are then evaluated once per rollout on all T x N transitions."""
        env = self.env
        if not (str(self.device).startswith("cuda") and self.cfg.get("fused_rollout", True) and hasattr(env, "_sim")
                and env.num_privileged_obs is None):
            return None
        try:
            from .fused_actor import FusedActor
            fused = FusedActor(self.alg.actor_critic, self.device, seed=int(getattr(env.cfg, "seed", 1)) + 7919,
                               step_counter=env._sim.buf["step_counter"])
        except Exception as exc:                              # unsupported actor shape: torch rollouts
            print(f"[runner] MFMA actor unavailable ({type(exc).__name__}: {exc}); torch policy in the rollout")
            return None
        self._time_outs = torch.zeros(self.num_steps_per_env, env.num_envs, 1, device=self.device)
        self._fused_step = None                               # decided at the first rollout: lg_step_policy or actor kernel + lg_step
        self._rolled = None                                   # ... or the whole rollout in one launch (lg_rollout_policy)
        return fused

    def _fused_env_step(self, obs):
        """One rollout step: ``lg_step_policy`` (actor inside the step kernel) where a fused kernel exists, else actor kernel + step."""
        env, fused = self.env, self._fused
        if self._fused_step is not False:
            try:
                (actions, mean), out = env.step_policy(fused)
                self._fused_step = True
                return actions, mean, out
            except RuntimeError as exc:
                # only "no fused kernel for this sim / actor pair" (lg_step_policy rc -4) selects the two-kernel path;
                # a real HIP failure must surface
                if self._fused_step or "fused policy step" not in str(exc):
                    raise
                self._fused_step = False
        actions, mean = fused.act_with_mean(obs)
        return actions, mean, env.step(actions)

    def _critic_values(self, st):
        """critic(obs) for all stored transitions: lg_mlp_forward / lg_mlp_wide_forward when the critic has one of the learner kernels' shapes, torch otherwise."""
        ac = self.alg.actor_critic
        cobs = (st.privileged_observations if st.privileged_observations is not None else st.observations).flatten(0, 1)
        tr = getattr(self, "_critic_fwd", None)
        if tr is None or (tr is not False and (tr.inputs[0].data_ptr() != cobs.data_ptr() or tr.mb * len(getattr(self, "_critic_chunks", None) or [0]) != cobs.shape[0])):
            from .mlp_kernels import MlpTrainer, WideMlpTrainer
            tr = False
            self._critic_chunks = None
            if isinstance(ac.critic, torch.nn.Sequential) and cobs.is_cuda:
                tr = MlpTrainer([ac.critic], [cobs], cobs.shape[0], forward_only=True)
                if not tr.supported:                  # the 512-256-128 critics: the chain forward of the wide learner kernels
                    # in row chunks of a mini-batch: the wide workspace is sized for TRAINING on `mb` rows (activations, gradient slabs, dW
                    # partials: 0.8 GB for all 24 x 4096 rows, 6.4 GB at 32 768 envs) although a forward needs none of the gradient regions
                    rows_all = cobs.shape[0]
                    parts = next((c for c in (4, 3, 2) if rows_all % c == 0 and rows_all // c >= 4096), 1)
                    tr = WideMlpTrainer([ac.critic], [cobs], rows_all // parts, forward_only=True)
                    if tr.supported and parts > 1:
                        self._critic_chunks = [torch.arange(i * (rows_all // parts), (i + 1) * (rows_all // parts), device=cobs.device) for i in range(parts)]
                        self._critic_out = torch.empty(rows_all, 1, device=cobs.device)
            self._critic_fwd = tr
        if tr is not False and tr.supported:
            tr.refresh()                              # parameter addresses (stable; the values follow the optimiser)
            if getattr(self, "_critic_chunks", None):
                n = self._critic_chunks[0].numel()
                for i, rows in enumerate(self._critic_chunks):
                    self._critic_out[i * n:(i + 1) * n].copy_(tr.forward(rows)[0])
                return self._critic_out
            return tr.forward(None)[0]
        return ac.evaluate(cobs)

    def _rollout_steps_rolled(self, stats):
        """The whole rollout of an iteration as ONE launch (``lg_rollout_policy``: the multi-step kernel writes observations, actions,
        means, rewards and dones straight into the PPO storage) plus ONE bookkeeping launch (``lg_rollout_finish``: sigma, log-probs,
        time-out floats, episode statistics).  Returns None when the sim / actor pair has no multi-step kernel."""
        env, alg, fused = self.env, self.alg, self._fused
        st, T = alg.storage, self.num_steps_per_env
        N, A, dev = env.num_envs, st.actions.shape[-1], self.device
        roll = getattr(self, "_roll", None)
        if roll is None or roll["obs"].shape[0] != T + 1 or st.observations.data_ptr() != roll["obs"].data_ptr():
            with torch.inference_mode(False):         # (rollouts run under inference_mode; env.obs_buf becomes a view of this buffer and callers feed it to autograd modules)
                obs_all = torch.empty(T + 1, N, st.observations.shape[-1], device=dev)     # [T + 1]: the kernel leaves the next observations behind the stored ones
                st.observations = obs_all[:T]                                                # (the learner kernels read the storage through this view)
                roll = {"obs": obs_all, "actions": st.actions, "mean": st.mu, "rew": st.rewards.view(T, N), "dones": st.dones.view(T, N),
                        "time_outs": torch.zeros(T, N, dtype=torch.uint8, device=dev)}
            self._roll = roll
        try:
            env.rollout_policy(fused, T, storage=roll)
        except RuntimeError as exc:
            if "multi-step rollout kernel" not in str(exc):
                raise
            return None
        post = capi.lg_rollout_post()
        p = lambda x: x.data_ptr()
        post.actions, post.mean, post.rewards, post.dones, post.time_outs = p(st.actions), p(st.mu), p(st.rewards), p(st.dones), p(roll["time_outs"])
        post.std, post.sigma, post.log_prob, post.time_outs_f = p(alg.actor_critic.std), p(st.sigma), p(st.actions_log_prob), p(self._time_outs)
        post.cur_return, post.cur_length, post.sums = p(stats["cur_rew"]), p(stats["cur_len"]), p(stats["_sums"])
        post.steps, post.num_envs, post.num_actions = T, N, A
        rc = fused.lib.lg_rollout_finish(post, torch.cuda.current_stream(self.device).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"lg_rollout_finish failed ({rc}): {fused.lib.lg_last_error().decode()}")
        st.step = T
        st.values.copy_(self._critic_values(st).view(T, -1, 1))
        st.rewards.add_(alg.gamma * st.values * self._time_outs)
        obs = env.obs_buf
        return obs, obs

    def _rollout_steps_fused(self, stats):
        env, alg, fused = self.env, self.alg, self._fused
        st, T = alg.storage, self.num_steps_per_env
        if self._rolled is not False and "_sums" in stats and self.cfg.get("rolled_rollout", True):
            out = self._rollout_steps_rolled(stats)
            self._rolled = out is not None
            if out is not None:
                return out
        obs = env.get_observations()
        lib, step = fused.lib, capi.lg_rollout_step()
        p = lambda x: x.data_ptr()
        step.num_envs, step.num_obs, step.num_actions = env.num_envs, st.observations.shape[-1], st.actions.shape[-1]
        step.cur_return, step.cur_length, step.sums = p(stats["cur_rew"]), p(stats["cur_len"]), p(stats["_sums"])
        stream = torch.cuda.current_stream(self.device).cuda_stream
        ac = alg.actor_critic
        step.std = p(ac.std)                          # log-prob and sigma of the transition are stored by the same launch
        for t in range(T):
            prev_obs = obs
            actions, mean, (obs, _, rewards, dones, infos) = self._fused_env_step(obs)
            # storage writes + episode statistics of this transition: one launch (lg_rollout_record)
            touts = infos.get("time_outs")
            step.obs, step.actions, step.mean, step.rewards, step.dones = p(prev_obs), p(actions), p(mean), p(rewards), p(dones)
            step.time_outs = p(touts) if touts is not None else None
            step.storage_obs, step.storage_actions, step.storage_mu = p(st.observations[t]), p(st.actions[t]), p(st.mu[t])
            step.storage_rewards, step.storage_dones, step.storage_time_outs = p(st.rewards[t]), p(st.dones[t]), p(self._time_outs[t])
            step.storage_sigma, step.storage_log_prob = p(st.sigma[t]), p(st.actions_log_prob[t])
            assert dones.element_size() == 1 and (touts is None or touts.element_size() == 1) and rewards.is_contiguous()   # bool / uint8 flags
            rc = lib.lg_rollout_record(step, stream)
            if rc != 0:
                raise RuntimeError(f"lg_rollout_record failed ({rc}): {lib.lg_last_error().decode()}")
        st.step = T
        st.values.copy_(self._critic_values(st).view(T, -1, 1))                                # critic once on all transitions
        st.rewards.add_(alg.gamma * st.values * self._time_outs)                               # bootstrap on time-outs (PPO.process_env_step)
        return obs, obs

    # ------------------------------------------------------------------ graphed rollout
    def _rollout_steps(self, stats):
        """num_steps_per_env x (act -> env.step -> store); episode statistics as tensor ops (no host sync)."""
        if self._fused is not None:
            return self._rollout_steps_fused(stats)
        env, alg = self.env, self.alg
        obs = env.get_observations()
        pobs = env.get_privileged_observations()
        cobs = pobs if pobs is not None else obs
        for _ in range(self.num_steps_per_env):
            actions = alg.act(obs, cobs)
            obs, pobs, rewards, dones, infos = env.step(actions)
            cobs = pobs if pobs is not None else obs
            alg.process_env_step(rewards, dones, infos)
            d = dones.float()
            stats["cur_rew"] += rewards
            stats["cur_len"] += 1.0
            stats["sum_rew"] += (stats["cur_rew"] * d).sum()
            stats["sum_len"] += (stats["cur_len"] * d).sum()
            stats["count"] += d.sum()
            stats["cur_rew"] *= 1.0 - d
            stats["cur_len"] *= 1.0 - d
        return obs, cobs

    def _try_build_graphed_rollout(self):
        """Capture the whole rollout of one PPO iteration into a single HIP graph (one hipGraphLaunch per iteration
        instead of ~24 x 30 eager launches).  Needs this package's env (device step counter, ping-pong obs buffers)
        and an even number of steps so the observation buffers line up between replays."""
        env = self.env
        ok = (str(self.device).startswith("cuda") and hasattr(env, "begin_graph_capture") and self.num_steps_per_env % 2 == 0
              and self.cfg.get("graphed_rollout", True))
        if not ok:
            return None
        flip0, counter0 = getattr(env, "_obs_flip", 0), env.common_step_counter
        captured = False
        try:
            N, dev = env.num_envs, self.device
            sums = torch.zeros(3, device=dev)             # {sum of finished-episode returns, of their lengths, their count}
            stats = {"cur_rew": torch.zeros(N, device=dev), "cur_len": torch.zeros(N, device=dev),
                     "sum_rew": sums[0], "sum_len": sums[1], "count": sums[2], "_sums": sums}
            with torch.inference_mode():
                self._rollout_steps(stats)              # one eager warm-up rollout (allocators, lazy init); discarded
                self.alg.storage.clear()
                for v in stats.values():
                    v.zero_()
                torch.cuda.synchronize()
                flip0, counter0 = env._obs_flip, env.common_step_counter
                env.begin_graph_capture()
                captured = True
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    obs, cobs = self._rollout_steps(stats)
                    if hasattr(env, "capture_extras_flush"):
                        env.capture_extras_flush()          # extras["episode"] of the rollout's last step (its steps defer them)
                env.end_graph_capture(self.num_steps_per_env)
                captured = False
                assert env._obs_flip == flip0 and env.common_step_counter == counter0
            self.alg.storage.clear()
            return graph, stats, obs, cobs
        except Exception as exc:                          # fall back to eager launches
            print(f"[runner] graphed rollout unavailable ({type(exc).__name__}: {exc}); using eager steps")
            if hasattr(env, "_capturing"):
                env._capturing = False
                env._sim.set_deferred_extras(False)
            if captured:
                # steps issued during a capture that failed part-way were counted but never executed: put the observation
                # ping-pong and the step counter (push / resample / RNG phase) back where they were
                env._obs_flip = flip0
                env.obs_buf = env._obs_pair[flip0]
                env._sim.set_obs_output(env.obs_buf)
                env.common_step_counter = counter0
            self.alg.storage.clear()
            return None

    def learn(self, num_learning_iterations, init_at_random_ep_len=False):
        if init_at_random_ep_len:
            self.env.episode_length_buf[:] = torch.randint_like(self.env.episode_length_buf, high=int(self.env.max_episode_length))
        obs = self.env.get_observations()
        pobs = self.env.get_privileged_observations()
        cobs = pobs if pobs is not None else obs
        obs, cobs = obs.to(self.device), cobs.to(self.device)
        self.alg.actor_critic.train()
        rewbuffer, lenbuffer = deque(maxlen=100), deque(maxlen=100)
        cur_rew = torch.zeros(self.env.num_envs, dtype=torch.float, device=self.device)
        cur_len = torch.zeros(self.env.num_envs, dtype=torch.float, device=self.device)
        graphed = self._try_build_graphed_rollout()
        last = self.current_learning_iteration + num_learning_iterations
        for it in range(self.current_learning_iteration, last):
            t0 = time.time()
            if graphed is not None:
                graph, stats, obs, cobs = graphed
                with torch.inference_mode():
                    graph.replay()
                    self.env.common_step_counter += self.num_steps_per_env
                    self.alg.storage.step = self.num_steps_per_env
                    if self.log_dir is not None:
                        s_rew, s_len, cnt = (float(v) for v in torch.stack((stats["sum_rew"], stats["sum_len"], stats["count"])).cpu())
                        if cnt > 0:
                            rewbuffer.append(s_rew / cnt); lenbuffer.append(s_len / cnt)
                        for k in ("sum_rew", "sum_len", "count"):
                            stats[k].zero_()
                    t1 = time.time()
                    self.alg.compute_returns(cobs)
            else:
                with torch.inference_mode():
                    for _ in range(self.num_steps_per_env):
                        actions = self.alg.act(obs, cobs)
                        obs, pobs, rewards, dones, infos = self.env.step(actions)
                        cobs = pobs if pobs is not None else obs
                        obs, cobs, rewards, dones = obs.to(self.device), cobs.to(self.device), rewards.to(self.device), dones.to(self.device)
                        self.alg.process_env_step(rewards, dones, infos)
                        if self.log_dir is not None:
                            cur_rew += rewards
                            cur_len += 1
                            ids = (dones > 0).nonzero(as_tuple=False)
                            rewbuffer.extend(cur_rew[ids][:, 0].cpu().numpy().tolist())
                            lenbuffer.extend(cur_len[ids][:, 0].cpu().numpy().tolist())
                            cur_rew[ids] = 0
                            cur_len[ids] = 0
                    t1 = time.time()
                    self.alg.compute_returns(cobs)
            mean_value_loss, mean_surrogate_loss = self.alg.update()
            if self._fused is not None:
                self._fused.sync_device()                     # the MFMA actor follows the optimiser (device-side repack)
            t2 = time.time()
            self.tot_timesteps += self.num_steps_per_env * self.env.num_envs
            self.tot_time += t2 - t0
            if self.log_dir is not None:
                fps = int(self.num_steps_per_env * self.env.num_envs / (t2 - t0))
                mr = statistics.mean(rewbuffer) if len(rewbuffer) else float("nan")
                ml = statistics.mean(lenbuffer) if len(lenbuffer) else float("nan")
                row = self._log_scalars(it, fps, t1 - t0, t2 - t1, mean_value_loss, mean_surrogate_loss, mr, ml)
                print(f"it {it}/{last}  steps/s {fps}  collect {t1 - t0:.3f}s  learn {t2 - t1:.3f}s  value_loss {mean_value_loss:.4f}  "
                      f"surrogate {mean_surrogate_loss:.4f}  std {row['Policy/mean_noise_std']:.3f}  "
                      f"mean_reward {mr:.3f}  mean_ep_len {ml:.1f}")
                if it % self.save_interval == 0:
                    self.save(os.path.join(self.log_dir, f"model_{it}.pt"))
        self.current_learning_iteration += num_learning_iterations
        if self.log_dir is not None:
            self.save(os.path.join(self.log_dir, f"model_{self.current_learning_iteration}.pt"))

    def _log_scalars(self, it, fps, t_collect, t_learn, value_loss, surrogate, mean_reward, mean_len):
        """Per-iteration scalars under rsl_rl's TensorBoard tag names ([EXTERNAL] OnPolicyRunner.log): always to
        ``<log_dir>/progress.csv`` (no dependency), and to a SummaryWriter when tensorboard is importable."""
        ep = (getattr(self.env, "extras", None) or {}).get("episode") or {}
        keys = sorted(ep)
        dev_keys = [k for k in keys if torch.is_tensor(ep[k])]
        # one device -> host transfer for everything that lives on the device (mean action std + the episode means)
        dev_vals = torch.stack([self.alg.actor_critic.std.detach().mean().float()] + [ep[k].detach().float().reshape(()) for k in dev_keys]).cpu().tolist()
        row = {"iteration": it, "Loss/value_function": value_loss, "Loss/surrogate": surrogate, "Loss/learning_rate": self.alg.learning_rate,
               "Policy/mean_noise_std": dev_vals[0], "Perf/total_fps": fps, "Perf/collection_time": t_collect,
               "Perf/learning_time": t_learn, "Train/mean_reward": mean_reward, "Train/mean_episode_length": mean_len,
               "total_timesteps": self.tot_timesteps}
        got = dict(zip(dev_keys, dev_vals[1:]))
        row.update({f"Episode/{k}": got[k] if k in got else float(ep[k]) for k in keys})
        if self.writer is None:
            os.makedirs(self.log_dir, exist_ok=True)
            self._csv = open(os.path.join(self.log_dir, "progress.csv"), "a", buffering=1)
            self._csv_cols = list(row)
            if self._csv.tell() == 0:
                self._csv.write(",".join(self._csv_cols) + "\n")
            try:
                from torch.utils.tensorboard import SummaryWriter
                self.writer = SummaryWriter(log_dir=self.log_dir, flush_secs=10)
            except Exception:
                self.writer = False                              # tensorboard not installed: CSV only
        self._csv.write(",".join(repr(row.get(c, "")) if not isinstance(row.get(c, ""), str) else row[c] for c in self._csv_cols) + "\n")
        if self.writer:
            for k, v in row.items():
                if k != "iteration":
                    self.writer.add_scalar(k, v, it)
        return row

    def save(self, path, infos=None):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save({"model_state_dict": self.alg.actor_critic.state_dict(),
                    "optimizer_state_dict": self.alg.optimizer.state_dict(),
                    "iter": self.current_learning_iteration, "infos": infos}, path)

    def load(self, path, load_optimizer=True):
        d = torch.load(path, map_location=self.device, weights_only=True)
        self.alg.actor_critic.load_state_dict(d["model_state_dict"])
        if load_optimizer:
            self.alg.optimizer.load_state_dict(d["optimizer_state_dict"])
            if hasattr(self.alg, "after_optimizer_load"):
                self.alg.after_optimizer_load()               # re-link the device learning rate, drop the captured update graph
        if getattr(self, "_fused", None) is not None:
            self._fused.sync_device()
        self.current_learning_iteration = d["iter"]
        return d["infos"]

    def get_inference_policy(self, device=None):
        self.alg.actor_critic.eval()
        if device is not None:
            self.alg.actor_critic.to(device)
        return self.alg.actor_critic.act_inference
