#!/usr/bin/env python3
"""Statistics of the first PPO rollouts/updates of a task (debug aid): rewards, values, returns, advantages, losses."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
import torch
from legged_games_gym_amd.envs import task_registry
from legged_games_gym_amd.utils import get_args
task = sys.argv[1] if len(sys.argv) > 1 else "a1"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
args = get_args(["--task", task, "--headless", "--sim_device", "cuda:0", "--rl_device", "cuda:0"])
env, env_cfg = task_registry.make_env(task, args)
runner, train_cfg = task_registry.make_alg_runner(env, task, args, log_root=None)
alg, st = runner.alg, runner.alg.storage
env.episode_length_buf[:] = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
q = lambda x: [round(float(v), 4) for v in torch.quantile(x.flatten().float()[:4_000_000], torch.tensor([0.0, 0.01, 0.5, 0.99, 1.0], device=x.device))]
for it in range(iters):
    stats = {k: torch.zeros(env.num_envs if k.startswith("cur") else (), device="cuda") for k in ("cur_rew", "cur_len", "sum_rew", "sum_len", "count")}
    with torch.inference_mode():
        obs, cobs = runner._rollout_steps(stats)
        alg.compute_returns(cobs)
    print(f"--- it {it}: obs|max| {float(st.observations.abs().max()):.2f}  rewards q {q(st.rewards)}  values q {q(st.values)}  returns q {q(st.returns)}  adv q {q(st.advantages)}  dones {float(st.dones.float().mean()):.5f}")
    print("    actions |max|", float(st.actions.abs().max()), " mu |max|", float(st.mu.abs().max()), " sigma mean", float(st.sigma.mean()))
    eps = (st.actions - st.mu) / st.sigma
    g = (st.advantages * (eps.square() - 1.0)).mean(dim=(0, 1))          # d(surrogate gain)/d(log sigma_i); entropy adds +entropy_coef
    print("    E[A (eps^2-1)] per dof:", [round(float(v), 4) for v in g], " mean", round(float(g.mean()), 4), " vs entropy_coef", alg.entropy_coef)
    vl, sl = alg.update()
    print(f"    value_loss {vl:.4f} surrogate {sl:.4f} lr {alg.learning_rate:.2e} std {float(alg.actor_critic.std.mean()):.4f}")
