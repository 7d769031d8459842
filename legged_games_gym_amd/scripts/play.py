"""Headless evaluation loop (the in-scope part of reference ``legged_gym/scripts/play.py:42-113``:
cfg overrides :45-51, resume + inference policy :55-59, the policy(obs) -> env.step loop :78-80).
``EXPORT_POLICY`` writes ``logs/<experiment>/exported/policies/policy_1.pt`` like :61-64.  Viewer, camera motion and the
matplotlib Logger are out of scope."""
import os

import torch

from legged_games_gym_amd import LEGGED_GYM_ROOT_DIR
from legged_games_gym_amd.utils.helpers import export_policy_as_jit

from legged_games_gym_amd.envs import *  # noqa: F401,F403
from legged_games_gym_amd.utils import get_args
from legged_games_gym_amd.utils.task_registry import task_registry


def play(args, steps=None):
    env_cfg, train_cfg = task_registry.get_cfgs(name=args.task)
    env_cfg.env.num_envs = min(env_cfg.env.num_envs, 50)
    env_cfg.terrain.num_rows = 5
    env_cfg.terrain.num_cols = 5
    env_cfg.terrain.curriculum = False
    env_cfg.noise.add_noise = False
    env_cfg.domain_rand.randomize_friction = False
    env_cfg.domain_rand.push_robots = False
    env, _ = task_registry.make_env(name=args.task, args=args, env_cfg=env_cfg)
    obs = env.get_observations()
    train_cfg.runner.resume = True
    ppo_runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args, train_cfg=train_cfg)
    policy = ppo_runner.get_inference_policy(device=env.device)
    if EXPORT_POLICY:
        path = os.path.join(LEGGED_GYM_ROOT_DIR, "logs", train_cfg.runner.experiment_name, "exported", "policies")
        print("Exported policy as jit script to:", export_policy_as_jit(ppo_runner.alg.actor_critic, path))
    n = steps if steps is not None else 10 * int(env.max_episode_length)
    tot = torch.zeros(env.num_envs, device=env.device)
    for _ in range(n):
        actions = policy(obs.detach())
        obs, _, rews, dones, infos = env.step(actions.detach())
        tot += rews
    print(f"mean reward per step over {n} steps: {(tot / n).mean().item():.4f}")
    return env


EXPORT_POLICY = False

if __name__ == "__main__":
    EXPORT_POLICY = True
    play(get_args())
