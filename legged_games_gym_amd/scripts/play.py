"""Headless evaluation loop (the in-scope part of reference ``legged_gym/scripts/play.py:42-113``:
cfg overrides :45-51, resume + inference policy :55-59, the policy(obs) -> env.step loop :78-80).
``EXPORT_POLICY`` writes ``logs/<experiment>/exported/policies/policy_1.pt`` like :61-64; the state / reward logging of
:66-113 goes through ``utils/logger.py`` (plots are written to a PNG, headless).  Viewer and camera motion are out of scope."""
import os

import torch

from legged_games_gym_amd import LEGGED_GYM_ROOT_DIR
from legged_games_gym_amd.utils.helpers import export_policy_as_jit
from legged_games_gym_amd.utils.logger import Logger

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
    logger = Logger(env.dt, out_dir=os.path.join(LEGGED_GYM_ROOT_DIR, "logs", train_cfg.runner.experiment_name, "exported"))
    robot_index, joint_index = 0, 1                          # which robot / joint is logged (:67-68)
    stop_state_log = 100                                     # steps before the states are plotted
    stop_rew_log = int(env.max_episode_length) + 1           # steps before the average episode rewards are printed
    n = steps if steps is not None else 10 * int(env.max_episode_length)
    tot = torch.zeros(env.num_envs, device=env.device)
    for i in range(n):
        actions = policy(obs.detach())
        obs, _, rews, dones, infos = env.step(actions.detach())
        tot += rews
        if i < stop_state_log:
            logger.log_states({
                "dof_pos_target": actions[robot_index, joint_index].item() * env.cfg.control.action_scale,
                "dof_pos": env.dof_pos[robot_index, joint_index].item(),
                "dof_vel": env.dof_vel[robot_index, joint_index].item(),
                "dof_torque": env.torques[robot_index, joint_index].item(),
                "command_x": env.commands[robot_index, 0].item(), "command_y": env.commands[robot_index, 1].item(),
                "command_yaw": env.commands[robot_index, 2].item(),
                "base_vel_x": env.base_lin_vel[robot_index, 0].item(), "base_vel_y": env.base_lin_vel[robot_index, 1].item(),
                "base_vel_z": env.base_lin_vel[robot_index, 2].item(), "base_vel_yaw": env.base_ang_vel[robot_index, 2].item(),
                "contact_forces_z": env.contact_forces[robot_index, env.feet_indices, 2].cpu().numpy()})
        elif i == stop_state_log:
            print("state plots written to:", logger.plot_states())
        if 0 < i < stop_rew_log:
            if infos["episode"]:
                num_episodes = int(torch.sum(env.reset_buf).item())
                if num_episodes > 0:
                    logger.log_rewards(infos["episode"], num_episodes)
        elif i == stop_rew_log:
            logger.print_rewards()
    print(f"mean reward per step over {n} steps: {(tot / n).mean().item():.4f}")
    return env


EXPORT_POLICY = False

if __name__ == "__main__":
    EXPORT_POLICY = True
    play(get_args())
