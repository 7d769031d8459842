#!/usr/bin/env python3
"""Per-step reward / observation statistics of a task under N(0, std) actions (debug aid for PPO ignition)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
import torch
from legged_games_gym_amd.envs import task_registry
from legged_games_gym_amd.utils import get_args
task = sys.argv[1] if len(sys.argv) > 1 else "a1"
std = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
args = get_args(["--task", task, "--headless", "--sim_device", "cuda:0", "--rl_device", "cuda:0"])
env, cfg = task_registry.make_env(task, args)
obs = env.get_observations()
g = torch.Generator(device="cuda").manual_seed(0)
tot, pos, mx, omax, resets, touts = 0.0, 0.0, 0.0, 0.0, 0, 0
hist = torch.zeros(8)
for t in range(steps):
    a = torch.randn(env.num_envs, env.num_actions, device="cuda", generator=g) * std
    obs, _, rew, dones, infos = env.step(a)
    tot += float(rew.mean()); pos += float((rew > 0).float().mean()); mx = max(mx, float(rew.max())); omax = max(omax, float(obs.abs().max()))
    resets += int(dones.sum()); touts += int(infos["time_outs"].sum()) if "time_outs" in infos else 0
    if float(rew.max()) > 0.05:
        i = int(rew.argmax()); print("spike", t, i, float(rew[i]), "done", bool(dones[i]))
print(f"{task} std {std}: mean rew/step {tot/steps:.5f}  frac>0 {pos/steps:.3f}  max rew {mx:.4f}  max|obs| {omax:.2f}  resets {resets} (time-outs {touts}) of {steps*env.num_envs}")
print("episode terms:", {k: round(float(v), 4) for k, v in infos["episode"].items()})
print("root z-rel min/mean", float((env.root_states[:, 2]).min()), float(env.root_states[:, 2].mean()), " |dof_vel| max", float(env.dof_vel.abs().max()))
