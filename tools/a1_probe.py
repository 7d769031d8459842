#!/usr/bin/env python3
"""A1 and the contact impedances (round-3 experiment): does scaling the engine's contact stiffness / damping with the robot's mass
(same damping ratio as ANYmal-C) change (a) how often A1 falls under N(0,1) actions and (b) whether PPO with the reference's defaults
ignites?     python tools/a1_probe.py [iters] [task]      (GPU box)"""
import os, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
import torch
from legged_games_gym_amd.envs import task_registry
from legged_games_gym_amd.utils import get_args
from legged_games_gym_amd.utils.packing import EngineOptions
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
task = sys.argv[2] if len(sys.argv) > 2 else "a1"
variants = {"default": None}
for s in [float(x) for x in os.environ.get("LG_SCALES", "0.239").split(",") if x]:
    variants[f"impedance x{s:g}"] = EngineOptions(contact_stiffness=1.0e6 * s, contact_damping=2.0e4 * s, friction_damping=2.5e4 * s)
for spec in [x for x in os.environ.get("LG_TRIPLES", "").split(";") if x]:        # "K,damping,friction" scale triples
    a, b, c = (float(v) for v in spec.split(","))
    variants[f"K x{a:g}, normal damping x{b:g}, friction damping x{c:g}"] = EngineOptions(contact_stiffness=1.0e6 * a, contact_damping=2.0e4 * b, friction_damping=2.5e4 * c)
if os.environ.get("LG_NO_DEFAULT"):
    variants.pop("default")
if os.environ.get("LG_ONLY"):
    variants = {k: v for k, v in variants.items() if os.environ["LG_ONLY"] in k}
for name, eng in variants.items():
    args = get_args(["--task", task, "--headless", "--sim_device", "cuda:0", "--rl_device", "cuda:0", "--max_iterations", str(iters)])
    env_cfg, train_cfg0 = task_registry.get_cfgs(task)
    env_cfg.sim.engine = eng
    with contextlib.redirect_stdout(io.StringIO()):
        env, env_cfg = task_registry.make_env(task, args, env_cfg=env_cfg)
        env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    resets, steps = 0, 300
    with torch.inference_mode():
        for _ in range(steps):
            _, _, _, dones, _ = env.step(torch.randn(env.num_envs, env.num_actions, device="cuda", generator=g))
            resets += int(dones.sum())
    print(f"[{name}] {task}: resets under N(0,1) actions: {resets} in {steps * env.num_envs} env-steps = 1 per {steps * env.num_envs / max(resets, 1):.0f};"
          f" base height {float(env.root_states[:, 2].mean() - env.env_origins[:, 2].mean()):.3f}", flush=True)
    if iters > 0:
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            runner, train_cfg = task_registry.make_alg_runner(env, task, args, log_root="/tmp/lg_a1_logs")
            runner.learn(num_learning_iterations=iters, init_at_random_ep_len=True)
        lines = [l for l in buf.getvalue().splitlines() if "mean_reward" in l]
        for l in lines[:: max(1, len(lines) // 8)] + lines[-1:]:
            print("   ", l.strip())
        print(f"    final action std {float(runner.alg.actor_critic.std.mean()):.3f}", flush=True)
    env_cfg.sim.engine = None
    del env
