#!/usr/bin/env python3
"""Standing / thrashing diagnostics of a task's robot for several ground stiffnesses: base height, joint-speed rms (jitter), foot normal
force mean / relative std over time (chatter), share of policy steps with the foot loaded, horizontal drift of the base (slipping).
   python tools/contact_probe.py [task] [action_std]"""
import os, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
import torch
from legged_games_gym_amd.envs import task_registry
from legged_games_gym_amd.utils import get_args
from legged_games_gym_amd.utils.packing import EngineOptions
task = sys.argv[1] if len(sys.argv) > 1 else "a1"
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
for K in [float(x) for x in os.environ.get("LG_K", "1e6,1e5").split(",")]:
    args = get_args(["--task", task, "--headless", "--sim_device", "cuda:0", "--rl_device", "cuda:0", "--num_envs", "512"])
    env_cfg, _ = task_registry.get_cfgs(task)
    env_cfg.sim.engine = EngineOptions(contact_stiffness=K)
    env_cfg.terrain.mesh_type, env_cfg.terrain.curriculum = "plane", False
    env_cfg.domain_rand.push_robots = False
    env_cfg.noise.add_noise = False
    with contextlib.redirect_stdout(io.StringIO()):
        env, _ = task_registry.make_env(task, args, env_cfg=env_cfg)
        env.reset()
    env.set_fixed_commands(0.0, 0.0, 0.0)
    g = torch.Generator(device="cuda").manual_seed(0)
    feet = env.feet_indices
    fz, hs, qd, tq, xy0, resets = [], [], [], [], None, 0
    with torch.inference_mode():
        for t in range(250):
            a = torch.randn(env.num_envs, env.num_actions, device="cuda", generator=g) * sigma
            _, _, _, dones, _ = env.step(a)
            resets += int(dones.sum())
            if t >= 50:
                if xy0 is None:
                    xy0 = env.root_states[:, :2].clone(); alive = torch.ones(env.num_envs, dtype=torch.bool, device="cuda")
                alive &= ~dones
                fz.append(env.contact_forces[:, feet, 2].clone()); hs.append(env.root_states[:, 2].clone())
                qd.append(env.dof_vel.clone()); tq.append(env.torques.clone())
    fz = torch.stack(fz); hs = torch.stack(hs); qd = torch.stack(qd); tq = torch.stack(tq)
    drift = (env.root_states[:, :2] - xy0).norm(dim=1)[alive] / (200 * env.dt)
    m = fz.mean(0); sd = fz.std(0)
    print(f"[{task} K={K:g} action std {sigma:g}] base z {float(hs.mean()):.3f}  dof_vel rms {float(qd.square().mean().sqrt()):.3f} rad/s  torque rms {float(tq.square().mean().sqrt()):.2f} Nm  "
          f"foot Fz mean {float(m.mean()):.1f} N, std over time / mean {float((sd / m.clamp(min=1.0)).mean()):.2f}, loaded (>1 N) {float((fz > 1.0).float().mean()):.2f} of steps  "
          f"base drift {float(drift.mean()):.3f} m/s  resets {resets}  weight/4 {float(env._robot_mass if hasattr(env, '_robot_mass') else 0) * 9.81 / 4:.1f}", flush=True)
    env_cfg.sim.engine = None
    del env
