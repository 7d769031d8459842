#!/usr/bin/env python3
"""Long-run stability soak: N envs x S policy steps with a random-init policy (graph replay); checks state stays finite,
quaternions unit, velocities bounded; prints episode statistics."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
import torch
from legged_games_gym_amd.envs import task_registry
from legged_games_gym_amd.utils import get_args
from legged_games_gym_amd.rl import ActorCritic
from legged_games_gym_amd.utils.helpers import class_to_dict
task = sys.argv[1] if len(sys.argv) > 1 else "anymal_c_flat"; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
args = get_args(["--task", task, "--headless", "--sim_device", "cuda:0", "--rl_device", "cuda:0"])
env, cfg = task_registry.make_env(task, args)
_, tcfg = task_registry.get_cfgs(task)
torch.manual_seed(1)
pol = ActorCritic(env.num_obs, env.num_obs, env.num_actions, **class_to_dict(tcfg.policy)).cuda()
env.reset()
with torch.inference_mode():
    step = env.make_graphed_step(pol.act)
    t0 = time.time(); resets = torch.zeros((), device="cuda"); tout = torch.zeros((), device="cuda")
    for i in range(steps):
        step()
        resets += env.reset_buf.sum(); tout += env.time_out_buf.sum()
        if (i + 1) % 5000 == 0:
            r = env.root_states
            ok = bool(torch.isfinite(r).all()) and bool(torch.isfinite(env.obs_buf).all()) and bool(torch.isfinite(env.dof_state).all())
            qn = r[:, 3:7].norm(dim=1)
            print(f"step {i+1}: finite={ok} |q| in [{float(qn.min()):.6f},{float(qn.max()):.6f}] max|v|={float(r[:,7:10].abs().max()):.2f} max|w|={float(r[:,10:13].abs().max()):.2f} "
                  f"max|qd|={float(env.dof_vel.abs().max()):.2f} z in [{float(r[:,2].min()):.3f},{float(r[:,2].max()):.3f}] resets={int(resets)} timeouts={int(tout)} "
                  f"max|F|={float(env.contact_forces.abs().max()):.0f} elapsed={time.time()-t0:.1f}s", flush=True)
            assert ok
print("soak ok:", task, env.num_envs, "envs x", steps, "steps")
