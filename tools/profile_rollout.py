#!/usr/bin/env python3
"""Per-workgroup, per-step wall-clock stamps of the multi-step rollout kernel (lg_rollout_policy) in the LIGHT profiling build:
is a workgroup's step time random from step to step (then walking through the steps alone averages the tail away) or persistent
(then it does not)?   python tools/profile_sections.py build light   (here), then on the GPU box:   python tools/profile_rollout.py [T]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LG_HIP_LIB"] = os.path.join(ROOT, "legged_games_gym_amd", "csrc", "liblegged_hip_prof_light.so")
import numpy as np
import torch
from legged_games_gym_amd import capi
from legged_games_gym_amd.envs import task_registry
from legged_games_gym_amd.utils import get_args
from legged_games_gym_amd.rl import ActorCritic, FusedActor
from legged_games_gym_amd.utils.helpers import class_to_dict
T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
args = get_args(["--task", "anymal_c_flat", "--num_envs", "4096", "--headless", "--sim_device", "cuda:0", "--rl_device", "cuda:0"])
env_cfg, _ = task_registry.get_cfgs("anymal_c_flat")
if os.environ.get("LG_NO_SC"):               # the same workload without self-collision (asset.self_collisions = 1)
    env_cfg.asset.self_collisions = 1
env, cfg = task_registry.make_env("anymal_c_flat", args, env_cfg=env_cfg)
print("self-collision:", env.self_collision_modelled)
env.set_fixed_commands(0.5, 0.0, 0.0)
_, tcfg = task_registry.get_cfgs("anymal_c_flat")
torch.manual_seed(1)
ac = ActorCritic(env.num_obs, env.num_obs, env.num_actions, **class_to_dict(tcfg.policy)).to("cuda")
fused = FusedActor(ac, "cuda:0", seed=11)
env.reset()
st = None
for _ in range(10):
    st = env.rollout_policy(fused, T, storage=st)
torch.cuda.synchronize()
lib = capi.load_library()
lib.lg_debug_profile_roll.argtypes, lib.lg_debug_profile_roll.restype = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)], ctypes.c_int
out = (ctypes.c_uint64 * (33 * 1024))()
assert lib.lg_debug_profile_roll(env._sim.sim.handle, out) == 0
S = np.array(list(out), dtype=np.float64).reshape(33, 1024)[:T + 1, :256] * 10.0 / 1e3        # us
S -= S[0].min()
d = np.diff(S, axis=0)                        # [T, 256] step durations per workgroup
print(f"rollout kernel, {T} steps, 256 workgroups: start skew max {S[0].max():.2f} us; launch ends at {S[-1].max():.1f} us = {S[-1].max() / T:.2f} us per step")
print(f"  step duration of a workgroup: mean {d.mean():.2f}  median {np.median(d):.2f}  p90 {np.quantile(d, 0.9):.2f}  p99 {np.quantile(d, 0.99):.2f}  max {d.max():.2f} us")
tot = d.sum(axis=0)
print(f"  per-workgroup total / T: min {tot.min() / T:.2f}  median {np.median(tot) / T:.2f}  p90 {np.quantile(tot, 0.9) / T:.2f}  max {tot.max() / T:.2f} us")
print(f"  per-step max over workgroups (what a launch per step would cost): mean {d.max(axis=1).mean():.2f} us;  per-step median over workgroups: mean {np.median(d, axis=1).mean():.2f} us")
wg_mean = d.mean(axis=0)
print(f"  variance split: between workgroups (persistent) std {wg_mean.std():.2f} us; within a workgroup step to step std {(d - wg_mean).std():.2f} us")
print("  slowest workgroups (blockIdx: mean step us):", [(int(b), round(float(wg_mean[b]), 2)) for b in np.argsort(wg_mean)[-8:]])
print("  XCD (blockIdx % 8) mean step us:", [round(float(wg_mean[np.arange(256) % 8 == x].mean()), 2) for x in range(8)])
