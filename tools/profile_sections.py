#!/usr/bin/env python3
"""Where does k_step spend its time?  Builds the HIP sources with -DLG_PROFILE (s_memtime section counters, lane 0 of
every workgroup), runs the bench workload for a few hundred steps and prints the per-section share.

  python tools/profile_sections.py build            # here (hipcc cross-compiles) -> csrc/liblegged_hip_prof.so
  python tools/profile_sections.py run [task] [N]   # on the GPU box
  ... build light / run [task] [N] light            # -DLG_PROFILE_LIGHT: only each workgroup's start / end clock (spread of the product kernel)
"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "legged_games_gym_amd", "csrc")
LIB = os.path.join(CSRC, "liblegged_hip_prof.so")
NAMES = ["prologue (tables, state loads)", "torques (actuator LSTM / PD)", "kinematics + body terms + contact setup", "inward ABA recursion (per pass)",
         "base: butterfly + 6x6 solve", "outward accelerations + contact evaluate", "integrate + force sums", "post-physics (rewards, reset, obs)", "extras finisher",
         "post: commands, heights, push, contact force export", "post: termination + reward terms", "post: reward sum + episode sums", "post: reset block", "post: observations"]

LIGHT = "light" in sys.argv          # -DLG_PROFILE_LIGHT: start / end of every workgroup only (the product kernel's own spread)
if LIGHT:
    LIB = os.path.join(CSRC, "liblegged_hip_prof_light.so")
if sys.argv[1:2] == ["build"]:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-fno-slp-vectorize",
           "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-mllvm", "-amdgpu-mfma-vgpr-form", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0", "-DLG_PROFILE"] + (["-DLG_PROFILE_LIGHT"] if LIGHT else []) + ["-o", LIB, os.path.join(CSRC, "lg_kernels.hip")]
    print(" ".join(cmd)); subprocess.run(cmd, check=True); sys.exit(0)

os.environ["LG_HIP_LIB"] = LIB
import torch
from legged_games_gym_amd import capi
from legged_games_gym_amd.envs import task_registry
from legged_games_gym_amd.utils import get_args
task = sys.argv[2] if len(sys.argv) > 2 else "anymal_c_flat"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
args = get_args(["--task", task, "--num_envs", str(n), "--headless", "--sim_device", "cuda:0", "--rl_device", "cuda:0"])
env_cfg, _ = task_registry.get_cfgs(task)
if "nosc" in sys.argv:                      # anymal_c_flat without self-collision (asset.self_collisions = 1)
    env_cfg.asset.self_collisions = 1
env, cfg = task_registry.make_env(task, args, env_cfg=env_cfg)
print("self-collision:", env.self_collision_modelled)
if "defer" in sys.argv:                     # lg_set_deferred_extras(1): no finisher in the launch (the previous step's is done by workgroup 0)
    env._sim.set_deferred_extras(True)
    print("deferred extras: on")
lib = capi.load_library()
lib.lg_debug_profile.argtypes, lib.lg_debug_profile.restype = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int], ctypes.c_int
out = (ctypes.c_uint64 * 20)()
g = torch.Generator(device="cuda").manual_seed(0)
for _ in range(50):
    env.step(torch.randn(env.num_envs, env.num_actions, device="cuda", generator=g))
handle = env._sim.sim.handle
assert lib.lg_debug_profile(handle, out, 1) == 0
steps = 300
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
acts = [torch.randn(env.num_envs, env.num_actions, device="cuda", generator=g) for _ in range(8)]
fused = None
if "policy" in sys.argv:                    # the bench's launch: actor fused into the step (lg_step_policy), random-init policy
    from legged_games_gym_amd.rl import ActorCritic, FusedActor
    from legged_games_gym_amd.utils.helpers import class_to_dict
    _, tcfg = task_registry.get_cfgs(task)
    torch.manual_seed(1)
    ac = ActorCritic(env.num_obs, env.num_obs, env.num_actions, **class_to_dict(tcfg.policy)).to("cuda")
    fused = FusedActor(ac, "cuda:0", seed=11)
    for _ in range(50):
        env.step_policy(fused)
    assert lib.lg_debug_profile(handle, out, 1) == 0
t0.record()
for i in range(steps):
    if fused is not None: env.step_policy(fused)
    else: env.step(acts[i % 8])
t1.record(); torch.cuda.synchronize()
assert lib.lg_debug_profile(handle, out, 0) == 0
v = [int(x) for x in out]
wgs = steps * ((env.num_envs * (2 if task == "cassie" else 4) + 63) // 64)
tot, wall = v[14] / wgs, v[15] / wgs * 10.0          # cycles per workgroup-step; ns (100 MHz wall clock)
print(f"{task} N={env.num_envs}: {t0.elapsed_time(t1) / steps * 1e3:.1f} us per env.step (instrumented build); per workgroup {tot:.0f} cycles = {wall / 1e3:.1f} us -> {tot / wall:.2f} GHz counter")
for i, name in enumerate([] if LIGHT else NAMES):
    c = v[i] / wgs
    print(f"  {name:45s} {c:9.0f} cyc  {c / tot * 100:5.1f} %  {c / tot * wall / 1e3:6.2f} us")
print(f"  slowest workgroup of any launch: {v[16]} cycles = {v[16] / (tot / wall) / 1e3:.1f} us (the kernel ends when it does)")
print(f"  {'(unattributed)':45s} {tot - sum(v[:14]) / wgs:9.0f} cyc")

# ---- the last launch, workgroup by workgroup: which sections differ between workgroups, and when each one started / ended
import numpy as np
nb = min(1024, (env.num_envs * (2 if task == "cassie" else 4) + 63) // 64)
lib.lg_debug_profile_blocks.argtypes, lib.lg_debug_profile_blocks.restype = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int], ctypes.c_int
blk = (ctypes.c_uint64 * (20 * nb))()
assert lib.lg_debug_profile_blocks(handle, blk, nb) == 0
B = np.array(list(blk), dtype=np.float64).reshape(nb, 20)
ghz = tot / wall
print(f"last launch, {nb} workgroups: total per workgroup min {B[:, 14].min() / ghz / 1e3:.1f}  median {np.median(B[:, 14]) / ghz / 1e3:.1f}  p90 {np.quantile(B[:, 14], 0.9) / ghz / 1e3:.1f}  max {B[:, 14].max() / ghz / 1e3:.1f} us")
start = (B[:, 16] - B[:, 16].min()) * 10.0 / 1e3; end = (B[:, 17] - B[:, 16].min()) * 10.0 / 1e3
print(f"  start skew: median {np.median(start):.2f} max {start.max():.2f} us;  last end {end.max():.1f} us;  median end {np.median(end):.1f} us")
if LIGHT:
    dur = end - start
    print(f"  own duration (wall clock): min {dur.min():.1f}  p10 {np.quantile(dur, 0.1):.1f}  median {np.median(dur):.1f}  p90 {np.quantile(dur, 0.9):.1f}  p99 {np.quantile(dur, 0.99):.1f}  max {dur.max():.1f} us")
    print(f"  end of workgroup: p10 {np.quantile(end, 0.1):.1f}  median {np.median(end):.1f}  p90 {np.quantile(end, 0.9):.1f}  p99 {np.quantile(end, 0.99):.1f}  last {end.max():.1f} us")
    print("  XCD (blockIdx % 8) mean end us:", [round(float(end[np.arange(nb) % 8 == x].mean()), 1) for x in range(8)])
    cnt = B[:, 19].astype(np.uint64); nrs = B[:, 18]
    basep, scp = ((cnt >> np.uint64(8)) & np.uint64(0xFF)).astype(int), ((cnt >> np.uint64(16)) & np.uint64(0xFF)).astype(int)
    calm = (nrs == 0) & (basep == 0) & (scp == 0)
    print(f"  workgroups without a reset, a trunk contact or a self-collision pass: {int(calm.sum())} of {nb}, own duration median {np.median(dur[calm]):.1f} max {dur[calm].max():.1f} us;"
          f"  the others: median {np.median(dur[~calm]) if (~calm).any() else 0:.1f} max {dur[~calm].max() if (~calm).any() else 0:.1f} us")
    print("  slowest workgroups (blockIdx: own us | resets, trunk-contact passes, self-collision passes):",
          [(int(b), round(float(dur[b]), 1), int(nrs[b]), int(basep[b]), int(scp[b])) for b in np.argsort(dur)[-10:]])
    sys.exit(0)
slow = np.argsort(B[:, 14])[-max(1, nb // 20):]
for i, name in enumerate(NAMES):
    print(f"  {name:45s} median {np.median(B[:, i]) / ghz / 1e3:6.2f}  p90 {np.quantile(B[:, i], 0.9) / ghz / 1e3:6.2f}  max {B[:, i].max() / ghz / 1e3:6.2f}  | slowest 5% of workgroups: {B[slow, i].mean() / ghz / 1e3:6.2f} us")
print("  slowest workgroups (blockIdx: total us, start us):", [(int(b), round(B[b, 14] / ghz / 1e3, 1), round(float(start[b]), 1)) for b in slow[-8:]])
nr = B[:, 18]
print(f"  resets in the launch: {int(nr.sum())} envs in {int((nr > 0).sum())} workgroups;  total us with resets {B[nr > 0, 14].mean() / ghz / 1e3 if (nr > 0).any() else 0:.1f}  without {B[nr == 0, 14].mean() / ghz / 1e3:.1f};  corr(total, resets) {np.corrcoef(B[:, 14], nr)[0, 1]:.2f}")
# rare-path census (LG_PROF_COUNT, slot 19): how often ANY lane of the rigid-body wave took a path, per workgroup, and what a taken path costs
cnt = B[:, 19].astype(np.uint64)
F = {"limb contact branches": (cnt & np.uint64(0xFF)), "base contact branch": ((cnt >> np.uint64(8)) & np.uint64(0xFF)), "self-collision passes": ((cnt >> np.uint64(16)) & np.uint64(0xFF)),
     "joint-limit branches": ((cnt >> np.uint64(24)) & np.uint64(0xFF)), "speed-limit branches": ((cnt >> np.uint64(32)) & np.uint64(0xFF)), "limb contact lanes": ((cnt >> np.uint64(48)) & np.uint64(0xFFFF))}
cols = [(name, np.asarray(f, dtype=np.float64)) for name, f in list(F.items()) + [("reset envs", nr)]]
var = [(name, f) for name, f in cols if f.max() > f.min()]            # a count that is the same in every workgroup is part of the intercept
X = np.stack([f for _, f in var] + [np.ones(nb)], axis=1)
coef, *_ = np.linalg.lstsq(X, B[:, 14] / ghz / 1e3, rcond=None)
cost = dict(zip([name for name, _ in var], coef[:-1]))
print("  rare-path census per workgroup (count: median / p90 / max; least-squares cost per event in us; mean contribution in us):")
for name, f in cols:
    if name in cost:
        c = cost[name]
        print(f"    {name:24s} {np.median(f):6.0f} / {np.quantile(f, 0.9):6.0f} / {f.max():6.0f}   {c:+8.3f} us each   {c * f.mean():+7.2f} us mean   {c * (f.max() - np.median(f)):+7.2f} us max - median")
    else:
        print(f"    {name:24s} {np.median(f):6.0f} / {np.quantile(f, 0.9):6.0f} / {f.max():6.0f}   the same in every workgroup (part of the intercept)")
res = B[:, 14] / ghz / 1e3 - X @ coef
print(f"    intercept {coef[-1]:.1f} us; residual std {res.std():.2f} us (total std {(B[:, 14] / ghz / 1e3).std():.2f})")
print("  XCD (blockIdx % 8) mean total us:", [round(B[np.arange(nb) % 8 == x, 14].mean() / ghz / 1e3, 1) for x in range(8)])
if (nr > 0).any():
    print("  per section, workgroups WITH a reset env vs without (mean us):")
    for i, name in enumerate(NAMES):
        a, b = B[nr > 0, i].mean() / ghz / 1e3, B[nr == 0, i].mean() / ghz / 1e3
        print(f"    {name:45s} {a:6.2f}  {b:6.2f}   {a - b:+6.2f}")
    print("  rows of the workgroups with resets (us per section 0..13 | total | resets):")
    for b in np.nonzero(nr > 0)[0][:24]:
        print("   ", int(b), " ".join(f"{B[b, i] / ghz / 1e3:5.1f}" for i in range(14)), "|", f"{B[b, 14] / ghz / 1e3:5.1f}", "|", int(nr[b]))
