import ctypes, os, sys
ROOT="/root/repo"; sys.path.insert(0, ROOT)
os.environ["LG_HIP_LIB"] = os.path.join(ROOT, "legged_games_gym_amd", "csrc", "liblegged_hip_prof.so")
import numpy as np, torch
from legged_games_gym_amd import capi
from legged_games_gym_amd.envs import task_registry
from legged_games_gym_amd.utils import get_args
from legged_games_gym_amd.rl import ActorCritic, FusedActor
from legged_games_gym_amd.utils.helpers import class_to_dict
sys.path.insert(0, os.path.join(ROOT,"tools"))
NAMES = ["prologue (tables, state loads)", "torques (actuator LSTM / PD)", "kinematics + body terms + contact setup", "inward ABA recursion (per pass)",
         "base: butterfly + 6x6 solve", "outward accelerations + contact evaluate", "integrate + force sums", "post-physics (rewards, reset, obs)", "extras finisher",
         "post: commands, heights, push, contact force export", "post: termination + reward terms", "post: reward sum + episode sums", "post: reset block", "post: observations"]
T=20
args = get_args(["--task", "anymal_c_flat", "--num_envs", "4096", "--headless", "--sim_device", "cuda:0", "--rl_device", "cuda:0"])
env, cfg = task_registry.make_env("anymal_c_flat", args)
env.set_fixed_commands(0.5, 0.0, 0.0)
_, tcfg = task_registry.get_cfgs("anymal_c_flat")
torch.manual_seed(1)
ac = ActorCritic(env.num_obs, env.num_obs, env.num_actions, **class_to_dict(tcfg.policy)).to("cuda")
fused = FusedActor(ac, "cuda:0", seed=11)
env.reset()
lib = capi.load_library()
lib.lg_debug_profile.argtypes, lib.lg_debug_profile.restype = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int], ctypes.c_int
out = (ctypes.c_uint64 * 20)()
st=None
for _ in range(5): st = env.rollout_policy(fused, T, storage=st)
h = env._sim.sim.handle
assert lib.lg_debug_profile(h, out, 1) == 0
R=10
for _ in range(R): st = env.rollout_policy(fused, T, storage=st)
assert lib.lg_debug_profile(h, out, 0) == 0
v=[int(x) for x in out]
wgsteps = R*T*256
tot, wall = v[14]/wgsteps, v[15]/wgsteps*10.0
print(f"rollout kernel (instrumented): per workgroup-step {tot:.0f} cycles = {wall/1e3:.2f} us")
acc=0
for i,n in enumerate(NAMES):
    c=v[i]/wgsteps; acc+=c
    print(f"  {n:50s} {c:8.0f} cyc {c/tot*wall/1e3:6.2f} us")
print(f"  {'(unattributed: actor, step boundary, ...)':50s} {tot-acc:8.0f} cyc {(tot-acc)/tot*wall/1e3:6.2f} us")
