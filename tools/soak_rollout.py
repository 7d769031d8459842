import sys; sys.path.insert(0,'/root/repo')
import torch, time
from legged_games_gym_amd.envs import task_registry
from legged_games_gym_amd.utils import get_args
from legged_games_gym_amd.rl import ActorCritic, FusedActor
from legged_games_gym_amd.utils.helpers import class_to_dict
args = get_args(["--task", "anymal_c_flat", "--num_envs", "4096", "--headless", "--sim_device", "cuda:0", "--rl_device", "cuda:0"])
env, cfg = task_registry.make_env("anymal_c_flat", args)
_, tcfg = task_registry.get_cfgs("anymal_c_flat")
torch.manual_seed(1)
ac = ActorCritic(env.num_obs, env.num_obs, env.num_actions, **class_to_dict(tcfg.policy)).to("cuda")
fused = FusedActor(ac, "cuda:0", seed=11)
env.reset()
with torch.inference_mode():
    replay, st = env.make_graphed_rollout(fused, 20)
    t0=time.time(); resets=0
    for i in range(1000):
        replay()
        if i % 100 == 99:
            torch.cuda.synchronize()
            r=env.root_states; q=r[:,3:7].norm(dim=1)
            assert torch.isfinite(st["obs"]).all() and torch.isfinite(r).all() and torch.isfinite(env.dof_vel).all()
            assert (q-1).abs().max() < 1e-3 and env.dof_vel.abs().max() <= 20.001
            resets += int(st["dones"].sum())
    torch.cuda.synchronize()
print(f"soak: 20000 policy steps x 4096 envs through lg_rollout_policy in {time.time()-t0:.2f} s; state finite, unit quaternions, |dof_vel| <= 20; resets in the sampled segments {resets}; device status {env._sim.sim.device_status(True)}; episode means {[round(float(v),4) for v in env._sim.buf['episode_means'][:4]]}")
