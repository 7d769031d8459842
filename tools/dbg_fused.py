import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from test_gpu_env_surface import _env
from legged_games_gym_amd.rl import ActorCritic, FusedActor
from legged_games_gym_amd.utils.helpers import class_to_dict
from legged_games_gym_amd.envs import task_registry
outs = []
for fused_step in (False, True):
    env, cfg = _env("anymal_c_flat", 200)
    _, train_cfg = task_registry.get_cfgs("anymal_c_flat")
    torch.manual_seed(3)
    ac = ActorCritic(env.num_obs, env.num_obs, env.num_actions, **class_to_dict(train_cfg.policy)).to("cuda")
    actor = FusedActor(ac, "cuda:0", seed=11)
    obs, _ = env.reset()
    rec = []
    for t in range(6):
        if fused_step:
            (act, mean), (obs, _, rew, dones, _) = env.step_policy(actor)
            act, mean = act.clone(), mean.clone()
        else:
            actor._host_step = env.common_step_counter
            act, mean = (x.clone() for x in actor.act_with_mean(obs))
            obs, _, rew, dones, _ = env.step(act)
        rec.append((act, mean, obs.clone(), rew.clone(), dones.clone(), env.dof_pos.clone()))
    outs.append(rec)
names = ["act", "mean", "obs", "rew", "dones", "dof_pos"]
for t, (a, b) in enumerate(zip(*outs)):
    for n, x, y in zip(names, a, b):
        d = (x.float() - y.float()).abs()
        if float(d.max()) != 0.0:
            idx = torch.nonzero(d > 0)
            print("step", t, n, "maxdiff", float(d.max()), "count", idx.shape[0], "first", idx[:6].tolist())
