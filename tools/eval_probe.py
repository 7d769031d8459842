import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
import torch
from legged_games_gym_amd.envs import task_registry
from legged_games_gym_amd.utils import get_args
task = sys.argv[1]; iters = int(sys.argv[2])
args = get_args(["--task", task, "--headless", "--sim_device", "cuda:0", "--rl_device", "cuda:0", "--num_envs", "1024"])
env, env_cfg = task_registry.make_env(task, args)
runner, train_cfg = task_registry.make_alg_runner(env, task, args, log_root=None)
runner.learn(num_learning_iterations=iters, init_at_random_ep_len=True)
policy = runner.get_inference_policy(device=env.device)
obs = env.get_observations()
print("before fix: cmd mean", env.commands.mean(0).tolist(), "blv", env.base_lin_vel.mean(0).tolist())
with torch.inference_mode():
    for i in range(200):
        obs, _, rew, dones, infos = env.step(policy(obs))
print("training cmds: |cmd_xy| mean", float(env.commands[:, :2].norm(dim=1).mean()), "tracking err", float((env.commands[:, :2] - env.base_lin_vel[:, :2]).norm(dim=1).mean()))
env.set_fixed_commands(0.5, 0.0, 0.0)
with torch.inference_mode():
    for i in range(300):
        obs, _, rew, dones, infos = env.step(policy(obs))
        if i % 100 == 99:
            print(i, "cmd", env.commands.mean(0).tolist(), "blv", env.base_lin_vel.mean(0).tolist(), "obs cmd", obs[:, 9:12].mean(0).tolist(), "up", float((env.projected_gravity[:, 2] < -0.9).float().mean()))
