#!/usr/bin/env python3
"""End-to-end check: PPO (bundled runner) on anymal_c_flat for a few hundred iterations; prints the reward trend."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
import torch
from legged_games_gym_amd.envs import task_registry
from legged_games_gym_amd.utils import get_args
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
task = sys.argv[2] if len(sys.argv) > 2 else "anymal_c_flat"
args = get_args(["--task", task, "--headless", "--sim_device", "cuda:0", "--rl_device", "cuda:0", "--max_iterations", str(iters)])
env_cfg, train_cfg0 = task_registry.get_cfgs(task)
if os.environ.get("LG_SEED"):            # the env seeds from the registered train cfg (reference quirk), not from --seed
    train_cfg0.seed = int(os.environ["LG_SEED"])
if os.environ.get("LG_ENTROPY"):
    train_cfg0.algorithm.entropy_coef = float(os.environ["LG_ENTROPY"])
if os.environ.get("LG_PLANE"):
    env_cfg.terrain.mesh_type = "plane"; env_cfg.terrain.curriculum = False; env_cfg.terrain.measure_heights = bool(int(os.environ.get("LG_HEIGHTS", "1")))
if os.environ.get("LG_NO_PUSH"):
    env_cfg.domain_rand.push_robots = False
if os.environ.get("LG_NO_NOISE"):
    env_cfg.noise.add_noise = False
if os.environ.get("LG_NO_CURRICULUM"):
    env_cfg.terrain.curriculum = False
if os.environ.get("LG_KD"):
    env_cfg.control.damping = {k: float(os.environ["LG_KD"]) for k in env_cfg.control.damping}
if os.environ.get("LG_UNCLIPPED"):
    env_cfg.rewards.only_positive_rewards = False
if os.environ.get("LG_INIT_STD"):
    train_cfg0.policy.init_noise_std = float(os.environ["LG_INIT_STD"])
env, env_cfg = task_registry.make_env(task, args, env_cfg=env_cfg)
runner, train_cfg = task_registry.make_alg_runner(env, task, args, log_root=os.environ.get("LG_LOG_ROOT", "/tmp/lg_train_logs"))
t0 = time.time()
runner.learn(num_learning_iterations=iters, init_at_random_ep_len=True)
print(f"total {time.time()-t0:.1f}s for {iters} iterations x 24 x {env.num_envs} = {iters*24*env.num_envs/1e6:.1f}M env-steps")
# evaluate: mean tracking / upright after training
obs = env.get_observations()
policy = runner.get_inference_policy(device=env.device)
env.set_fixed_commands(0.5, 0.0, 0.0)
with torch.inference_mode():
    for _ in range(300):
        obs, _, rew, dones, infos = env.step(policy(obs))
print("episode terms:", {k: round(float(v), 4) for k, v in infos["episode"].items()})
print("eval: mean base vx %.3f (command 0.5), upright frac %.3f, mean rew/step %.4f, resets/step %.4f" % (
    float(env.base_lin_vel[:, 0].mean()), float((env.projected_gravity[:, 2] < -0.9).float().mean()), float(rew.mean()), float(dones.float().mean())))
