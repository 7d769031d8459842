"""CLI entry point (reference ``legged_gym/scripts/train.py:39-47``):
``python -m legged_games_gym_amd.scripts.train --task=anymal_c_flat --headless``"""
from legged_games_gym_amd.envs import *  # noqa: F401,F403  (registers the tasks)
from legged_games_gym_amd.utils import get_args
from legged_games_gym_amd.utils.task_registry import task_registry


def train(args):
    env, env_cfg = task_registry.make_env(name=args.task, args=args)
    ppo_runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args)
    ppo_runner.learn(num_learning_iterations=train_cfg.runner.max_iterations, init_at_random_ep_len=True)


if __name__ == "__main__":
    train(get_args())
