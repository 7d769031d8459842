"""CLI entry point (reference ``legged_gym/scripts/train.py:39-47``):
``python -m legged_games_gym_amd.scripts.train --task=anymal_c_flat --headless``

Multi-GPU (new, the reference is single-process): launch one process per GPU,
``python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m legged_games_gym_amd.scripts.train
--task=anymal_c_rough --headless``.  Every rank owns its own ``num_envs`` environments on ``cuda:LOCAL_RANK`` (env seed
= registered seed + rank), the policy is replicated (rank 0's initial weights are broadcast) and kept identical by the
collectives of ``rl/ppo.py`` (RCCL: all-gather of returns/advantages, gradient and mean-KL all-reduce); only rank 0 writes logs
and checkpoints."""
import os

from legged_games_gym_amd.envs import *  # noqa: F401,F403  (registers the tasks)
from legged_games_gym_amd.utils import get_args
from legged_games_gym_amd.utils.task_registry import task_registry


def _init_distributed(args):
    """torchrun / torch.distributed.run environment -> process group, per-rank device, per-rank env seed."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    import torch
    import torch.distributed as dist
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LG_LOCAL_DEVICE", os.environ.get("LOCAL_RANK", "0")))     # LG_LOCAL_DEVICE: rehearsal of N ranks on one GPU
    on_gpu = torch.cuda.is_available() and not str(args.rl_device).startswith("cpu")
    if on_gpu:
        torch.cuda.set_device(local)
        args.sim_device = args.rl_device = f"cuda:{local}"
        args.sim_device_id = local
    dist.init_process_group(backend=os.environ.get("LG_DIST_BACKEND", "nccl" if on_gpu else "gloo"))      # "nccl" is RCCL on ROCm
    _, train_cfg = task_registry.get_cfgs(args.task)
    train_cfg.seed = int(train_cfg.seed) + rank          # the env seeds from the registered train cfg (reference quirk Q10)
    return rank, world


def train(args):
    rank, world = _init_distributed(args)
    env, env_cfg = task_registry.make_env(name=args.task, args=args)
    ppo_runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args, **({} if rank == 0 else {"log_root": None}))
    ppo_runner.learn(num_learning_iterations=train_cfg.runner.max_iterations, init_at_random_ep_len=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    train(get_args())
