"""N>1 path of the PPO update with world_size-2 gloo on CPU: the all-gathered advantage normalisation and
the flattened gradient all-reduce must reproduce the single-process result on the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from legged_games_gym_amd.rl import ActorCritic, PPO


def _fill(storage, seed, n_env, T, n_obs, n_act):
    g = torch.Generator().manual_seed(seed)
    storage.observations.copy_(torch.randn(T, n_env, n_obs, generator=g))
    storage.actions.copy_(torch.randn(T, n_env, n_act, generator=g))
    storage.rewards.copy_(torch.randn(T, n_env, 1, generator=g))
    storage.dones.copy_((torch.rand(T, n_env, 1, generator=g) < 0.05).byte())
    storage.values.copy_(torch.randn(T, n_env, 1, generator=g))
    storage.actions_log_prob.copy_(-torch.rand(T, n_env, 1, generator=g))
    storage.mu.copy_(torch.randn(T, n_env, n_act, generator=g))
    storage.sigma.copy_(torch.ones(T, n_env, n_act))
    storage.step = T


def _make(n_env, T=6, n_obs=10, n_act=3):
    torch.manual_seed(0)
    ac = ActorCritic(n_obs, n_obs, n_act, actor_hidden_dims=[16], critic_hidden_dims=[16])
    alg = PPO(ac, num_learning_epochs=1, num_mini_batches=1, schedule="fixed", learning_rate=1e-3)
    alg.init_storage(n_env, T, [n_obs], [None], [n_act])
    return alg


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    alg = _make(8)
    _fill(alg.storage, 100 + rank, 8, 6, 10, 3)
    alg.storage.compute_returns(torch.zeros(8, 1), 0.99, 0.95)
    adv = alg.storage.advantages.clone()
    torch.manual_seed(5)                     # same mini-batch permutation on both ranks
    alg.update()
    torch.save({"adv": adv, "params": [p.detach().clone() for p in alg.actor_critic.parameters()]}, f"{out}/r{rank}.pt")
    dist.destroy_process_group()


def test_world2_matches_single_process(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "r1.pt", weights_only=True)
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)                             # replicas stay identical after the all-reduced step
    # single process on the concatenated envs
    alg = _make(16)
    a0, a1 = _make(8), _make(8)
    _fill(a0.storage, 100, 8, 6, 10, 3); _fill(a1.storage, 101, 8, 6, 10, 3)
    for name in ("observations", "actions", "rewards", "dones", "values", "actions_log_prob", "mu", "sigma"):
        getattr(alg.storage, name).copy_(torch.cat((getattr(a0.storage, name), getattr(a1.storage, name)), dim=1))
    alg.storage.step = 6
    alg.storage.compute_returns(torch.zeros(16, 1), 0.99, 0.95)
    want = alg.storage.advantages
    got = torch.cat((r0["adv"], r1["adv"]), dim=1)
    assert torch.allclose(got, want, atol=1e-5)              # global normalisation via the all-gather


def test_split_k_linear_has_the_gradients_of_nn_linear():
    """rl/ppo.py:_LinearSplitK (the weight gradient as a batched GEMM over row chunks) == nn.Linear under autograd."""
    import torch.nn as nn
    from legged_games_gym_amd.rl.ppo import _mlp_split_k
    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(37, 64), nn.ELU(), nn.Linear(64, 32), nn.ELU(), nn.Linear(32, 5)).double()
    for M in (4096, 777):                                  # 16 chunks / no divisor -> single GEMM
        x = torch.randn(M, 37, dtype=torch.float64)
        dy = torch.randn(M, 5, dtype=torch.float64)
        net.zero_grad(); net(x).backward(dy)
        want = [p.grad.clone() for p in net.parameters()]
        net.zero_grad(); _mlp_split_k(net, x).backward(dy)
        for w, p in zip(want, net.parameters()):
            assert torch.allclose(w, p.grad, rtol=1e-10, atol=1e-12)
