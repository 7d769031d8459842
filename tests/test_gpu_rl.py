"""-m gpu: the device-side pieces of the PPO stack: GAE scan kernel and the graph-captured update."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _storage(T=24, N=777, obs=48, act=12, seed=0, st=None):
    """Random rollout; refills ``st`` in place when given (the runner re-uses one storage: a captured update graph reads it)."""
    from legged_games_gym_amd.rl.ppo import RolloutStorage
    st = st if st is not None else RolloutStorage(N, T, [obs], [None], [act], device="cuda")
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    st.observations.copy_(r(T, N, obs)); st.actions.copy_(r(T, N, act)); st.rewards.copy_(r(T, N, 1) * 0.02)
    st.dones.copy_((torch.rand(T, N, 1, device="cuda", generator=g) < 0.03).byte())
    st.values.copy_(r(T, N, 1) * 0.3); st.mu.copy_(r(T, N, act) * 0.2); st.sigma.fill_(1.0)
    st.actions_log_prob.copy_(torch.distributions.Normal(st.mu, st.sigma).log_prob(st.actions).sum(-1, keepdim=True))
    st.step = T
    return st


def test_gae_kernel_matches_the_torch_scan():
    st = _storage()
    last = torch.randn(st.num_envs, 1, device="cuda") * 0.3
    assert st._gae_kernel(last, 0.99, 0.95)
    ret_k, adv_k = st.returns.clone(), st.advantages.clone()
    st.returns.zero_(); st.advantages.zero_()
    st._gae_torch(last, 0.99, 0.95)
    # the kernel contracts r + nd*gamma*next into FMAs: a few ulp on values of magnitude ~1
    assert float((ret_k - st.returns).abs().max()) < 2e-6 and float((adv_k - st.advantages).abs().max()) < 2e-6
    # episode boundaries cut the recursion: the last step of a finished episode sees only its own reward
    t, e = 5, int(torch.nonzero(st.dones[5, :, 0])[0])
    assert abs(float(ret_k[t, e, 0]) - float(st.rewards[t, e, 0])) < 1e-6


def test_graph_captured_update_equals_the_eager_update():
    """Same data, same initial weights, same permutations: three updates with the captured mini-batch graph (first one is the
    eager warm-up) against three eager updates."""
    from legged_games_gym_amd.rl import ActorCritic
    from legged_games_gym_amd.rl.ppo import PPO
    torch.manual_seed(0)
    ac0 = ActorCritic(48, 48, 12, actor_hidden_dims=[128, 64, 32], critic_hidden_dims=[128, 64, 32], activation="elu").cuda()
    out = []
    for graphed in (False, True):
        alg = PPO(copy.deepcopy(ac0), num_learning_epochs=2, num_mini_batches=4, learning_rate=1e-3, schedule="adaptive", desired_kl=0.01,
                  entropy_coef=0.01, device="cuda", graphed_update=graphed)
        losses = []
        for it in range(3):
            alg.storage = _storage(seed=it, st=alg.storage)
            alg.storage.compute_returns(torch.zeros(alg.storage.num_envs, 1, device="cuda"), 0.99, 0.95)
            B = alg.storage.num_envs * alg.storage.num_transitions_per_env
            perm = torch.randperm(B - B % 4, device="cuda", generator=torch.Generator(device="cuda").manual_seed(100 + it))
            losses.append(alg.update(perm))                   # same mini-batch permutation on both paths
        out.append((losses, [p.detach().clone() for p in alg.actor_critic.parameters()], alg.learning_rate))
    (l_e, p_e, lr_e), (l_g, p_g, lr_g) = out
    assert abs(lr_e - lr_g) < 1e-9 * max(1.0, lr_e) + 1e-12 or abs(lr_e - lr_g) / lr_e < 1e-5
    for (ve, se), (vg, sg) in zip(l_e, l_g):
        assert abs(ve - vg) < 1e-5 and abs(se - sg) < 1e-5
    for a, b in zip(p_e, p_g):
        assert float((a - b).abs().max()) < 2e-5
