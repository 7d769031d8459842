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


def test_fused_ppo_loss_matches_autograd():
    """lg_ppo_loss: loss statistics and d loss / d (mu, std, value) against torch autograd on the reference's loss expression."""
    from legged_games_gym_amd import capi
    lib = capi.load_library()
    torch.manual_seed(1)
    st = _storage(T=8, N=500, seed=3)
    st.compute_returns(torch.zeros(500, 1, device="cuda"), 0.99, 0.95)
    B, A, mb = 8 * 500, 12, 1500
    ix = torch.randperm(B, device="cuda")[:mb]
    for clipped in (1, 0):
        mu = (st.mu.flatten(0, 1)[ix] + 0.3 * torch.randn(mb, A, device="cuda")).requires_grad_()
        std = (0.8 + 0.4 * torch.rand(A, device="cuda")).requires_grad_()
        val = (st.values.flatten(0, 1)[ix] + 0.4 * torch.randn(mb, 1, device="cuda")).requires_grad_()
        act, olp, omu, osg = (t.flatten(0, 1)[ix] for t in (st.actions, st.actions_log_prob, st.mu, st.sigma))
        adv, tval, ret = (t.flatten(0, 1)[ix] for t in (st.advantages, st.values, st.returns))
        clip, vc, ec = 0.2, 1.0, 0.01
        dist_ = torch.distributions.Normal(mu, mu * 0.0 + std)
        lp = dist_.log_prob(act).sum(-1)
        ratio = torch.exp(lp - olp.squeeze())
        a = adv.squeeze()
        surrogate = torch.max(-a * ratio, -a * torch.clamp(ratio, 1 - clip, 1 + clip)).mean()
        if clipped:
            vclip = tval + (val - tval).clamp(-clip, clip)
            vloss = torch.max((val - ret).pow(2), (vclip - ret).pow(2)).mean()
        else:
            vloss = (ret - val).pow(2).mean()
        ent = dist_.entropy().sum(-1).mean()
        kl = torch.sum(torch.log(std / osg + 1e-5) + (osg.square() + (omu - mu).square()) / (2.0 * std.square()) - 0.5, dim=-1).mean()
        (surrogate + vc * vloss - ec * ent).backward()
        d_mu, d_val, d_std, stats = torch.empty(mb, A, device="cuda"), torch.empty(mb, 1, device="cuda"), torch.zeros(A, device="cuda"), torch.zeros(4, device="cuda")
        p = lambda t: t.data_ptr()
        rc = lib.lg_ppo_loss(p(mu), p(std), p(val), p(ix), p(st.actions), p(st.actions_log_prob), p(st.mu), p(st.sigma), p(st.advantages), p(st.values),
                             p(st.returns), clip, vc, ec, clipped, p(d_mu), p(d_std), p(d_val), p(stats), mb, A, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        ref = torch.stack((surrogate, vloss, kl, ent)).detach()
        assert float((stats - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max())), (stats, ref)
        assert float((d_mu - mu.grad).abs().max()) < 1e-6 + 1e-4 * float(mu.grad.abs().max())
        assert float((d_val - val.grad).abs().max()) < 1e-7 + 1e-4 * float(val.grad.abs().max())
        assert float((d_std - std.grad).abs().max()) < 1e-6 + 1e-4 * float(std.grad.abs().max())


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
        assert abs(ve - vg) < 2e-5 and abs(se - sg) < 2e-5
    for a, b in zip(p_e, p_g):
        assert float((a - b).abs().max()) < 1e-4          # fused loss kernel vs torch loss: rounding, amplified by three Adam updates


def _mlp(i, o, seed):
    import torch.nn as nn
    torch.manual_seed(seed)
    return nn.Sequential(nn.Linear(i, 128), nn.ELU(), nn.Linear(128, 64), nn.ELU(), nn.Linear(64, 32), nn.ELU(), nn.Linear(32, o)).cuda()


@pytest.mark.parametrize("mb,gather", [(1000, True), (24576, True), (37, False)])
def test_mlp_kernels_match_autograd(mb, gather):
    """lg_mlp_forward / lg_mlp_backward (MFMA learner kernels) against torch autograd on the same nn.Sequential modules."""
    from legged_games_gym_amd.rl.mlp_kernels import MlpTrainer
    actor, critic = _mlp(48, 12, 0), _mlp(48, 1, 1)
    R = 30000
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(R, 48, device="cuda", generator=g)
    rows = torch.randperm(R, device="cuda", generator=g)[:mb] if gather else None
    tr = MlpTrainer([actor, critic], [x, x], mb)
    assert tr.supported
    mu, val = tr.forward(rows)
    xb = x[rows] if gather else x[:mb]
    mu_ref, val_ref = actor(xb), critic(xb)
    assert float((mu - mu_ref.detach()).abs().max()) < 2e-5 and float((val - val_ref.detach()).abs().max()) < 2e-5
    d_mu = torch.randn(mb, 12, device="cuda", generator=g) / mb
    d_val = torch.randn(mb, 1, device="cuda", generator=g) / mb
    torch.autograd.backward([mu_ref, val_ref], [d_mu, d_val])
    want = [p.grad.clone() for net in (actor, critic) for p in net.parameters()]
    for net in (actor, critic):
        for p in net.parameters():
            p.grad.fill_(float("nan"))                   # the kernel overwrites every element
    tr.refresh()
    tr.grad_outputs[0].copy_(d_mu); tr.grad_outputs[1].copy_(d_val)
    tr.backward(rows)
    got = [p.grad for net in (actor, critic) for p in net.parameters()]
    for w, h in zip(want, got):
        scale = float(w.abs().max()) + 1e-12
        assert float((w - h).abs().max()) < 2e-4 * scale + 1e-9, (w.shape, float((w - h).abs().max()), scale)
    # fixed reduction order: bit-reproducible
    first = [h.clone() for h in got]
    tr.backward(rows)
    assert all(torch.equal(a, b) for a, b in zip(first, got))


@pytest.mark.parametrize("wide", [False, True])
def test_adam_kernel_matches_torch_adam_with_clipping(wide):
    """lg_adam_step against clip_grad_norm_ + torch.optim.Adam(capturable=True).step() on the optimiser's own state tensors
    (wide: the 235-512-256-128 networks of the rough tasks, 290 k parameters)."""
    import torch.nn as nn
    from legged_games_gym_amd import capi
    lib = capi.load_library()

    def wide_mlp():
        torch.manual_seed(3)
        return nn.Sequential(nn.Linear(235, 512), nn.ELU(), nn.Linear(512, 256), nn.ELU(), nn.Linear(256, 128), nn.ELU(), nn.Linear(128, 12)).cuda()
    nets = [wide_mlp(), wide_mlp()] if wide else [_mlp(48, 12, 3), _mlp(48, 12, 3)]
    lrs = [torch.tensor(1e-3, device="cuda"), torch.tensor(1e-3, device="cuda")]
    opts = [torch.optim.Adam(n.parameters(), lr=l, capturable=True) for n, l in zip(nets, lrs)]
    g = torch.Generator(device="cuda").manual_seed(9)
    scratch = torch.zeros(capi.LG_ADAM_SCRATCH_FLOATS, device="cuda")
    for it in range(6):
        grads = [torch.randn(q.shape, device="cuda", generator=g) * (3.0 if it % 2 else 0.01) for q in nets[0].parameters()]
        for n in nets:
            for q, gr in zip(n.parameters(), grads):
                q.grad = gr.clone()
        kl = torch.tensor([0.05, 0.001, 0.01][it % 3], device="cuda")
        # torch side (reference order: KL rule, clip, step)
        lr = lrs[0]
        lr.copy_(torch.where(kl > 0.02, torch.clamp(lr / 1.5, min=1e-5), torch.where((kl < 0.005) & (kl > 0), torch.clamp(lr * 1.5, max=1e-2), lr)))
        nn.utils.clip_grad_norm_(nets[0].parameters(), 1.0)
        opts[0].step()
        if it == 0:                       # the first step creates the optimiser state: torch's on both sides
            lrs[1].copy_(lr); nn.utils.clip_grad_norm_(nets[1].parameters(), 1.0); opts[1].step()
            continue
        params = list(nets[1].parameters())
        table = (capi.lg_adam_tensor * len(params))()
        for i, q in enumerate(params):
            stt = opts[1].state[q]
            table[i].param, table[i].grad, table[i].exp_avg, table[i].exp_avg_sq = q.data_ptr(), q.grad.data_ptr(), stt["exp_avg"].data_ptr(), stt["exp_avg_sq"].data_ptr()
            table[i].step, table[i].numel = stt["step"].data_ptr(), q.numel()
        rc = lib.lg_adam_step(table, len(params), lrs[1].data_ptr(), 0.9, 0.999, 1e-8, 1.0, kl.data_ptr(), 0.01, scratch.data_ptr(),
                              torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.lg_last_error()
        assert abs(float(lrs[0]) - float(lrs[1])) < 1e-9
        for a, b in zip(nets[0].parameters(), nets[1].parameters()):
            assert float((a - b).detach().abs().max()) < 2e-6, it
        for a, b in zip(nets[0].parameters(), nets[1].parameters()):
            assert float((opts[0].state[a]["exp_avg_sq"] - opts[1].state[b]["exp_avg_sq"]).abs().max()) < 1e-6
            assert float(opts[0].state[a]["step"]) == float(opts[1].state[b]["step"])


def test_rollout_record_kernel_matches_the_torch_bookkeeping():
    """lg_rollout_record == the copies into the rollout storage + the runner's episode statistics (tensor-op version)."""
    from legged_games_gym_amd import capi
    lib = capi.load_library()
    N, O, A, T = 1000, 48, 12, 5
    g = torch.Generator(device="cuda").manual_seed(2)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    st = {k: torch.zeros(T, N, d, device="cuda") for k, d in (("obs", O), ("act", A), ("mu", A), ("rew", 1), ("tout", 1))}
    st["done"] = torch.zeros(T, N, 1, device="cuda", dtype=torch.uint8)
    cur_rew, cur_len, sums = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda"), torch.zeros(3, device="cuda")
    ref = {"cur_rew": torch.zeros(N, device="cuda"), "cur_len": torch.zeros(N, device="cuda"), "sums": torch.zeros(3, device="cuda")}
    step = capi.lg_rollout_step()
    step.num_envs, step.num_obs, step.num_actions = N, O, A
    step.cur_return, step.cur_length, step.sums = cur_rew.data_ptr(), cur_len.data_ptr(), sums.data_ptr()
    std = 0.5 + torch.rand(A, device="cuda", generator=g)
    st["sig"], st["lp"] = torch.zeros(T, N, A, device="cuda"), torch.zeros(T, N, 1, device="cuda")
    step.std = std.data_ptr()
    keep = []
    for t in range(T):
        obs, act, mu, rew = r(N, O), r(N, A), r(N, A), r(N)
        done = (torch.rand(N, device="cuda", generator=g) < 0.2).to(torch.uint8)
        tout = (done.bool() & (torch.rand(N, device="cuda", generator=g) < 0.5)).to(torch.uint8)
        keep.append((obs, act, mu, rew, done, tout))
        step.obs, step.actions, step.mean, step.rewards, step.dones, step.time_outs = (x.data_ptr() for x in (obs, act, mu, rew, done, tout))
        step.storage_obs, step.storage_actions, step.storage_mu = st["obs"][t].data_ptr(), st["act"][t].data_ptr(), st["mu"][t].data_ptr()
        step.storage_rewards, step.storage_dones, step.storage_time_outs = st["rew"][t].data_ptr(), st["done"][t].data_ptr(), st["tout"][t].data_ptr()
        step.storage_sigma, step.storage_log_prob = st["sig"][t].data_ptr(), st["lp"][t].data_ptr()
        assert lib.lg_rollout_record(step, torch.cuda.current_stream().cuda_stream) == 0, lib.lg_last_error()
        d = done.float()
        ref["cur_rew"] += rew; ref["cur_len"] += 1.0
        ref["sums"] += torch.stack(((ref["cur_rew"] * d).sum(), (ref["cur_len"] * d).sum(), d.sum()))
        ref["cur_rew"] *= 1.0 - d; ref["cur_len"] *= 1.0 - d
    for t, (obs, act, mu, rew, done, tout) in enumerate(keep):
        assert torch.equal(st["obs"][t], obs) and torch.equal(st["act"][t], act) and torch.equal(st["mu"][t], mu)
        assert torch.equal(st["rew"][t, :, 0], rew) and torch.equal(st["done"][t, :, 0], done) and torch.equal(st["tout"][t, :, 0], tout.float())
        want_lp = torch.distributions.Normal(mu, std.expand_as(mu)).log_prob(act).sum(-1)
        assert torch.equal(st["sig"][t], std.expand(N, A)) and float((st["lp"][t, :, 0] - want_lp).abs().max()) < 2e-5
    assert torch.equal(cur_len, ref["cur_len"]) and float((cur_rew - ref["cur_rew"]).abs().max()) < 1e-6
    assert float(sums[2]) == float(ref["sums"][2]) and float(sums[1]) == float(ref["sums"][1])
    assert abs(float(sums[0]) - float(ref["sums"][0])) < 1e-3          # atomic accumulation order


def test_mlp_kernels_ragged_widths_and_a_single_net():
    """Input width not a multiple of 4 (guarded scalar loads, zero-padded LDS image), 7 outputs, one net per launch, no gather."""
    import torch.nn as nn
    from legged_games_gym_amd.rl.mlp_kernels import MlpTrainer
    torch.manual_seed(4)
    net = nn.Sequential(nn.Linear(45, 128), nn.ELU(), nn.Linear(128, 64), nn.ELU(), nn.Linear(64, 32), nn.ELU(), nn.Linear(32, 7)).cuda()
    mb = 16 * 300 + 5
    x = torch.randn(mb, 45, device="cuda")
    tr = MlpTrainer([net], [x], mb)
    assert tr.supported
    (y,) = tr.forward(None)
    ref = net(x)
    assert float((y - ref.detach()).abs().max()) < 2e-5
    dy = torch.randn(mb, 7, device="cuda") / mb
    ref.backward(dy)
    want = [p.grad.clone() for p in net.parameters()]
    for p in net.parameters():
        p.grad.fill_(float("nan"))
    tr.refresh()
    tr.grad_outputs[0].copy_(dy)
    tr.backward(None)
    for w, p in zip(want, net.parameters()):
        assert float((w - p.grad).abs().max()) < 2e-4 * float(w.abs().max()) + 1e-9
    # a shape the kernels are not built for is reported, not mis-computed
    wide = nn.Sequential(nn.Linear(48, 512), nn.ELU(), nn.Linear(512, 256), nn.ELU(), nn.Linear(256, 128), nn.ELU(), nn.Linear(128, 12)).cuda()
    assert not MlpTrainer([wide], [torch.randn(64, 48, device="cuda")], 64).supported


@pytest.mark.parametrize("clipped,critic_obs", [(1, 48), (0, 48), (1, 45)])
def test_fused_ppo_minibatch_equals_forward_loss_backward(clipped, critic_obs):
    """lg_ppo_minibatch (forward + PPO loss + backward in one kernel) against lg_mlp_forward -> lg_ppo_loss -> lg_mlp_backward."""
    from legged_games_gym_amd import capi
    from legged_games_gym_amd.rl.mlp_kernels import MlpTrainer
    lib = capi.load_library()
    st = _storage(T=8, N=1000, seed=7)
    st.compute_returns(torch.zeros(1000, 1, device="cuda"), 0.99, 0.95)
    B, A, mb = 8000, 12, 16 * 187 + 9
    ix = torch.randperm(B, device="cuda")[:mb]
    actor, critic = _mlp(48, 12, 0), _mlp(critic_obs, 1, 1)
    std = (0.7 + 0.5 * torch.rand(A, device="cuda"))
    obs = st.observations.flatten(0, 1)
    cobs = obs if critic_obs == 48 else torch.randn(B, critic_obs, device="cuda")      # privileged observations of another width
    tr = MlpTrainer([actor, critic], [obs, cobs], mb)
    p = lambda t: t.data_ptr()
    # three-call path
    mu, val = tr.forward(ix)
    d_std, stats = torch.zeros(A, device="cuda"), torch.zeros(4, device="cuda")
    rc = lib.lg_ppo_loss(p(mu), p(std), p(val), p(ix), p(st.actions), p(st.actions_log_prob), p(st.mu), p(st.sigma), p(st.advantages), p(st.values),
                         p(st.returns), 0.2, 1.0, 0.01, clipped, p(tr.grad_outputs[0]), p(d_std), p(tr.grad_outputs[1]), p(stats), mb, A,
                         torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    tr.backward(ix)
    want = [q.grad.clone() for net in (actor, critic) for q in net.parameters()]
    for net in (actor, critic):
        for q in net.parameters():
            q.grad.fill_(float("nan"))
    # fused path
    b = capi.lg_ppo_batch()
    b.actions, b.old_log_prob, b.old_mu, b.old_sigma = p(st.actions), p(st.actions_log_prob), p(st.mu), p(st.sigma)
    b.advantages, b.old_values, b.returns, b.std = p(st.advantages), p(st.values), p(st.returns), p(std)
    b.clip, b.value_coef, b.entropy_coef, b.use_clipped_value = 0.2, 1.0, 0.01, clipped
    d_std2, stats2 = torch.full((A,), float("nan"), device="cuda"), torch.full((4,), float("nan"), device="cuda")
    b.d_std, b.stats = p(d_std2), p(stats2)
    tr.ppo_minibatch(ix, b)
    got = [q.grad for net in (actor, critic) for q in net.parameters()]
    for w, h in zip(want, got):
        assert float((w - h).abs().max()) < 1e-5 * float(w.abs().max()) + 1e-10, w.shape
    assert float((d_std - d_std2).abs().max()) < 1e-6 and float((stats - stats2).abs().max()) < 1e-5, (d_std, d_std2, stats, stats2)


def test_kernel_update_with_fixed_schedule_keeps_the_learning_rate():
    """schedule='fixed': lg_adam_step gets no KL pointer, the device learning rate must not move; parameters and Adam state do."""
    from legged_games_gym_amd.rl import ActorCritic
    from legged_games_gym_amd.rl.ppo import PPO
    torch.manual_seed(0)
    ac = ActorCritic(48, 48, 12, actor_hidden_dims=[128, 64, 32], critic_hidden_dims=[128, 64, 32], activation="elu").cuda()
    before = [p.detach().clone() for p in ac.parameters()]
    alg = PPO(ac, num_learning_epochs=2, num_mini_batches=4, learning_rate=3e-4, schedule="fixed", entropy_coef=0.01, device="cuda")
    for it in range(3):                                   # eager warm-up, capture, replay
        alg.storage = _storage(seed=it, st=alg.storage)
        alg.storage.compute_returns(torch.zeros(alg.storage.num_envs, 1, device="cuda"), 0.99, 0.95)
        v, s = alg.update()
        assert v == v and s == s                          # finite running losses
    assert alg._mlp is not None and alg._graph is not None and alg._graph_whole
    assert abs(alg.learning_rate - 3e-4) < 1e-10
    assert all(float(st_["step"]) == 3 * 8 for st_ in alg.optimizer.state.values())
    assert all(not torch.equal(a, b) for a, b in zip(before, ac.parameters()))
    assert all(bool(torch.isfinite(p).all()) for p in ac.parameters())


def _dp_worker(rank, world, port, out, obs=48, hidden=(128, 64, 32)):
    """One data-parallel rank of the KERNEL update (learner kernels + lg_adam_step, one flat all-reduce per mini-batch step) on the
    shared test GPU; gloo stands in for RCCL (two ranks cannot share one device under RCCL)."""
    import os
    import torch.distributed as dist
    from legged_games_gym_amd.rl import ActorCritic, PPO
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    from legged_games_gym_amd import capi
    capi.load_library().lg_mlp_wide_set_precision(0)      # exact f32 products: the first Adam step is lr * sign(g) wherever |g| is tiny,
                                                           # so the single-process comparison below needs gradients equal to rounding
    ac = ActorCritic(obs, obs, 12, actor_hidden_dims=list(hidden), critic_hidden_dims=list(hidden))
    alg = PPO(ac, num_learning_epochs=1, num_mini_batches=1, schedule="adaptive", desired_kl=0.01, learning_rate=1e-3, entropy_coef=0.01, device="cuda:0")
    T, N = 8, 256
    alg.init_storage(N, T, [obs], [None], [12])
    snaps = []
    for it in range(4):
        _storage(T=T, N=N, seed=100 * it + rank, st=alg.storage, obs=obs)
        alg.storage.compute_returns(torch.zeros(N, 1, device="cuda"), 0.99, 0.95)
        perm = torch.randperm(T * N, device="cuda", generator=torch.Generator(device="cuda").manual_seed(it))
        alg.update(perm=perm)
        snaps.append([p.detach().cpu().clone() for p in alg.actor_critic.parameters()] + [torch.tensor(alg.learning_rate)])
    used_flat = getattr(alg, "_gflat", None) is not None and alg._mlp is not None
    torch.save({"snaps": snaps, "flat": used_flat, "launches": alg.dp_launches, "graph_error": getattr(alg, "_dp_graph_error", None)}, f"{out}/r{rank}.pt")
    dist.destroy_process_group()


@pytest.mark.parametrize("obs,hidden", [(48, (128, 64, 32)), (235, (512, 256, 128))])
def test_data_parallel_kernel_update_two_ranks_one_gpu(tmp_path, obs, hidden):
    """the N > 1 update on the kernel path, for the flat networks (lg_ppo_minibatch) and for the wide ones of config 4
    (lg_mlp_wide_*: the 8 x 4096 anymal_c_rough case).  Two gloo ranks on the one GPU: (i) replicas stay BIT-identical over 4
    updates (the gradients come out of one all-reduced flat buffer on both), (ii) the first update equals the single-process update on
    the concatenated batch (mean of the two shards' gradients and KLs, global advantage normalisation through the all-gather)."""
    import socket
    import torch.multiprocessing as mp
    from legged_games_gym_amd.rl import ActorCritic, PPO
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path), obs, hidden), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt", weights_only=True); r1 = torch.load(tmp_path / "r1.pt", weights_only=True)
    assert r0["flat"] and r1["flat"]                                     # the kernel path with the flat gradient buffer was the one that ran
    # updates 2..4 ran as [backward graph] -> eager flat all-reduce -> [optimiser graph]: no kernel launch from the host
    assert r0["graph_error"] is None and r0["launches"] == {"graph_replays": 2, "collectives": 1, "kernel_launches_from_host": 0}, (r0["graph_error"], r0["launches"])
    for a, b in zip(r0["snaps"], r1["snaps"]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    # single process, both shards as one batch of 512 envs (first update only: later ones start from different storages per rank)
    from legged_games_gym_amd import capi
    prev = capi.load_library().lg_mlp_wide_set_precision(0)
    torch.manual_seed(0)
    ac = ActorCritic(obs, obs, 12, actor_hidden_dims=list(hidden), critic_hidden_dims=list(hidden))
    alg = PPO(ac, num_learning_epochs=1, num_mini_batches=1, schedule="adaptive", desired_kl=0.01, learning_rate=1e-3, entropy_coef=0.01, device="cuda:0")
    T, N = 8, 256
    alg.init_storage(2 * N, T, [obs], [None], [12])
    a, b = _storage(T=T, N=N, seed=0, obs=obs), _storage(T=T, N=N, seed=1, obs=obs)
    for name in ("observations", "actions", "rewards", "dones", "values", "actions_log_prob", "mu", "sigma"):
        getattr(alg.storage, name).copy_(torch.cat((getattr(a, name), getattr(b, name)), dim=1))
    alg.storage.step = T
    alg.storage.compute_returns(torch.zeros(2 * N, 1, device="cuda"), 0.99, 0.95)
    alg.update()
    capi.load_library().lg_mlp_wide_set_precision(prev)
    for want, got in zip(alg.actor_critic.parameters(), r0["snaps"][0]):
        d = (want.detach().cpu() - got).abs()
        # equal up to summation order; a handful of elements whose gradient is at rounding level may take the step the other way (2 lr)
        assert float((d > 2e-6 + 2e-4 * got.abs()).float().mean()) < 1e-3 and float(d.max()) <= 2.1e-3, (float(d.max()), float((d > 2e-6).float().mean()))
    assert abs(alg.learning_rate - float(r0["snaps"][0][-1])) < 1e-9


def _wide_mlp(i, o, seed):
    import torch.nn as nn
    torch.manual_seed(seed)
    return nn.Sequential(nn.Linear(i, 512), nn.ELU(), nn.Linear(512, 256), nn.ELU(), nn.Linear(256, 128), nn.ELU(), nn.Linear(128, o)).cuda()


@pytest.mark.parametrize("precision", [1, 0])
@pytest.mark.parametrize("obs,mb,gather", [(235, 24576, True), (169, 24576, True), (235, 1000, True), (169, 37, False)])
def test_wide_mlp_kernels_match_autograd(obs, mb, gather, precision):
    """lg_mlp_wide_forward / lg_mlp_wide_backward (layer-wise f32-MFMA GEMMs, csrc/lg_gemm.h) for the [235 | 169, 512, 256, 128, 12 | 1]
    networks of configs 3-5 (legged_robot_config.py:205-208) against torch autograd in float64 on the same modules; mb = 24 576 is
    the mini-batch of the reference's PPO settings (98 304 transitions / 4).  precision 0: exact f32 MFMA; 1 (default): split-bf16
    products (hi*hi + hi*lo + lo*hi, f32 accumulation; ~2^-15 per product) -- tolerances stated per precision."""
    import time
    from legged_games_gym_amd import capi
    from legged_games_gym_amd.rl.mlp_kernels import WideMlpTrainer
    lib = capi.load_library()
    old = lib.lg_mlp_wide_set_precision(precision)
    try:
        _wide_mlp_check(obs, mb, gather, precision, time, WideMlpTrainer)
    finally:
        lib.lg_mlp_wide_set_precision(old)


def _wide_mlp_check(obs, mb, gather, precision, time, WideMlpTrainer):
    out_tol, grad_tol = (2e-5, 1e-4) if precision == 0 else (1e-4, 3e-4)
    actor, critic = _wide_mlp(obs, 12, 0), _wide_mlp(obs, 1, 1)
    R = 40000
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(R, obs, device="cuda", generator=g)
    rows = torch.randperm(R, device="cuda", generator=g)[:mb] if gather else None
    xc = x if gather else x.clone()                      # same tensor: one gathered copy serves both nets; a clone: one copy per net
    tr = WideMlpTrainer([actor, critic], [x, xc], mb)
    assert tr.supported and not tr.has_fused_minibatch
    mu, val = tr.forward(rows)
    xb = (x[rows] if gather else x[:mb]).double()
    a64, c64 = copy.deepcopy(actor).double(), copy.deepcopy(critic).double()
    mu_ref, val_ref = a64(xb), c64(xb)
    assert float((mu.double() - mu_ref.detach()).abs().max()) < out_tol and float((val.double() - val_ref.detach()).abs().max()) < out_tol
    d_mu = torch.randn(mb, 12, device="cuda", generator=g) / mb
    d_val = torch.randn(mb, 1, device="cuda", generator=g) / mb
    torch.autograd.backward([mu_ref, val_ref], [d_mu.double(), d_val.double()])
    want = [p.grad.clone() for net in (a64, c64) for p in net.parameters()]
    for net in (actor, critic):
        for p in net.parameters():
            p.grad = torch.full_like(p, float("nan"))    # the kernels overwrite every element
    tr.refresh()
    tr.grad_outputs[0].copy_(d_mu); tr.grad_outputs[1].copy_(d_val)
    tr.backward(rows)
    got = [p.grad for net in (actor, critic) for p in net.parameters()]
    for w, h in zip(want, got):
        scale = float(w.abs().max()) + 1e-12
        assert float((w - h.double()).abs().max()) < grad_tol * scale + 1e-10, (w.shape, float((w - h.double()).abs().max()), scale)
    first = [h.clone() for h in got]                     # fixed reduction order: bit-reproducible
    tr.forward(rows); tr.backward(rows)
    assert all(torch.equal(a, b) for a, b in zip(first, got))
    if mb == 24576:                                      # informative: time of one forward + backward of both nets
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            tr.forward(rows); tr.backward(rows)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 100
        flops = 6.0 * mb * sum(p.numel() for net in (actor, critic) for p in net.parameters() if p.dim() == 2)
        print(f"wide MLP forward + backward, obs {obs}, mb {mb}, {'split-bf16' if precision else 'exact f32'}: {ms:.3f} ms = {flops / ms / 1e9:.1f} TFLOP/s (f32 MFMA peak 157)")


def test_transposed_lds_read_mapping_matches_the_documented_one():
    """lg_gemm.h's frag_tr builds MFMA operands with ds_read_b64_tr_b16 from a [k][feature] image; the lane mapping it assumes
    (lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3; lane i receives column i) is checked on the device by a
    stand-alone probe that __graft_entry__.build() compiles."""
    import os, subprocess
    probe = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(__file__))), "tools", "ubench", "tr16_probe")
    if not os.path.isfile(probe):
        pytest.skip("tools/ubench/tr16_probe not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([probe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "matches the documented mapping" in r.stdout, r.stdout + r.stderr
