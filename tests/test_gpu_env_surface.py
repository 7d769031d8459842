"""-m gpu: the outer boundary -- task_registry / VecEnv surface the reference's callers use."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _env(task="anymal_c_flat", n=64):
    from legged_games_gym_amd.envs import task_registry
    from legged_games_gym_amd.utils import get_args
    args = get_args(["--task", task, "--num_envs", str(n), "--headless", "--sim_device", "cuda:0", "--rl_device", "cuda:0"])
    return task_registry.make_env(task, args)


def test_vecenv_attributes_and_step_contract():
    env, cfg = _env()
    assert (env.num_envs, env.num_obs, env.num_privileged_obs, env.num_actions) == (64, 48, None, 12)
    assert env.max_episode_length == 1000 and abs(env.dt - 0.02) < 1e-12 and env.device == "cuda:0"
    assert env.obs_buf.shape == (64, 48) and env.rew_buf.shape == (64,) and env.episode_length_buf.dtype == torch.long
    assert env.root_states.shape == (64, 13) and env.dof_state.shape == (64 * 12, 2) and env.contact_forces.shape == (64, 17, 3)
    assert env.dof_pos.shape == (64, 12) and env.dof_pos.stride() == (24, 2)           # interleaved view, as in the reference
    assert env.sea_hidden_state.shape == (2, 64 * 12, 8) and env.sea_hidden_state_per_env.shape == (2, 64, 12, 8)
    assert env.feet_indices.tolist() == [4, 8, 12, 16] and env.termination_contact_indices.tolist() == [0]
    assert sorted(env.episode_sums) == sorted(["action_rate", "ang_vel_xy", "collision", "dof_acc", "feet_air_time", "lin_vel_z",
                                               "orientation", "torques", "tracking_ang_vel", "tracking_lin_vel"])
    obs, priv = env.reset()
    assert priv is None and obs.shape == (64, 48)
    env.episode_length_buf[:] = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))   # what rsl_rl does
    out = env.step(torch.randn(64, 12, device="cuda"))
    assert len(out) == 5
    obs, priv, rew, dones, infos = out
    assert dones.dtype == torch.bool and rew.dtype == torch.float32 and "time_outs" in infos and "episode" in infos
    assert set(infos["episode"]) == {"rew_" + k for k in env.episode_sums}
    assert env.get_observations() is env.obs_buf and torch.isfinite(obs).all()
    assert env.common_step_counter == 2            # reset() steps once
    with pytest.raises(ValueError):
        env.step(torch.zeros(3, 12, device="cuda"))


def test_registry_errors_and_cassie_rough_creation():
    from legged_games_gym_amd.envs import task_registry
    from legged_games_gym_amd.utils import get_args
    with pytest.raises(ValueError, match="not registered"):
        task_registry.make_env("high_level_game", get_args(["--headless"]))
    with pytest.raises(ValueError):
        task_registry.make_alg_runner(env=None, name=None, args=get_args(["--headless"]))
    env, cfg = _env("cassie", 32)                   # heightfield terrain + curriculum + height sampling + PD control
    assert env.num_obs == 169 and env.measured_heights.shape == (32, 121) and env.custom_origins
    assert env.height_samples.shape == (env.terrain.tot_rows, env.terrain.tot_cols)
    obs, _ = env.reset()
    for _ in range(30):
        obs, _, rew, dones, infos = env.step(torch.zeros(32, 12, device="cuda"))
    assert torch.isfinite(obs).all() and "terrain_level" in infos["episode"]
    assert obs.shape == (32, 169) and float(obs[:, 48:].abs().max()) <= 5.0 + 0.6


@pytest.mark.parametrize("task,height", [("a1", 0.26), ("anymal_b", 0.47)])
def test_a1_and_anymal_b_tasks_stand_on_rough_terrain(task, height):
    """The two remaining locomotion tasks of the reference registry (envs/__init__.py): A1 (PD control, box shapes)
    and ANYmal-B (actuator net); zero actions -> the robots stay up on the curriculum terrain's easy rows."""
    env, cfg = _env(task, 64)
    assert env.num_obs == 235 and env.num_actions == 12 and env.cfg.asset.name == task
    env.reset()
    z0 = torch.zeros(64, 12, device="cuda")
    dones = 0
    for _ in range(100):
        obs, _, rew, d, infos = env.step(z0)
        dones += int(d.sum())
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    rel_z = (env.root_states[:, 2] - env.measured_heights.mean(dim=1).clamp(-1, 1) * 0).cpu()
    assert dones < 32                                             # a few fall on steeper tiles; most keep standing
    assert float(env.contact_forces[:, env.feet_indices, 2].sum(dim=1).median()) > 0.5 * 9.81 * {"a1": 12.454, "anymal_b": 30.62}[task]
    assert float((env.base_lin_vel.norm(dim=1) < 0.5).float().mean()) > 0.7 and rel_z.isfinite().all()


def test_bundled_runner_learns_a_few_iterations(tmp_path):
    from legged_games_gym_amd.envs import task_registry
    from legged_games_gym_amd.utils import get_args
    from legged_games_gym_amd.utils.helpers import get_load_path
    args = get_args(["--task", "anymal_c_flat", "--num_envs", "128", "--headless", "--sim_device", "cuda:0", "--rl_device", "cuda:0", "--max_iterations", "2"])
    env, _ = task_registry.make_env("anymal_c_flat", args)
    runner, train_cfg = task_registry.make_alg_runner(env, "anymal_c_flat", args, log_root=str(tmp_path))
    runner.learn(num_learning_iterations=2, init_at_random_ep_len=True)
    path = get_load_path(str(tmp_path))
    assert path.endswith("model_2.pt")
    policy = runner.get_inference_policy(device=env.device)
    assert policy(env.get_observations()).shape == (128, 12)


@pytest.mark.parametrize("precision", [1, 0])
def test_fused_actor_matches_torch_forward(precision):
    """lg_policy_act vs the plain PyTorch fp32 forward of the same ActorCritic, all three compiled-in shapes.  precision 0: the f32-MFMA
    kernels (2e-5 of the output scale); precision 1 (the default): the wide shapes run k_policy_act_wide, split-bf16 products with
    the lo*lo term dropped -- 16 significand bits per operand, tolerance 1e-4 of the output scale stated here."""
    from legged_games_gym_amd import capi
    from legged_games_gym_amd.rl import ActorCritic, FusedActor
    old = capi.load_library().lg_mlp_wide_set_precision(precision)
    try:
        _fused_actor_cases(2e-5 if precision == 0 else 1e-4)
    finally:
        capi.load_library().lg_mlp_wide_set_precision(old)


def _fused_actor_cases(tol):
    from legged_games_gym_amd.rl import ActorCritic, FusedActor
    for n_obs, hidden in ((48, [128, 64, 32]), (235, [512, 256, 128]), (169, [512, 256, 128])):
        torch.manual_seed(3)
        ac = ActorCritic(n_obs, n_obs, 12, actor_hidden_dims=hidden, critic_hidden_dims=hidden).cuda()
        with torch.no_grad():
            ac.std.copy_(torch.linspace(0.3, 1.4, 12))
        fa = FusedActor(ac, "cuda:0", seed=5)
        obs = torch.randn(1000, n_obs, device="cuda") * 2.0            # 1000: not a multiple of 16 or 32 (tail lanes)
        with torch.no_grad():
            want = ac.actor(obs)
        actions, mean = fa.act_with_mean(obs)
        actions, mean = actions.clone(), mean.clone()          # the wrapper reuses its output buffers
        torch.cuda.synchronize()
        scale = float(want.abs().max())
        assert float((mean - want).abs().max()) < tol * max(1.0, scale), (n_obs, float((mean - want).abs().max()))
        z = ((actions - mean) / ac.std.detach()).flatten()
        assert abs(float(z.mean())) < 0.03 and abs(float(z.std()) - 1.0) < 0.03 and float(z.abs().max()) < 6.0
        a2 = fa.act(obs).clone()
        assert not torch.equal(a2, actions)                            # fresh noise per call
        assert torch.allclose(fa.act_inference(obs), want, atol=tol * max(1.0, scale))
        with torch.no_grad():                                          # sync() re-uploads changed weights
            ac.actor[0].weight.mul_(0.5)
        fa.sync()
        with torch.no_grad():
            want2 = ac.actor(obs)
        assert torch.allclose(fa.act_inference(obs), want2, atol=tol * max(1.0, float(want2.abs().max())))


def test_fused_actor_follows_the_optimiser_through_the_device_repack():
    """lg_policy_load_device: after the torch parameters change, sync_device() (no host copy) must give the same actor as a
    fresh host-side upload, for the flat and the wide shapes."""
    from legged_games_gym_amd.rl import ActorCritic, FusedActor
    for n_obs, hidden in ((48, [128, 64, 32]), (235, [512, 256, 128])):
        torch.manual_seed(4)
        ac = ActorCritic(n_obs, n_obs, 12, actor_hidden_dims=hidden, critic_hidden_dims=hidden, activation="elu").to("cuda")
        fa = FusedActor(ac, "cuda:0", seed=1)
        with torch.no_grad():
            for prm in ac.parameters():
                prm.add_(0.05 * torch.randn_like(prm))
            ac.std.mul_(0.7)
        fa.sync_device()
        obs = torch.randn(333, n_obs, device="cuda")
        got = fa.act_inference(obs).clone()
        want = ac.act_inference(obs).detach()
        assert float((got - want).abs().max()) < (2e-5 if n_obs == 48 else 1e-4) * max(1.0, float(want.abs().max()))
        fb = FusedActor(ac, "cuda:0", seed=1)                 # host-side pack of the same parameters
        assert torch.equal(fb.act_inference(obs), got)
        a1, m1 = fa.act_with_mean(obs); a1 = a1.clone()
        a2, m2 = fb.act_with_mean(obs)
        assert torch.equal(a1, a2)                            # same std, same noise stream


def test_fused_policy_step_equals_actor_kernel_plus_step():
    """lg_step_policy (actor inside the step kernel) == lg_policy_act followed by lg_step on the same state and noise stream."""
    from legged_games_gym_amd.rl import ActorCritic, FusedActor
    from legged_games_gym_amd.utils.helpers import class_to_dict
    outs = []
    for fused_step in (False, True):
        env, cfg = _env("anymal_c_flat", 200)                  # 200 envs: a partial last workgroup
        from legged_games_gym_amd.envs import task_registry
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
                actor._host_step = env.common_step_counter      # same Philox step as the fused kernel uses (counter + 1)
                act, mean = (x.clone() for x in actor.act_with_mean(obs))
                obs, _, rew, dones, _ = env.step(act)
            rec.append((act, mean, obs.clone(), rew.clone(), dones.clone(), env.dof_pos.clone()))
        outs.append(rec)
    # same algorithm, same noise stream; the two template instantiations are compiled separately (fp contraction may differ by an
    # ulp per sub-step, which six policy steps of contact dynamics amplify): flags bit-equal, floats within 5e-5
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            assert torch.equal(x, y) if x.dtype == torch.bool else float((x - y).abs().max()) < 5e-5
    with pytest.raises(RuntimeError, match="fused policy step"):
        env2, _ = _env("cassie", 16)
        env2.step_policy(actor)


@pytest.mark.parametrize("task", ["anymal_c_flat", "anymal_c_rough"])
def test_two_rank_training_on_one_gpu(tmp_path, task):
    """scripts/train.py under torch.distributed.run: 2 ranks (sharing cuda:0 here, gloo instead of RCCL) shard the envs, rank 0's
    weights are broadcast, the PPO collectives keep the replicas in step, rank 0 alone logs.  Rehearsal of the 8-GPU launch, for the
    flat task and for config 4's task (anymal_c_rough: height field, wide actor / learner kernels)."""
    import os, socket, subprocess, sys
    sck = socket.socket(); sck.bind(("127.0.0.1", 0)); port = sck.getsockname()[1]; sck.close()
    repo = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
    env = dict(os.environ, LG_LOCAL_DEVICE="0", LG_DIST_BACKEND="gloo", PYTHONPATH=repo, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           "-m", "legged_games_gym_amd.scripts.train", f"--task={task}", "--headless", "--num_envs", "256", "--max_iterations", "3"]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("it ")]
    assert len(lines) == 3 and "it 2/3" in lines[-1]                 # one logger (rank 0), all iterations done


def test_step_returns_a_fresh_observation_tensor_like_the_reference():
    """legged_robot.py:215 re-creates obs_buf each step; rsl_rl's PPO.act holds the previous tensor by reference until
    process_env_step.  The obs returned by step t must therefore survive step t+1 untouched."""
    env, _ = _env()
    obs0, _ = env.reset()
    keep = obs0.clone()
    obs1, *_ = env.step(torch.randn(64, 12, device="cuda"))
    assert obs1.data_ptr() != obs0.data_ptr() and torch.equal(obs0, keep) and obs1 is env.obs_buf is env.get_observations()
    keep1 = obs1.clone()
    obs2, *_ = env.step(torch.randn(64, 12, device="cuda"))
    assert torch.equal(obs1, keep1) and not torch.equal(obs2, obs1)


def test_train_then_headless_play_resumes_from_checkpoint(tmp_path, monkeypatch):
    """scripts/train.py -> scripts/play.py flow (reference train.py:39-43, play.py:42-80): run dir naming, model_<it>.pt,
    get_load_path picking the latest run / checkpoint, inference policy driving the env."""
    import legged_games_gym_amd as pkg
    from legged_games_gym_amd.utils import task_registry as tr_mod
    from legged_games_gym_amd.envs import task_registry
    from legged_games_gym_amd.utils import get_args
    from legged_games_gym_amd.scripts.play import play
    monkeypatch.setattr(tr_mod, "LEGGED_GYM_ROOT_DIR", str(tmp_path))       # logs/<experiment>/<date>_<run>/ under tmp
    args = get_args(["--task", "anymal_c_flat", "--num_envs", "64", "--headless", "--sim_device", "cuda:0", "--rl_device", "cuda:0",
                     "--max_iterations", "1"])
    env, _ = task_registry.make_env("anymal_c_flat", args)
    runner, cfg = task_registry.make_alg_runner(env, "anymal_c_flat", args)
    runner.learn(1)
    import os
    root = tmp_path / "logs" / "flat_anymal_c"
    runs = os.listdir(root)
    assert len(runs) == 1 and sorted(os.listdir(root / runs[0])) == ["model_0.pt", "model_1.pt", "progress.csv"]
    env2 = play(get_args(["--task", "anymal_c_flat", "--headless", "--sim_device", "cuda:0", "--rl_device", "cuda:0"]), steps=5)
    assert env2.num_envs == 50 and not env2.cfg.noise.add_noise and torch.isfinite(env2.obs_buf).all()
    # restore the registered (shared, mutated-in-place like the reference) configs for the other tests
    c, t = task_registry.get_cfgs("anymal_c_flat")
    c.env.num_envs, c.noise.add_noise, c.domain_rand.randomize_friction, c.domain_rand.push_robots = 4096, True, True, True
    t.runner.resume = False


def test_resume_continues_training_on_the_kernel_update_path(tmp_path):
    """save -> fresh runner -> load(optimizer too) -> learn: the kernel update (lg_ppo_minibatch + lg_adam_step on the restored
    torch.optim.Adam state) picks up where the first runner stopped -- same weights and optimiser state as an uninterrupted run."""
    from legged_games_gym_amd.envs import task_registry
    from legged_games_gym_amd.utils import get_args

    def make():
        args = get_args(["--task", "anymal_c_flat", "--num_envs", "256", "--headless", "--sim_device", "cuda:0", "--rl_device", "cuda:0"])
        env, _ = task_registry.make_env("anymal_c_flat", args)
        runner, _ = task_registry.make_alg_runner(env, "anymal_c_flat", args, log_root=None)
        return runner
    a = make()
    a.learn(num_learning_iterations=3)
    ckpt = str(tmp_path / "model_3.pt")
    a.save(ckpt)
    b = make()
    b.load(ckpt)
    assert a.alg._mlp is not None, "the flat networks must take the learner-kernel path"
    for pa, pb in zip(a.alg.actor_critic.parameters(), b.alg.actor_critic.parameters()):
        assert torch.equal(pa, pb)
    steps = [float(s["step"]) for s in b.alg.optimizer.state.values()]
    assert steps and all(s == 3 * 20 for s in steps)              # 3 updates x 5 epochs x 4 mini-batches
    b.learn(num_learning_iterations=3)                            # eager warm-up update, capture, replay on the restored state
    assert all(float(s["step"]) == 6 * 20 for s in b.alg.optimizer.state.values())
    assert all(bool(torch.isfinite(p).all()) for p in b.alg.actor_critic.parameters())
    assert any(not torch.equal(pa, pb) for pa, pb in zip(a.alg.actor_critic.parameters(), b.alg.actor_critic.parameters()))


@pytest.mark.parametrize("task", ["anymal_c_flat", "cassie"])
def test_reference_style_smoke_script(task, capsys):
    """legged_games_gym_amd/tests/test_env.py (mirror of the reference's manual smoke script): zero actions, <= 10 envs."""
    from legged_games_gym_amd.tests.test_env import test_env as smoke_env
    from legged_games_gym_amd.utils import get_args
    args = get_args(["--task", task, "--headless", "--sim_device", "cuda:0", "--rl_device", "cuda:0"])
    resets, time_outs = smoke_env(args, steps=1200)
    assert "Done" in capsys.readouterr().out
    assert time_outs >= 10 or resets >= 10                        # every env finished at least one 1000-step episode


def test_rough_task_trains_on_the_wide_kernels_and_critic_pass_matches_torch(tmp_path):
    """anymal_c_rough through the bundled runner: the rollout's critic pass over the stored transitions runs the chain forward of the
    wide learner kernels (runner._critic_values -> lg_mlp_wide_forward) and must agree with the torch critic; two iterations of the
    update on lg_mlp_wide_* leave finite parameters and move them."""
    from legged_games_gym_amd.envs import task_registry
    from legged_games_gym_amd.utils import get_args
    from legged_games_gym_amd.rl.mlp_kernels import WideMlpTrainer
    args = get_args(["--task", "anymal_c_rough", "--num_envs", "128", "--headless", "--sim_device", "cuda:0", "--rl_device", "cuda:0",
                     "--max_iterations", "2"])
    env, _ = task_registry.make_env("anymal_c_rough", args)
    runner, _ = task_registry.make_alg_runner(env, "anymal_c_rough", args, log_root=str(tmp_path))
    ac = runner.alg.actor_critic
    before = [p.detach().clone() for p in ac.parameters()]
    runner.learn(2)
    assert all(torch.isfinite(p).all() for p in ac.parameters())
    assert any(not torch.equal(a, b) for a, b in zip(before, ac.parameters()))
    st = runner.alg.storage
    got = runner._critic_values(st).clone().flatten()
    assert isinstance(runner._critic_fwd, WideMlpTrainer) and runner._critic_fwd.supported
    cobs = (st.privileged_observations if st.privileged_observations is not None else st.observations).flatten(0, 1)
    with torch.no_grad():
        want = ac.evaluate(cobs).flatten()
    assert float((got - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))
    c, _ = task_registry.get_cfgs("anymal_c_rough")
    c.env.num_envs = 4096


@pytest.mark.parametrize("skip,bit", [(1, 0x1), (2, 0x2)])
def test_missed_handover_is_a_sticky_error(skip, bit):
    """A bounded LDS hand-over poll that runs out must not end in rc 0 with wrong physics: the kernel sets a sticky status bit
    (host-mapped word), every later call on the handle fails with -20 until lg_clear_device_status().  lg_debug_handover()
    withholds the flag (1: the rigid-body wave's frame flag, 2: the helpers' self-collision flags) with a short poll bound."""
    import numpy as np
    from tests.common import make_setup, grid_origins
    from legged_games_gym_amd.device_sim import DeviceSim
    N = 64
    cfg, robot, p, names, model, w = make_setup("anymal_c_flat", N)
    assert p.self_collision == 1
    d = DeviceSim(p, model, robot, torch.device("cuda:0"), w)
    d.buf["env_origins"].copy_(torch.from_numpy(grid_origins(N)))
    ids = torch.arange(N, dtype=torch.int32)
    d.reset_idx(ids, 0)
    act = torch.zeros(N, 12, device="cuda")
    d.step(act, 1)
    assert d.sim.device_status(True) == 0
    d.sim.debug_handover(skip, 64)
    d.step(act, 2)                                        # launches fine: the status was clean when the call was made
    st = d.sim.device_status(True)
    assert st & bit, hex(st)
    with pytest.raises(RuntimeError, match="hand-over"):
        d.step(act, 3)
    with pytest.raises(RuntimeError, match="-20"):
        d.reset_idx(ids, 3)
    d.sim.debug_handover(0, 0)
    with pytest.raises(RuntimeError):                     # sticky: switching the hook off does not clear it
        d.step(act, 3)
    d.sim.clear_device_status()
    d.reset_idx(ids, 3)
    d.step(act, 4)
    assert d.sim.device_status(True) == 0 and torch.isfinite(d.buf["root_states"]).all()


def test_rollout_policy_segment_length_limits():
    """steps = 1 (first and last step at once) equals one lg_step_policy; steps outside [1, LG_MAX_ROLL_STEPS] and malformed storage are refused."""
    from legged_games_gym_amd import capi
    from legged_games_gym_amd.rl import ActorCritic, FusedActor
    from legged_games_gym_amd.utils.helpers import class_to_dict
    from legged_games_gym_amd.envs import task_registry
    outs = []
    for rolled in (False, True):
        env, _ = _env("anymal_c_flat", 70)
        _, train_cfg = task_registry.get_cfgs("anymal_c_flat")
        torch.manual_seed(3)
        ac = ActorCritic(env.num_obs, env.num_obs, env.num_actions, **class_to_dict(train_cfg.policy)).to("cuda")
        actor = FusedActor(ac, "cuda:0", seed=5)
        env.reset()
        if rolled:
            st = env.rollout_policy(actor, 1)
            outs.append((st["obs"][1].clone(), st["actions"][0].clone(), st["rew"][0].clone()))
            with pytest.raises(RuntimeError, match="steps must be in"):
                env.rollout_policy(actor, capi.LG_MAX_ROLL_STEPS + 1)
            bad = dict(st); bad["rew"] = st["rew"].double()
            with pytest.raises(ValueError, match="rollout storage 'rew'"):
                env._sim.rollout_policy(actor, bad, 5)
        else:
            (act, _), (obs, _, rew, _, _) = env.step_policy(actor)
            outs.append((obs.clone(), act.clone(), rew.clone()))
    for x, y in zip(*outs):
        assert float((x - y).abs().max()) < 5e-5


@pytest.mark.parametrize("self_collision", [True, False])
def test_rollout_policy_equals_sequential_fused_steps(self_collision):
    """lg_rollout_policy (T fused policy steps in ONE launch, per-step outputs in [t]-indexed rollout storage) == T lg_step_policy
    calls on the same state and noise stream: flags bit-equal, floats within the separately-compiled-instantiation band (as above)."""
    from legged_games_gym_amd.rl import ActorCritic, FusedActor
    from legged_games_gym_amd.utils.helpers import class_to_dict
    from legged_games_gym_amd.envs import task_registry
    T, N = 7, 200
    outs = []
    for rolled in (False, True):
        env_cfg, train_cfg = task_registry.get_cfgs("anymal_c_flat")
        env_cfg.asset.self_collisions = 0 if self_collision else 1
        env_cfg.env.episode_length_s = 0.1                       # 5 policy steps: time-outs, resets and finished episodes inside the segment
        from legged_games_gym_amd.utils import get_args
        args = get_args(["--task", "anymal_c_flat", "--num_envs", str(N), "--headless", "--sim_device", "cuda:0", "--rl_device", "cuda:0"])
        env, _ = task_registry.make_env("anymal_c_flat", args, env_cfg=env_cfg)
        torch.manual_seed(3)
        ac = ActorCritic(env.num_obs, env.num_obs, env.num_actions, **class_to_dict(train_cfg.policy)).to("cuda")
        actor = FusedActor(ac, "cuda:0", seed=11)
        obs, _ = env.reset()
        if rolled:
            st = env.rollout_policy(actor, T)
            rec = {k: v.clone() for k, v in st.items()}
            assert env.obs_buf.data_ptr() == st["obs"][T].data_ptr()
        else:
            rec = {"obs": [obs.clone()], "actions": [], "mean": [], "rew": [], "dones": [], "time_outs": []}
            for t in range(T):
                (act, mean), (obs, _, rew, dones, extras) = env.step_policy(actor)
                rec["obs"].append(obs.clone()); rec["actions"].append(act.clone()); rec["mean"].append(mean.clone())
                rec["rew"].append(rew.clone()); rec["dones"].append(dones.clone()); rec["time_outs"].append(extras["time_outs"].clone())
            rec = {k: torch.stack(v) for k, v in rec.items()}
        torch.cuda.synchronize()
        state = {k: env._sim.buf[k].clone() for k in ("root_states", "dof_state", "episode_length_buf", "commands", "last_actions", "sea_hidden_state",
                                                      "episode_sums", "episode_means", "rew_buf", "reset_buf", "time_out_buf", "step_counter", "feet_air_time")}
        outs.append((rec, state, env.common_step_counter))
    (ra, sa, ca), (rb, sb, cb) = outs
    assert ca == cb
    assert ra["dones"].any() and ra["time_outs"].any() and not ra["dones"].all()
    for group in (zip(ra.items(), rb.items()), zip(sa.items(), sb.items())):
        for (k, x), (_, y) in group:
            if x.dtype in (torch.bool, torch.int64, torch.int32, torch.uint8):
                assert torch.equal(x, y), k
            else:
                assert float((x - y).abs().max()) < 5e-5, (k, float((x - y).abs().max()))
    with pytest.raises(RuntimeError, match="multi-step rollout kernel"):
        env2, _ = _env("cassie", 16)
        env2.rollout_policy(actor, 3)


def test_runner_rolled_rollout_fills_the_storage_like_the_per_step_rollout():
    """The runner's one-launch rollout (lg_rollout_policy + lg_rollout_finish: the multi-step kernel writes straight into the PPO storage)
    against its per-step rollout (lg_step_policy + lg_rollout_record per step) from the same state: the same storage and statistics."""
    from legged_games_gym_amd.envs import task_registry
    from legged_games_gym_amd.utils import get_args
    out = []
    for rolled in (False, True):
        args = get_args(["--task", "anymal_c_flat", "--num_envs", "200", "--headless", "--sim_device", "cuda:0", "--rl_device", "cuda:0"])
        env_cfg, train_cfg = task_registry.get_cfgs("anymal_c_flat")
        env_cfg.env.episode_length_s = 0.2                       # episodes end inside the rollout (time-outs: the bootstrap term is exercised)
        env, _ = task_registry.make_env("anymal_c_flat", args, env_cfg=env_cfg)
        train_cfg.runner.rolled_rollout = rolled
        runner, _ = task_registry.make_alg_runner(env, "anymal_c_flat", args, train_cfg=train_cfg, log_root=None)
        del train_cfg.runner.rolled_rollout
        N, dev = env.num_envs, env.device
        sums = torch.zeros(3, device=dev)
        stats = {"cur_rew": torch.zeros(N, device=dev), "cur_len": torch.zeros(N, device=dev), "sum_rew": sums[0], "sum_len": sums[1], "count": sums[2], "_sums": sums}
        with torch.inference_mode():
            runner._rollout_steps(stats)
        torch.cuda.synchronize()
        assert bool(runner._rolled) is rolled
        st = runner.alg.storage
        out.append({k: getattr(st, k).clone() for k in ("observations", "actions", "mu", "sigma", "rewards", "dones", "actions_log_prob", "values")}
                   | {"time_outs": runner._time_outs.clone(), "sums": sums.clone(), "cur_rew": stats["cur_rew"].clone(), "cur_len": stats["cur_len"].clone(),
                      "obs_after": env.obs_buf.clone()})
    a, b = out
    assert a["dones"].any() and a["time_outs"].any() and float(a["sums"][2]) > 0
    for k in a:
        if a[k].dtype in (torch.bool, torch.uint8):
            assert torch.equal(a[k], b[k]), k
        else:
            scale = max(1.0, float(a[k].abs().max()))
            assert float((a[k] - b[k]).abs().max()) < 1e-4 * scale, (k, float((a[k] - b[k]).abs().max()))


def test_command_curriculum_tick_resamples_the_reset_envs_from_the_widened_range():
    """reset_idx applies update_command_curriculum BEFORE _resample_commands (legged_robot.py:159-176): on a curriculum tick that widens
    lin_vel_x, the envs the fused step reset re-draw their commands from the new range (same Philox block: u is unchanged, the range is not),
    the command slots of their observations follow, and extras carries max_command_x."""
    import numpy as np
    from tests import philox_np as ph
    from legged_games_gym_amd.envs import task_registry
    from legged_games_gym_amd.utils import get_args
    N = 256
    args = get_args(["--task", "anymal_c_flat", "--num_envs", str(N), "--headless", "--sim_device", "cuda:0", "--rl_device", "cuda:0"])
    env_cfg, _ = task_registry.get_cfgs("anymal_c_flat")
    env_cfg.commands.curriculum, env_cfg.commands.max_curriculum = True, 2.0
    env_cfg.commands.ranges.lin_vel_x = [-0.5, 0.5]
    env_cfg.env.episode_length_s = 0.1                        # max_episode_length = 5: every env times out on step 6, ticks at steps 5, 10, ...
    try:
        env, _ = task_registry.make_env("anymal_c_flat", args, env_cfg=env_cfg)
        env.reset()
        act = torch.zeros(N, 12, device="cuda")
        widened = False
        for s in range(1, 31):
            lo0, hi0 = env.command_ranges["lin_vel_x"]
            # make the rule fire: the reset envs' mean tracking sum must exceed 80 % of the maximum
            env.episode_sums["tracking_lin_vel"][:] = env.reward_scales["tracking_lin_vel"] * env.max_episode_length
            obs, _, _, dones, extras = env.step(act)
            torch.cuda.synchronize()
            lo1, hi1 = env.command_ranges["lin_vel_x"]
            if (lo1, hi1) != (lo0, hi0) and dones.any():
                widened = True
                ids = torch.nonzero(dones).flatten().cpu().numpy()
                u = ph.uniforms(int(env._params.seed), ids, env.common_step_counter, ph.CMD_RESET, 0)
                want = (hi1 - lo1) * u[:, 0] + lo1
                keep = np.sqrt(want ** 2 + (2.0 * u[:, 1] - 1.0) ** 2) > 0.2          # lin_vel_y range is [-1, 1] in this config
                got = env.commands[dones, 0].cpu().numpy()
                np.testing.assert_allclose(got, want * keep, atol=1e-6)
                assert np.abs(got).max() > max(abs(lo0), abs(hi0)) - 1e-3 or len(ids) < 20       # some draws lie outside the OLD range
                np.testing.assert_allclose(obs[dones, 9].cpu().numpy(), got * 2.0, atol=1e-5)
                assert abs(extras["episode"]["max_command_x"] - hi1) < 1e-9
                break
        assert widened
    finally:
        env_cfg.commands.curriculum = False
        env_cfg.commands.ranges.lin_vel_x = [-1.0, 1.0]
        env_cfg.env.episode_length_s = 20


def test_graphed_rollout_segments_continue_each_other():
    """make_graphed_rollout: each replay starts from the previous replay's obs[T] (read in place through `obs0`, copied to obs[0] by the
    kernel).  Three replays of T = 6 against 3 x 6 sequential lg_step_policy calls from the same state: same final state, and each
    segment's obs[0] is the previous segment's obs[T]."""
    from legged_games_gym_amd.rl import ActorCritic, FusedActor
    from legged_games_gym_amd.utils.helpers import class_to_dict
    from legged_games_gym_amd.envs import task_registry
    T, N, W = 6, 200, 1                       # W: make_graphed_rollout's own eager warm-up segment
    finals = []
    for rolled in (False, True):
        env, _ = _env("anymal_c_flat", N)
        _, train_cfg = task_registry.get_cfgs("anymal_c_flat")
        torch.manual_seed(3)
        ac = ActorCritic(env.num_obs, env.num_obs, env.num_actions, **class_to_dict(train_cfg.policy)).to("cuda")
        actor = FusedActor(ac, "cuda:0", seed=11)
        env.reset()
        if rolled:
            with torch.inference_mode():
                replay, st = env.make_graphed_rollout(actor, T, warmup=W)
                last = st["obs"][T].clone()
                for _ in range(3):
                    replay()
                    torch.cuda.synchronize()
                    assert torch.equal(st["obs"][0], last)
                    last = st["obs"][T].clone()
        else:
            with torch.inference_mode():
                for _ in range((W + 3) * T):
                    env.step_policy(actor)
        torch.cuda.synchronize()
        finals.append(({k: env._sim.buf[k].clone() for k in ("root_states", "dof_state", "episode_length_buf", "commands", "sea_hidden_state")},
                       env.common_step_counter, env.obs_buf.clone()))
    (a, ca, oa), (b, cb, ob) = finals
    assert ca == cb
    assert torch.equal(a["episode_length_buf"], b["episode_length_buf"])
    for k in a:
        if a[k].dtype.is_floating_point:
            assert float((a[k] - b[k]).abs().max()) < 2e-4, (k, float((a[k] - b[k]).abs().max()))
    assert float((oa - ob).abs().max()) < 2e-4
