"""Host logic with the reference's semantics: helpers.py:41-56,103-125,159-210 and task_registry.py:50-162."""
import os

import pytest

from legged_games_gym_amd.envs import configs
from legged_games_gym_amd.utils import helpers
from legged_games_gym_amd.utils.helpers import class_to_dict, get_args, get_load_path, update_cfg_from_args, update_class_from_dict


def test_get_load_path_ordering(tmp_path):
    for run in ("Mar01_10-00-00_", "Mar02_09-00-00_x", "exported"):
        os.makedirs(tmp_path / run)
    for m in ("model_50.pt", "model_100.pt", "model_1500.pt", "model_900.pt", "events.out"):
        (tmp_path / "Mar02_09-00-00_x" / m).write_text("")
    (tmp_path / "Mar01_10-00-00_" / "model_7.pt").write_text("")
    assert get_load_path(str(tmp_path)).endswith(os.path.join("Mar02_09-00-00_x", "model_1500.pt"))    # zero-padded sort key
    assert get_load_path(str(tmp_path), load_run="Mar01_10-00-00_").endswith("model_7.pt")
    assert get_load_path(str(tmp_path), checkpoint=900).endswith(os.path.join("Mar02_09-00-00_x", "model_900.pt"))
    with pytest.raises(ValueError, match="No runs"):
        get_load_path(str(tmp_path / "missing"))


def test_args_and_cfg_overrides():
    a = get_args(["--task", "cassie", "--num_envs", "12", "--seed", "9", "--max_iterations", "3", "--headless",
                  "--sim_device", "cuda:1", "--run_name", "r", "--resume"])
    assert (a.task, a.num_envs, a.seed, a.headless, a.sim_device, a.sim_device_id, a.rl_device, a.horovod) == ("cassie", 12, 9, True, "cuda:1", 1, "cuda:0", False)
    d = get_args([])
    assert d.task == "anymal_c_flat" and d.num_envs is None and d.resume is False
    env_cfg, train_cfg = configs.CassieRoughCfg(), configs.CassieRoughCfgPPO()
    update_cfg_from_args(env_cfg, train_cfg, a)
    assert env_cfg.env.num_envs == 12 and train_cfg.seed == 9 and train_cfg.runner.max_iterations == 3
    assert train_cfg.runner.run_name == "r" and train_cfg.runner.resume is True and train_cfg.runner.load_run == -1
    sp = helpers.parse_sim_params(a, {"sim": class_to_dict(env_cfg.sim)})
    assert sp.dt == 0.005 and sp.physx.contact_offset == 0.01 and sp.physx.num_position_iterations == 4 and sp.use_gpu_pipeline


def test_update_class_from_dict_and_seed():
    class A:
        x = 1

        class B:
            y = 2
    update_class_from_dict(A, {"x": 5, "B": {"y": 7}, "z": 3})
    assert A.x == 5 and A.B.y == 7 and A.z == 3
    import numpy as np, random, torch
    helpers.set_seed(123)
    r1 = (random.random(), np.random.rand(), torch.rand(1).item())
    helpers.set_seed(123)
    assert r1 == (random.random(), np.random.rand(), torch.rand(1).item())


def test_registry_surface():
    from legged_games_gym_amd.envs import task_registry, Anymal, Cassie
    assert set(task_registry.task_classes) == {"anymal_c_rough", "anymal_c_flat", "anymal_b", "a1", "cassie"}
    assert task_registry.get_task_class("anymal_c_flat") is Anymal and task_registry.get_task_class("cassie") is Cassie
    env_cfg, train_cfg = task_registry.get_cfgs("anymal_c_flat")
    assert env_cfg.seed == train_cfg.seed == 1 and train_cfg.runner.experiment_name == "flat_anymal_c" and train_cfg.runner.max_iterations == 300
    assert class_to_dict(train_cfg)["policy"]["actor_hidden_dims"] == [128, 64, 32]
    with pytest.raises(ValueError, match="not registered"):
        task_registry.make_env("nope", get_args(["--headless"]))


def test_reward_scale_without_function_raises_like_the_reference():
    from legged_games_gym_amd.utils import packing
    from legged_games_gym_amd.utils.model_compiler import load_model
    cfg = configs.AnymalCFlatCfg()
    cfg.rewards.scales.feet_stumble = -1.0                    # quirk Q3: no _reward_feet_stumble
    with pytest.raises(AttributeError, match="_reward_feet_stumble"):
        packing.build_params(cfg, load_model(cfg.asset.file), 0.005, 8, 1)


def test_export_policy_as_jit_round_trip(tmp_path):
    import torch
    from legged_games_gym_amd.rl import ActorCritic
    from legged_games_gym_amd.utils.helpers import export_policy_as_jit
    torch.manual_seed(0)
    ac = ActorCritic(48, 48, 12, actor_hidden_dims=[32, 16], critic_hidden_dims=[32, 16], activation="elu")
    target = export_policy_as_jit(ac, str(tmp_path / "exported" / "policies"))
    assert target.endswith("policy_1.pt")
    jit = torch.jit.load(target)                    # a file this test just wrote
    x = torch.randn(5, 48)
    assert torch.allclose(jit(x), ac.act_inference(x), atol=1e-6)


def test_logger_surface_and_headless_plot(tmp_path, capsys):
    import numpy as np
    import torch
    from legged_games_gym_amd.utils.logger import Logger
    lg = Logger(0.02, out_dir=str(tmp_path))
    for i in range(20):
        lg.log_states({"dof_pos": 0.1 * i, "dof_pos_target": 0.1 * i + 0.01, "dof_vel": 1.0, "dof_torque": 2.0, "base_vel_x": 0.5, "command_x": 0.5,
                       "contact_forces_z": np.array([1.0, 2.0, 3.0, 4.0])})
    lg.log_rewards({"rew_tracking_lin_vel": torch.tensor(0.5), "terrain_level": torch.tensor(3.0)}, 4)
    lg.log_rewards({"rew_tracking_lin_vel": torch.tensor(1.0)}, 4)
    lg.print_rewards()
    out = capsys.readouterr().out
    assert "rew_tracking_lin_vel: 0.75" in out and "terrain_level" not in out and "Total number of episodes: 8" in out
    path = lg.plot_states()
    assert os.path.isfile(path) and os.path.getsize(path) > 100
    lg.reset()
    assert not lg.state_log and not lg.rew_log


def test_runner_scalar_log_uses_rsl_rl_tag_names(tmp_path):
    """OnPolicyRunner._log_scalars: progress.csv with rsl_rl's TensorBoard tags as columns (tensorboard itself is optional)."""
    import types
    import torch
    from legged_games_gym_amd.rl.runner import OnPolicyRunner
    r = OnPolicyRunner.__new__(OnPolicyRunner)
    r.log_dir, r.writer, r.device, r.tot_timesteps = str(tmp_path), None, "cpu", 98304
    r.alg = types.SimpleNamespace(learning_rate=1e-3, actor_critic=types.SimpleNamespace(std=torch.ones(12)))
    r.env = types.SimpleNamespace(extras={"episode": {"rew_tracking_lin_vel": torch.tensor(0.5), "rew_torques": -0.25}})
    r._log_scalars(0, 1000, 0.1, 0.2, 0.3, -0.01, float("nan"), 12.0)
    r._log_scalars(1, 2000, 0.1, 0.2, 0.2, -0.02, 1.5, 13.0)
    lines = open(tmp_path / "progress.csv").read().strip().splitlines()
    cols = lines[0].split(",")
    for tag in ("Loss/value_function", "Loss/surrogate", "Loss/learning_rate", "Policy/mean_noise_std", "Perf/total_fps",
                "Train/mean_reward", "Train/mean_episode_length", "Episode/rew_torques", "Episode/rew_tracking_lin_vel"):
        assert tag in cols
    assert len(lines) == 3 and lines[2].split(",")[cols.index("Train/mean_reward")] == "1.5"
    assert lines[2].split(",")[cols.index("Episode/rew_torques")] == "-0.25"


def test_no_task_is_flagged_experimental_and_custom_actuator_files_are_refused(tmp_path):
    """`make_env` warns for tasks registered in `task_registry.experimental` (a1 was, until round 3: it trains with the reference's
    PPO defaults since its ground stiffness is set per robot, packing.ROBOT_ENGINE_OPTIONS); a user's own TorchScript actuator file
    must not be silently replaced by the bundled net."""
    import numpy as np
    import pytest
    from legged_games_gym_amd.envs import task_registry
    from legged_games_gym_amd.envs.configs import A1RoughCfg, AnymalCFlatCfg
    from legged_games_gym_amd.utils import packing
    from legged_games_gym_amd.utils.model_compiler import load_model
    assert not task_registry.experimental
    a1, anymal = A1RoughCfg(), AnymalCFlatCfg()
    pa, _ = packing.build_params(a1, load_model(a1.asset.file), a1.sim.dt, 4, 1)
    pb, _ = packing.build_params(anymal, load_model(anymal.asset.file), anymal.sim.dt, 4, 1)
    assert abs(pa.contact_stiffness - 5.0e4) < 1e-3 and abs(pb.contact_stiffness - 1.0e6) < 1e-3
    assert pa.contact_damping == pb.contact_damping and pa.friction_damping == pb.friction_damping
    assert packing.load_actuator_weights("/somewhere/resources/actuator_nets/anydrive_v3_lstm.pt").size == 972
    assert packing.load_actuator_weights(None).size == 972
    with pytest.raises(ValueError, match="compile_models"):
        packing.load_actuator_weights(str(tmp_path / "my_own_net.pt"))
    blob = tmp_path / "net.f32"
    np.arange(972, dtype="<f4").tofile(blob)
    assert packing.load_actuator_weights(str(blob))[5] == 5.0


def test_committed_pmc_summary_has_what_bench_reads():
    """bench.py takes `roofline.traffic` / `mfma_busy_pct` from profiles/r03_pmc_summary.json (tools/collect_r03.sh + pmc_summary_r03.py):
    the three single-GPU BASELINE configs must be there with the keys it reads, at the env counts it runs them with."""
    import json
    import os
    repo = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    pm = json.load(open(bench.PMC_SUMMARY))["tasks"]
    for task, envs in (("anymal_c_flat", 4096), ("anymal_c_rough", 4096), ("cassie", 8192)):
        e = pm[task]
        assert e["envs_per_gpu"] == envs
        k = e["k_step"]
        assert k["steps_per_launch"] >= 1 and k["traffic_bytes_per_launch"] > 1e6 and 0.0 <= k["mfma_busy_frac_of_busy_cycles"] < 1.0
        assert k["algorithmic_bytes_per_policy_step"] == bench.BYTES_PER_ENV_STEP[task] * envs
    assert pm["anymal_c_flat"]["k_step"]["steps_per_launch"] == 20 and "true, true>" in pm["anymal_c_flat"]["k_step"]["kernel"]     # the multi-step kernel


def test_committed_kernel_resource_table_keeps_the_headline_kernels_out_of_scratch():
    """csrc/kernel_resources.txt (written by __graft_entry__.build() from hipcc's resource remarks) is the evidence DESIGN.md quotes for
    register use.  Guard what the measurements rest on: the rollout kernel of the headline (ANYmal-C flat, actuator net, fused actor,
    self-collision, multi-step) and the wide learner / actor kernels have no spilled register and no private segment; no step kernel has a
    private segment beyond the 1 KB that build() itself refuses (kernel arguments copied to scratch); every kernel is listed once."""
    import os, re
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "legged_games_gym_amd", "csrc", "kernel_resources.txt")
    rows = {}
    for line in open(path):
        m = re.match(r"(\S+)\s+VGPRs (\d+)\s+AGPRs (\d+)\s+spill (\d+)\s+scratch (\d+)\s+LDS (\d+)\s+occupancy (\d+)", line)
        if m:
            assert m.group(1) not in rows, m.group(1)
            rows[m.group(1)] = dict(zip(("vgpr", "agpr", "spill", "scratch", "lds", "occ"), map(int, m.groups()[1:])))
    headline = "_Z6k_stepI12AnymalTraitsLb1ELb0ELb1ELi4ELb1ELb1EEv5KArgs"           # <Anymal, NET, plane, POL, 4 waves, SC, ROLL>
    assert headline in rows and rows[headline]["spill"] == 0 and rows[headline]["scratch"] == 0, rows.get(headline)
    assert rows[headline]["lds"] <= 160 * 1024
    for name, r in rows.items():
        if "k_step" in name:
            assert r["scratch"] <= 1024, (name, r)
        if any(k in name for k in ("k_mlp_chain_fwd64", "k_policy_act_wide", "k_gemm_wide_bf16x3", "k_policy_act")):
            assert r["spill"] == 0 and r["scratch"] == 0, (name, r)
    cassie = "_Z6k_stepI12CassieTraitsLb0ELb1ELb0ELi4ELb0ELb0EEv5KArgs"             # config 5: fewer spilled registers than at the start of round 3 (83)
    assert rows[cassie]["spill"] <= 60, rows[cassie]
