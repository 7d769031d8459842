"""G5: the reset / RNG half of the torch side, pinned by the reference's OWN code.

tests/golden/post_physics_full_<task>.npz hold inputs and outputs of the reference's whole ``post_physics_step``
(legged_robot.py:106-137) -- ``_post_physics_step_callback`` with ``_resample_commands`` (:347-369, incl. the ``> 0.2`` zeroing)
and ``_push_robots`` (:438-444), ``reset_idx`` (:147-191) with ``_update_terrain_curriculum`` (:446-469: move up / move down /
solved-last-level / clip at 0), ``_reset_dofs`` (:397-412), ``_reset_root_states`` (:414-436), and ``compute_observations`` with
noise (:212-230) -- ast-extracted from /root/reference and executed by tools/make_golden.py:g5 with every random draw answered
from the build's counter-based stream (tests/philox_np.py: numpy Philox4x32-10, Random123 KATs in tests/test_philox.py).
``reset_idx_<task>.npz``: the stand-alone ``reset_idx(env_ids)``; ``command_curriculum.npz``: ``update_command_curriculum`` (:471-483).

Here: the CPU oracle against those fixtures (ints bit-exact, floats 1e-6); tests/test_gpu_parity.py runs the HIP path against
the same files."""
import os

import numpy as np
import pytest

from tests.common import make_setup, golden_tweak, TASK_CFG
from tests.test_oracle_torch_side import load_fixture_inputs
from oracle.oracle import OracleSim

FULL = {"anymal_c_flat": "anymal_c_flat", "anymal_c_rough": "anymal_c_rough", "cassie": "cassie"}
FLOAT_TOL = dict(rtol=1e-6, atol=1e-6)


def full_setup(g, task, N, decimation=0):
    """Params / model / terrain of a G5 fixture: the same tweak the generator applied to the reference's config classes."""
    from legged_games_gym_amd.utils.terrain import Terrain
    kind = str(g["kind"])
    tw = golden_tweak(kind)
    terr = None
    if "height_samples" in g.files:
        mine = TASK_CFG[task](); tw(mine)
        np.random.seed(3)
        terr = Terrain(mine.terrain, N)
        terr.height_field_raw[:] = g["height_samples"]
        terr.env_origins[:] = g["terrain_origins"]
    cfg, robot, p, names, model, w = make_setup(task, N, seed=int(g["seed"]), plane=(terr is None), terrain=terr, tweak=tw)
    p.decimation = decimation
    return terr, cfg, robot, p, names, model, w


def load_full_inputs(put, g, terr):
    put("env_origins", g["in_env_origins"])
    put("episode_sums", g["in_episode_sums"])
    if terr is not None:
        put("terrain_levels", g["in_terrain_levels"])
        put("terrain_types", g["in_terrain_types"])


def check_full_outputs(get, g, names, cfg, has_net):
    """Shared by the oracle test here and the HIP test (tests/test_gpu_parity.py): ``get(name)`` -> numpy array."""
    rs = g["reset_buf"].astype(bool)
    assert rs.sum() >= 5 and (~rs).sum() >= 20
    # integers / booleans: bit-exact
    np.testing.assert_array_equal(get("reset_buf").astype(bool), rs)
    np.testing.assert_array_equal(get("time_out_buf").astype(bool), g["time_out_buf"].astype(bool))
    np.testing.assert_array_equal(get("time_out_buf").astype(bool), g["extras_time_outs"].astype(bool))     # extras["time_outs"] (:189-191)
    np.testing.assert_array_equal(get("episode_length_buf"), g["episode_length_buf"])
    np.testing.assert_array_equal(get("last_contacts").astype(bool), g["last_contacts"].astype(bool))
    if "terrain_levels" in g.files:
        np.testing.assert_array_equal(get("terrain_levels"), g["terrain_levels"])
        delta = (g["terrain_levels"] - g["in_terrain_levels"])[rs]
        if cfg.terrain.curriculum:                         # up, down, unchanged, and the solved-last-level draw (a jump below -1)
            assert (delta == 1).any() and (delta == -1).any() and (delta == 0).any() and (delta < -1).any()
        else:
            assert not delta.any()
    # floats: 1e-6 (the build's urange() is one fma, torch's two roundings)
    for k in ("root_states", "dof_state", "commands", "last_actions", "last_dof_vel", "last_root_vel", "feet_air_time", "env_origins",
              "base_lin_vel", "base_ang_vel", "projected_gravity"):
        np.testing.assert_allclose(get(k).reshape(g[k].shape), g[k], err_msg=k, **FLOAT_TOL)
    np.testing.assert_allclose(get("rew_buf"), g["rew_buf"], rtol=2e-5, atol=3e-6)
    np.testing.assert_allclose(get("episode_sums"), g["episode_sums"], rtol=2e-5, atol=3e-6)
    assert np.all(get("episode_sums")[:, rs] == 0.0)
    if g["measured_heights"].size:
        mism = np.abs(get("measured_heights") - g["measured_heights"]) > 1e-6
        assert mism.mean() < 2e-3, mism.mean()            # a point within 1 ulp of a cell edge may truncate differently
        ok = ~mism.any(axis=1)
    else:
        ok = np.ones(rs.shape, bool)
    # observations WITH noise, for the reset envs too (built from the freshly written state, stale base-frame quantities: Q7)
    np.testing.assert_allclose(get("obs_buf")[ok], g["obs_buf"][ok], rtol=1e-5, atol=3e-6)
    assert np.abs(g["obs_buf"][rs][:, 24:36]).max() <= 1.5 * 0.05 + 1e-6       # dof_vel of a reset env is exactly 0: only noise is left
    # extras["episode"] (:179-188)
    np.testing.assert_allclose(get("episode_means")[:len(names)], g["episode_means"], rtol=2e-5, atol=2e-6)
    if float(g["terrain_level_mean"]) >= 0:
        assert abs(float(get("episode_means")[len(names)]) - float(g["terrain_level_mean"])) < 1e-5
    if has_net:                                            # anymal.py:56-60
        for k in ("sea_hidden_state", "sea_cell_state"):
            assert np.all(get(k).reshape(2, rs.size, 12, 8)[:, rs] == 0.0)
    # the fixture did exercise what it claims
    log = g["draw_log"]
    assert {1, 2, 3, 4, 5} <= set(log[:, 0].tolist())       # CMD_STEP, CMD_RESET, DOF, ROOT, PUSH


@pytest.mark.parametrize("task", list(FULL))
def test_full_post_physics_matches_reference_fixture(task, oracle_lib, golden_dir):
    g = np.load(os.path.join(golden_dir, f"post_physics_full_{task}.npz"))
    N = g["in_root_states"].shape[0]
    terr, cfg, robot, p, names, model, w = full_setup(g, task, N)
    assert names == [str(n) for n in g["reward_names"]]
    o = OracleSim(p, model, robot, w)
    if terr is not None:
        o.set_terrain(terr.heightsamples, terr.env_origins)
    load_fixture_inputs(o, g)
    load_full_inputs(lambda k, v: o.buf[k].__setitem__(Ellipsis, np.asarray(v).astype(o.buf[k].dtype).reshape(o.buf[k].shape)), g, terr)
    o.step(g["in_actions"], int(g["step"]))
    check_full_outputs(lambda k: o.buf[k], g, names, cfg, w is not None)


@pytest.mark.parametrize("task", ["anymal_c_flat", "anymal_c_rough", "cassie"])
def test_reset_idx_matches_reference_fixture(task, oracle_lib, golden_dir):
    g = np.load(os.path.join(golden_dir, f"reset_idx_{task}.npz"))
    N = g["in_root_states"].shape[0]
    terr, cfg, robot, p, names, model, w = full_setup(g, task, N)
    o = OracleSim(p, model, robot, w)
    if terr is not None:
        o.set_terrain(terr.heightsamples, terr.env_origins)
    load_fixture_inputs(o, g)
    load_full_inputs(lambda k, v: o.buf[k].__setitem__(Ellipsis, np.asarray(v).astype(o.buf[k].dtype).reshape(o.buf[k].shape)), g, terr)
    o.reset_idx(g["env_ids"], int(g["step"]))
    check_reset_idx_outputs(lambda k: o.buf[k], g, names)


def check_reset_idx_outputs(get, g, names):
    ids = g["env_ids"]
    np.testing.assert_array_equal(get("episode_length_buf"), g["episode_length_buf"])
    assert np.all(get("reset_buf")[ids] != 0)
    if "terrain_levels" in g.files:
        np.testing.assert_array_equal(get("terrain_levels"), g["terrain_levels"])
    for k in ("root_states", "dof_state", "commands", "last_actions", "last_dof_vel", "feet_air_time", "env_origins"):
        np.testing.assert_allclose(get(k).reshape(g[k].shape), g[k], err_msg=k, **FLOAT_TOL)
    np.testing.assert_allclose(get("episode_sums"), g["episode_sums"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(get("episode_means")[:len(names)], g["episode_means"], rtol=2e-5, atol=2e-6)
    if float(g["terrain_level_mean"]) >= 0:
        assert abs(float(get("episode_means")[len(names)]) - float(g["terrain_level_mean"])) < 1e-5
    keep = np.setdiff1d(np.arange(g["in_root_states"].shape[0]), ids)
    np.testing.assert_array_equal(get("root_states")[keep], g["in_root_states"][keep])        # the others are untouched


def test_fixtures_exercise_the_small_command_rule(golden_dir):
    """``commands[env_ids, :2] *= norm > 0.2`` (:368-369) fired somewhere in the reference-executed fixtures (it does for ~3 % of draws)."""
    zeroed = 0
    for f in ("post_physics_full_anymal_c_flat", "post_physics_full_anymal_c_rough", "post_physics_full_cassie", "reset_idx_anymal_c_flat", "reset_idx_anymal_c_rough"):
        g = np.load(os.path.join(golden_dir, f + ".npz"))
        zeroed += int(((np.linalg.norm(g["commands"][:, :2], axis=1) == 0.0) & (np.linalg.norm(g["in_commands"][:, :2], axis=1) > 0.0)).sum())
    assert zeroed >= 2, zeroed


def test_command_curriculum_rule_matches_reference_fixture(golden_dir):
    """update_command_curriculum (:471-483) against the host rule the env applies between fused steps."""
    from legged_games_gym_amd.envs.base.legged_robot import command_curriculum_update
    g = np.load(os.path.join(golden_dir, "command_curriculum.npz"))
    changed = 0
    for frac, lo, hi, want_lo, want_hi in g["rows"]:
        mean_sum = frac * float(g["scale_dt"]) * float(g["max_episode_length"])
        got = command_curriculum_update(mean_sum, float(g["max_episode_length"]), float(g["scale_dt"]), [lo, hi], float(g["max_curriculum"]))
        assert abs(got[0] - want_lo) < 1e-12 and abs(got[1] - want_hi) < 1e-12, (frac, lo, hi, got)
        changed += (want_lo, want_hi) != (lo, hi)
    assert changed >= 2
