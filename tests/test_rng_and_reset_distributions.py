"""The RNG-consuming pieces of the path (command resampling :347-369, reset :397-436, pushes :438-444, observation
noise :229-230) cannot be bit-compared with torch's global stream; they are checked as distributions on the oracle."""
import numpy as np

from tests.common import make_setup, grid_origins
from oracle.oracle import OracleSim


def test_reset_and_command_distributions(oracle_lib):
    N = 4096
    cfg, robot, p, names, model, w = make_setup("anymal_c_flat", N)
    o = OracleSim(p, model, robot, w, threads=4)
    org = grid_origins(N)
    o.buf["env_origins"][:] = org
    o.reset_idx(np.arange(N, dtype=np.int32), 0)
    root, q = o.buf["root_states"], o.dof_pos
    np.testing.assert_allclose(root[:, :2], org[:, :2], atol=1e-6)            # plane: no xy jitter on reset (:427-429)
    np.testing.assert_allclose(root[:, 2], 0.6, atol=1e-6)
    np.testing.assert_array_equal(root[:, 3:7], np.tile([0, 0, 0, 1], (N, 1)))
    v = root[:, 7:13]
    assert v.min() >= -0.5 and v.max() <= 0.5 and abs(v.mean()) < 0.01 and abs(v.std() - 1 / np.sqrt(12)) < 0.01
    q0 = np.array(list(p.default_dof_pos)[:12])
    nz = np.abs(q0) > 1e-6
    r = q[:, nz] / q0[nz]
    assert r.min() >= 0.5 and r.max() <= 1.5 and abs(r.mean() - 1.0) < 0.01 and abs(r.std() - 1 / np.sqrt(12)) < 0.01
    assert np.all(q[:, ~nz] == 0) and np.all(o.dof_vel == 0)
    cmd = o.buf["commands"]
    small = np.linalg.norm(cmd[:, :2], axis=1) <= 0.2
    assert np.all(cmd[small, :2] == 0)                                         # small commands zeroed (:369)
    big = cmd[~small]
    assert big[:, 0].min() >= -1 and big[:, 0].max() <= 1 and abs(cmd[:, 2]).max() <= 1.5 and abs(cmd[:, 2].std() - 3 / np.sqrt(12)) < 0.03
    # independent streams: dof draws are uncorrelated with root-velocity draws
    assert abs(np.corrcoef(r[:, 0], v[:, 0])[0, 1]) < 0.05


def test_noise_push_and_resample_schedule(oracle_lib):
    N = 2048

    def tweak(c):
        c.domain_rand.push_robots = True
    cfg, robot, p, names, model, w = make_setup("anymal_c_flat", N, tweak=tweak)
    p.decimation = 0
    o = OracleSim(p, model, robot, w, threads=4)
    o.reset_idx(np.arange(N, dtype=np.int32), 0)
    o.buf["root_states"][:, 7:13] = 0
    o.buf["dof_state"][:, 1] = 0
    o.buf["episode_length_buf"][:] = 198
    cmd0 = o.buf["commands"].copy()
    act = np.zeros((N, 12), np.float32)
    o.step(act, 7)                                         # ep_len 199: no resample, no push
    assert np.array_equal(o.buf["commands"], cmd0) and np.all(o.buf["root_states"][:, 7:9] == 0)
    noise = o.buf["obs_buf"].copy()
    # base lin vel is exactly 0 -> obs[:, 0:3] is pure noise of scale 0.1*2.0 ; commands / actions carry none
    assert np.abs(noise[:, 0:3]).max() <= 0.2 + 1e-6 and abs(noise[:, 0:3].std() - 0.2 / np.sqrt(3)) < 0.005
    assert np.abs(noise[:, 24:36]).max() <= 1.5 * 0.05 + 1e-6 and np.all(noise[:, 36:48] == 0)
    np.testing.assert_allclose(noise[:, 9:12], cmd0[:, :3] * np.array([2, 2, 0.25], np.float32), rtol=1e-6)
    o.step(act, 8)                                         # ep_len 200 = resampling interval (flat: 4 s / 0.02 s)
    assert (o.buf["commands"][:, :3] != cmd0[:, :3]).any(axis=1).mean() > 0.95
    assert not np.array_equal(noise, o.buf["obs_buf"])    # fresh noise every step
    o.step(act, 750)                                       # push step (15 s / 0.02 s)
    push = o.buf["root_states"][:, 7:9]
    assert np.abs(push).max() <= 1.0 and abs(push.std() - 2 / np.sqrt(12)) < 0.03 and np.all(o.buf["root_states"][:, 9] == 0)
