"""-m gpu: self-collision (asset.self_collisions = 0, the anymal_c_flat configuration: anymal_c_flat_config.py:42) of the HIP step
against the CPU oracle, and its size-independent invariant: links of one robot never end up overlapping by more than
contact_offset, whatever the actions."""
import numpy as np
import pytest
import torch

from tests.common import make_setup, grid_origins, randomize_env_params, min_self_clearance
from tests.test_oracle_physics import _crossing_pose

pytestmark = pytest.mark.gpu


def _pair(task, N, on=True, tweak=None):
    from oracle.oracle import OracleSim
    from legged_games_gym_amd.device_sim import DeviceSim

    def tw(cfg):
        cfg.asset.self_collisions = 0 if on else 1
        if tweak:
            tweak(cfg)
    cfg, robot, p, names, model, w = make_setup(task, N, tweak=tw)
    assert p.self_collision == int(on)
    o = OracleSim(p, model, robot, w, threads=8)
    d = DeviceSim(p, model, robot, torch.device("cuda:0"), w)
    return cfg, robot, p, o, d


def _put(o, d, name, val):
    o.buf[name][...] = np.asarray(val).astype(o.buf[name].dtype).reshape(o.buf[name].shape)
    d.buf[name].copy_(torch.from_numpy(o.buf[name]).to(d.buf[name].dtype).view(d.buf[name].shape))


def _get(d, name):
    torch.cuda.synchronize()
    t = d.buf[name]
    return (t.to(torch.uint8) if t.dtype == torch.bool else t).cpu().numpy()


from tests.test_oracle_physics import adversarial_actions as _adversarial_actions  # noqa: E402


@pytest.mark.parametrize("task", ["anymal_c_flat", "a1"])
def test_substep_parity_with_crossed_legs(task):
    """One 5 ms sub-step from poses whose front legs overlap by up to ~2 cm (plus random base / joint velocities): the kernel's
    pair detection, Jacobi-coupled implicit contact and force export against the oracle's, every env."""
    N = 256
    cfg, robot, p, o, d = _pair(task, N)
    fr, dm = randomize_env_params(N, 3)
    _put(o, d, "env_origins", grid_origins(N)); _put(o, d, "friction_coeffs", fr); _put(o, d, "base_mass_delta", dm)
    ids = np.arange(N, dtype=np.int32)
    o.reset_idx(ids, 0); d.reset_idx(torch.from_numpy(ids), 0)
    q0 = np.array(list(p.default_dof_pos)[:12], np.float64)
    names = list(robot.dof_names)
    haa = [i for i, n in enumerate(names) if n.endswith("HAA") or n.endswith("hip_joint")]
    rng = np.random.default_rng(1)
    if task == "anymal_c_flat":
        qx, sign, (lf, rf) = _crossing_pose(robot, q0)
    else:                                        # A1: find the inward direction of the front abduction joints the same way
        lf, rf = names.index("FL_hip_joint"), names.index("FR_hip_joint")
        sign = {}
        for j in (lf, rf):
            qa, qb = q0.copy(), q0.copy(); qa[j] += 0.4; qb[j] -= 0.4
            sign[j] = 1.0 if min_self_clearance(robot, qa) < min_self_clearance(robot, qb) else -1.0
        qx = q0.copy()
        for ang in np.linspace(0, 1.4, 141):
            qx = q0.copy(); qx[lf] += sign[lf] * ang; qx[rf] += sign[rf] * ang
            if min_self_clearance(robot, qx) < -0.005:
                break
    dof = o.buf["dof_state"].reshape(N, 12, 2).copy()
    scale = rng.uniform(0.6, 1.15, N)            # from clearly apart to ~2 cm of overlap
    dof[:, :, 0] = q0 + (qx - q0)[None, :] * scale[:, None] + rng.normal(0, 0.02, (N, 12))
    dof[:, :, 1] = rng.normal(0, 1.0, (N, 12))
    _put(o, d, "dof_state", dof.reshape(-1, 2))
    root = o.buf["root_states"].copy()
    root[:, 2] = 1.5                             # airborne: ground contact is not the subject here
    root[:, 7:13] = rng.normal(0, 0.3, (N, 6))
    _put(o, d, "root_states", root)
    tau = rng.normal(0, 10.0, (N, 12)).astype(np.float32)
    qd0 = dof[..., 1].copy()
    o.physics_substep(tau, True); d.physics_substep(torch.from_numpy(tau), True)
    cf_o, cf_d = o.buf["contact_forces"], _get(d, "contact_forces")
    touching = np.abs(cf_o).sum(axis=(1, 2)) > 1.0
    assert 0.2 < touching.mean() < 0.95                                   # both populations present
    assert np.array_equal(touching, np.abs(cf_d).sum(axis=(1, 2)) > 1.0)  # same contact decisions
    q_o, q_d = o.buf["dof_state"].reshape(N, 12, 2), _get(d, "dof_state").reshape(N, 12, 2)
    dqd = np.abs(q_o[..., 1] - qd0).max(axis=1)                           # centimetres of overlap at 1e6 N/m: hundreds of rad/s^2
    err_v = np.abs(q_o[..., 1] - q_d[..., 1]).max(axis=1)
    rel = 5e-4 if task == "a1" else 2e-4                                 # A1's 60 g feet / 170 g calves: larger accelerations per newton
    ratio = err_v / (1.0 + dqd)
    assert np.quantile(ratio, 0.99) <= rel and ratio.max() <= 5 * rel, (float(np.quantile(ratio, 0.99)), float(ratio.max()))
    assert np.abs(q_o[..., 0] - q_d[..., 0]).max() < 2e-5
    assert np.abs(o.buf["root_states"] - _get(d, "root_states")).max() < 1e-3
    f_scale = max(1.0, float(np.abs(cf_o).max()))
    assert np.abs(cf_o - cf_d).max() < 1e-3 * f_scale, (np.abs(cf_o - cf_d).max(), f_scale)
    assert np.abs(cf_o[touching]).max() > 10.0      # (exported value = estimate of the force actually exchanged, not the 10 kN bias)


def test_policy_step_parity_with_self_collision():
    """The fused policy step (actuator net, 4 sub-steps, post-physics) with self-collision on, standing on the plane, legs
    driven into each other: termination / collision-penalty inputs include the self-collision forces on both sides."""
    N = 256
    cfg, robot, p, o, d = _pair("anymal_c_flat", N)
    fr, dm = randomize_env_params(N, 3)
    _put(o, d, "env_origins", grid_origins(N)); _put(o, d, "friction_coeffs", fr); _put(o, d, "base_mass_delta", dm)
    ids = np.arange(N, dtype=np.int32)
    o.reset_idx(ids, 0); d.reset_idx(torch.from_numpy(ids), 0)
    act = _adversarial_actions(robot, p, N)
    z = torch.from_numpy(act)
    for it in range(1, 13):                      # drive the legs together on the device, then hand the state to the oracle
        d.step(z.cuda(), it)
    torch.cuda.synchronize()
    for name, dst in o.buf.items():
        t = d.buf[name]
        dst[...] = (t.to(torch.uint8) if t.dtype == torch.bool else t).cpu().numpy().astype(dst.dtype).reshape(dst.shape)
    o.step(act, 13); d.step(z.cuda(), 13)
    assert np.array_equal(o.buf["reset_buf"], _get(d, "reset_buf"))
    q_o, q_d = o.buf["dof_state"].reshape(N, 12, 2), _get(d, "dof_state").reshape(N, 12, 2)
    e_pos = np.abs(q_o[..., 0] - q_d[..., 0]).max(axis=1)
    assert np.quantile(e_pos, 0.98) < 1e-3 and np.median(e_pos) < 5e-5, (np.quantile(e_pos, 0.98), np.median(e_pos))
    bulk = e_pos < 1e-3
    assert np.abs(o.buf["rew_buf"] - _get(d, "rew_buf"))[bulk].max() < 2e-3
    bn = list(robot.body_names)
    legs = [i for i, n in enumerate(bn) if "SHANK" in n or "THIGH" in n]
    lateral = np.abs(o.buf["contact_forces"][:, legs, 1]).max(axis=1)
    assert (lateral > 20.0).mean() > 0.3         # leg-against-leg forces are horizontal: they are in the exported tensor


@pytest.mark.parametrize("N", [4096])
def test_no_interpenetration_under_adversarial_actions_full_size(N):
    """BASELINE.json configs[1] size; same protocol as the oracle's test (tests/test_oracle_physics.py): 50 policy steps of
    saturating inward targets (legs thrash into each other at up to the 20 rad/s joint speed limit), then 30 steps of the same
    targets scaled down (pressed together, quasi-static).  With self-collision (the flat config's setting) overlaps beyond
    contact_offset = 1 cm are rare and below 3 cm while thrashing, and millimetres once static; without it the legs sit ~10 cm
    inside each other."""
    from legged_games_gym_amd.device_sim import DeviceSim
    res = {}
    sample = np.arange(0, N, 16)                                  # 256 envs per checkpoint (float64 brute-force capsule distances)
    for on in (True, False):
        cfg, robot, p, names, model, w = make_setup("anymal_c_flat", N, tweak=lambda c: setattr(c.asset, "self_collisions", 0 if on else 1))
        d = DeviceSim(p, model, robot, torch.device("cuda:0"), w)
        fr, dm = randomize_env_params(N, 5)
        d.buf["env_origins"].copy_(torch.from_numpy(grid_origins(N)))
        d.buf["friction_coeffs"].copy_(torch.from_numpy(fr)); d.buf["base_mass_delta"].copy_(torch.from_numpy(dm))
        d.reset_idx(torch.arange(N, dtype=torch.int32), 0)
        act = torch.from_numpy(_adversarial_actions(robot, p, N, seed=1)).cuda()
        clear = []
        for it in range(1, 81):
            d.step(act if it <= 50 else 0.35 * act, it)
            if it in (10, 20, 30, 40, 50, 76, 80):
                q = _get(d, "dof_state").reshape(N, 12, 2)[..., 0].astype(np.float64)
                clear.append([min_self_clearance(robot, q[e], samples=17) for e in sample])
        assert np.isfinite(_get(d, "root_states")).all() and np.isfinite(_get(d, "obs_buf")).all()
        res[on] = np.array(clear)
        print("self-collision", "on: " if on else "off:", "min clearance per checkpoint", np.round(res[on].min(axis=1), 4),
              "share of envs beyond contact_offset", np.round((res[on] < -0.01).mean(axis=1), 3))
    thrash_on, thrash_off = res[True][:5], res[False][:5]
    assert thrash_off.min() < -0.05 and (thrash_off < -0.01).mean() > 0.25
    assert thrash_on.min() > -0.03 and (thrash_on < -0.01).mean() < 0.05, (thrash_on.min(), (thrash_on < -0.01).mean())
    # pressed together: all but a handful of the 512 samples (a robot that fell over and lies on its legs) within millimetres
    assert np.quantile(res[True][-2:], 0.01) > -0.005 and res[True][-2:].min() > -0.03, (np.quantile(res[True][-2:], 0.01), res[True][-2:].min())
    assert res[False][-2:].min() < -0.02
