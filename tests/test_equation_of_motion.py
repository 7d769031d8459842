"""The rigid-body half (H4: what PhysX does for the reference) tied to the published equations of motion.

PhysX itself is absent and nothing in the reference pins a trajectory, so contact stays "defined here" -- but the DYNAMICS CORE
of a sub-step is not a matter of definition: for an airborne robot one sub-step must satisfy Newton-Euler.

* ``test_substep_satisfies_newton_euler_*``: random airborne state and joint torques on all four robots, one sub-step of the
  oracle (CPU) / ``lg_physics_substep`` (-m gpu); the accelerations the step produced (velocity differences / dt) go into an
  independent float64 recursive Newton-Euler inverse dynamics (tests/eom.py, written from the textbook formulation, nothing
  shared with the engine): it must return the applied torques on the joints and a zero external wrench on the floating base.
  Tolerance 2e-5 of the magnitude of the terms involved (fp32 velocities differenced over dt = 5 ms).
* ``test_single_joint_oscillation_period_*``: SURVEY 8(c)'s pendulum-period known answer.  A floating base has no fixed
  pivot (in free fall a leg does not swing), so the restoring moment m g l sin(q) of the pendulum is supplied by the reference's
  own P-controller (legged_robot.py:384-386: tau = Kp (q0 - q)) in zero gravity under a heavy trunk: one leg joint is a
  torsional pendulum with T = 2 pi sqrt(I / Kp), I = composite inertia of the leg about the joint axis from float64 forward
  kinematics.  Checks inertia tables, the articulated-body recursion and the integrator's time scale in one number.
"""
import numpy as np
import pytest

from tests.common import make_setup, grid_origins
from tests import eom

ROBOTS = ["anymal_c_flat", "cassie", "a1", "anymal_b"]


def no_rng(cfg):
    cfg.noise.add_noise = False
    cfg.domain_rand.push_robots = False
    cfg.asset.self_collisions = 1                   # random joint angles put links inside each other; contact is not under test


def random_airborne_state(robot, p, N, seed):
    rng = np.random.default_rng(seed)
    n = robot.num_dof
    q0 = np.array(list(p.default_dof_pos)[:n])
    lo, hi = robot.dof_lower, robot.dof_upper
    q = q0 + rng.uniform(-0.5, 0.5, (N, n))
    lim = np.asarray(robot.dof_has_limits, bool)
    if lim.any():                                    # stay clear of the joint-limit springs
        q[:, lim] = np.clip(q[:, lim], lo[lim] + 0.15 * (hi - lo)[lim], hi[lim] - 0.15 * (hi - lo)[lim])
    vmax = np.where(robot.dof_velocity > 0, robot.dof_velocity, 20.0)
    qd = rng.uniform(-0.4, 0.4, (N, n)) * vmax      # well below the joint-speed limit (its fade starts at 90 %)
    quat = rng.normal(size=(N, 4)); quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    root = np.zeros((N, 13), np.float32)
    root[:, 0:2], root[:, 2] = grid_origins(N)[:, :2], 5.0
    root[:, 3:7] = quat
    root[:, 7:10] = rng.normal(0, 1.0, (N, 3))
    root[:, 10:13] = rng.normal(0, 2.0, (N, 3))
    eff = np.where(robot.dof_effort > 0, robot.dof_effort, 40.0)
    # large torques, but not so large that a light distal link runs into its joint-speed limit within the sub-step
    cap = np.array([eom.joint_inertia_about_axis(robot, q0, i // robot.chain_len, i % robot.chain_len) for i in range(n)]) * 0.25 * vmax / p.sim_dt
    tau = rng.uniform(-1.0, 1.0, (N, n)) * np.minimum(0.6 * eff, cap)
    dof = np.stack((q, qd), axis=-1).reshape(N * n, 2).astype(np.float32)
    return root, dof, tau.astype(np.float32)


def newton_euler_residuals(robot, p, root0, dof0, tau, root1, dof1, mass_delta):
    """Per env: (joint-torque error / scale, base force residual / (m g), base moment residual / scale)."""
    N, n, dt = root0.shape[0], robot.num_dof, float(p.sim_dt)
    out = []
    for e in range(N):
        q, qd = dof0.reshape(N, n, 2)[e, :, 0].astype(np.float64), dof0.reshape(N, n, 2)[e, :, 1].astype(np.float64)
        qd1 = dof1.reshape(N, n, 2)[e, :, 1].astype(np.float64)
        v0, w0 = root0[e, 7:10].astype(np.float64), root0[e, 10:13].astype(np.float64)
        a0, al0 = (root1[e, 7:10].astype(np.float64) - v0) / dt, (root1[e, 10:13].astype(np.float64) - w0) / dt
        t, f, m, scale = eom.inverse_dynamics(robot, root0[e, 3:7], v0, w0, q, qd, a0, al0, (qd1 - qd) / dt,
                                              gravity=tuple(p.gravity), base_mass_delta=float(mass_delta[e]))
        mg = (robot.total_mass + float(mass_delta[e])) * 9.81
        out.append((np.abs(t - tau[e]) / (scale + np.abs(tau[e])), np.linalg.norm(f) / mg, np.linalg.norm(m) / (scale.max() + mg * 0.3)))
    return out


def check_newton_euler(sim, robot, p, N, seed):
    rng = np.random.default_rng(seed + 1)
    root0, dof0, tau = random_airborne_state(robot, p, N, seed)
    dm = rng.uniform(-1.0, 3.0, N).astype(np.float32)
    sim.put("root_states", root0); sim.put("dof_state", dof0); sim.put("base_mass_delta", dm)
    sim.substep(tau)
    root1, dof1 = sim.get("root_states"), sim.get("dof_state")
    assert np.isfinite(root1).all() and np.isfinite(dof1).all()
    assert np.all(sim.get("contact_forces") == 0.0)
    res = newton_euler_residuals(robot, p, root0, dof0, tau, root1, dof1, dm)
    # The joint-speed limit (URDF ``velocity``) is part of the engine's dynamics: a light shank under a
    # large torque reaches it within one sub-step.  Newton-Euler with the applied torques holds for the envs that stayed below
    # the limit's fade-out (90 %) on every joint; the others are the limit's business, not this test's.
    vmax = np.where(robot.dof_velocity > 0, robot.dof_velocity, 20.0)
    free = (np.abs(dof1.reshape(N, -1, 2)[..., 1]) < 0.85 * vmax).all(axis=1)
    assert free.sum() >= N // 2, free.sum()
    res = [r for r, ok in zip(res, free) if ok]
    jt = np.array([r[0] for r in res]); bf = np.array([r[1] for r in res]); bm = np.array([r[2] for r in res])
    # the accelerations were not small: a wrong velocity-product term or inertia would show at the 1e-1 level
    assert np.abs(dof1.reshape(N, -1, 2)[..., 1] - dof0.reshape(N, -1, 2)[..., 1]).max() / p.sim_dt > 50.0
    assert jt.max() < 2e-5, ("joint torque residual", jt.max())
    assert bf.max() < 2e-5, ("base force residual / m g", bf.max())
    assert bm.max() < 2e-5, ("base moment residual", bm.max())
    return jt.max(), bf.max(), bm.max()


class OracleHandle:
    def __init__(self, task, N, tweak):
        from oracle.oracle import OracleSim
        self.cfg, self.robot, self.p, names, model, w = make_setup(task, N, tweak=tweak)
        self.o = OracleSim(self.p, model, self.robot, w)

    def put(self, k, v):
        self.o.buf[k][...] = np.asarray(v).astype(self.o.buf[k].dtype).reshape(self.o.buf[k].shape)

    def get(self, k):
        return self.o.buf[k].copy()

    def substep(self, tau):
        self.o.physics_substep(tau, True)

    def step(self, actions, counter):
        self.o.step(actions, counter)


class DeviceHandle:
    def __init__(self, task, N, tweak):
        import torch
        from legged_games_gym_amd.device_sim import DeviceSim
        self.cfg, self.robot, self.p, names, model, w = make_setup(task, N, tweak=tweak)
        self.d = DeviceSim(self.p, model, self.robot, torch.device("cuda:0"), w)
        self.torch = torch

    def put(self, k, v):
        t = self.d.buf[k]
        t.copy_(self.torch.from_numpy(np.ascontiguousarray(v)).to(t.dtype).view(t.shape))

    def get(self, k):
        self.torch.cuda.synchronize()
        t = self.d.buf[k]
        return (t.to(self.torch.uint8) if t.dtype == self.torch.bool else t).cpu().numpy()

    def substep(self, tau):
        self.d.physics_substep(self.torch.from_numpy(tau), True)

    def step(self, actions, counter):
        self.d.step(self.torch.from_numpy(np.ascontiguousarray(actions, np.float32)).cuda(), counter)


@pytest.mark.parametrize("task", ROBOTS)
def test_substep_satisfies_newton_euler_oracle(task, oracle_lib):
    h = OracleHandle(task, 24, no_rng)
    check_newton_euler(h, h.robot, h.p, 24, seed=ROBOTS.index(task))


@pytest.mark.gpu
@pytest.mark.parametrize("task", ROBOTS)
def test_substep_satisfies_newton_euler_hip(task):
    h = DeviceHandle(task, 200, no_rng)              # 200 envs: several workgroups and a ragged tail
    check_newton_euler(h, h.robot, h.p, 200, seed=10 + ROBOTS.index(task))


# ------------------------------------------------------------------------------------------------ pendulum period
KP = 25.0
# the held joints: stiff enough to follow quasi-statically, soft enough for the EXPLICIT P-controller at dt = 5 ms
# (shank + foot have 0.01 kg m^2 about the knee: Kd dt / I must stay well below 2)
LOCK = {"HFE": (600.0, 6.0), "KFE": (60.0, 0.5)}


def oscillator_tweak(cfg):
    no_rng(cfg)
    cfg.control.use_actuator_network = False
    cfg.control.control_type = "P"
    cfg.control.stiffness = {"HAA": KP, "HFE": LOCK["HFE"][0], "KFE": LOCK["KFE"][0]}        # the swinging joint / the two held ones
    cfg.control.damping = {"HAA": 0.0, "HFE": LOCK["HFE"][1], "KFE": LOCK["KFE"][1]}
    cfg.control.action_scale = 0.0                                              # actions play no role: tau = Kp (q0 - q) - Kd qd
    cfg.commands.resampling_time = 1.0e6
    cfg.env.episode_length_s = 1.0e4


def check_oscillation_period(h, N):
    robot, p = h.robot, h.p
    n, L = robot.num_dof, robot.chain_len
    q0 = np.array(list(p.default_dof_pos)[:n])
    amp = 0.05
    root = np.zeros((N, 13), np.float32)
    root[:, 0:2], root[:, 2], root[:, 6] = grid_origins(N)[:, :2], 5.0, 1.0
    q = np.tile(q0, (N, 1))
    q[:, 0::L] += amp                                                           # every leg's first joint (HAA) displaced
    h.put("root_states", root)
    h.put("dof_state", np.stack((q, np.zeros_like(q)), axis=-1).reshape(N * n, 2))
    h.put("base_mass_delta", np.full(N, 2.0e5, np.float32))                     # the trunk is an inertial frame to 1e-4
    dt = float(p.dt_policy)
    trace = []
    for s in range(1, 260):
        h.step(np.zeros((N, n), np.float32), s)
        trace.append(h.get("dof_state").reshape(N, n, 2)[:, 0::L, 0] - q0[0::L])
        assert not h.get("reset_buf").any()
    trace = np.array(trace)                                                      # [T, N, K]
    held = h.get("dof_state").reshape(N, n, 2)[..., 0] - q0
    assert np.abs(np.delete(held, np.arange(0, n, L), axis=1)).max() < 2e-3      # the held joints stayed put
    periods, want = [], []
    for k in range(robot.num_limbs):
        I = eom.joint_inertia_about_axis(robot, q0, k, 0)
        want.append(2.0 * np.pi * np.sqrt(I / KP))
        x = trace[:, 0, k]
        up = [i for i in range(len(x) - 1) if x[i] < 0.0 <= x[i + 1]]            # upward zero crossings, linearly interpolated
        t = [(i + x[i] / (x[i] - x[i + 1]) + 1) * dt for i in up]
        assert len(t) >= 3, "fewer than two full periods in the trace"
        periods.append((t[-1] - t[0]) / (len(t) - 1))
        assert 0.9 * amp < np.abs(x[len(x) // 2:]).max() < 1.02 * amp            # undamped: the amplitude is kept
    periods, want = np.array(periods), np.array(want)
    assert np.all(np.abs(periods / want - 1.0) < 5e-3), (periods, want)
    assert np.abs(trace[:, 0] - trace[:, -1]).max() < 1e-4                       # every env does the same
    return periods, want


def test_single_joint_oscillation_period_oracle(oracle_lib):
    h = OracleHandle("anymal_c_flat", 2, oscillator_tweak)
    from legged_games_gym_amd import capi
    capi._fill(h.p.gravity, (0.0, 0.0, 0.0))
    h.o.sim.set_params(h.p)
    check_oscillation_period(h, 2)


@pytest.mark.gpu
def test_single_joint_oscillation_period_hip():
    h = DeviceHandle("anymal_c_flat", 70, oscillator_tweak)
    from legged_games_gym_amd import capi
    capi._fill(h.p.gravity, (0.0, 0.0, 0.0))
    h.d.sim.set_params(h.p)
    check_oscillation_period(h, 70)


# ------------------------------------------------------------------------------------------------ reported contact forces
def check_contact_forces_are_the_external_force(h, N, seed, min_exact):
    """With contacts the base-wrench residual of the Newton-Euler inverse dynamics is no longer zero: it IS the external force.  The net
    contact forces the step exports (`contact_forces`, what refresh_net_contact_force_tensor exposes: legged_robot.py:112, consumers :142,
    :907, :945-968) must add up to it -- the numbers termination and rewards read are the forces that acted in the sub-step."""
    robot, p = h.robot, h.p
    rng = np.random.default_rng(seed)
    n = robot.num_dof
    q0 = np.array(list(p.default_dof_pos)[:n])
    q = q0 + rng.uniform(-0.02, 0.02, (N, n))           # near the default pose: the feet stay within millimetres of the plane
    qd = rng.normal(0, 0.5, (N, n))
    root = np.zeros((N, 13), np.float32)
    root[:, 0:2], root[:, 6] = grid_origins(N)[:, :2], 1.0
    # feet a few millimetres into the plane: lowest collision sphere of the default pose from the float64 model
    from tests.common import robot_capsules
    low = min(min(c[1][2], c[2][2]) - c[3] for c in robot_capsules(robot, q0))
    root[:, 2] = -low - rng.uniform(0.0005, 0.004, N)
    root[:, 7:10] = rng.normal(0, 0.2, (N, 3)); root[:, 10:13] = rng.normal(0, 0.3, (N, 3))
    eff = np.where(robot.dof_effort > 0, robot.dof_effort, 40.0)
    tau = (rng.uniform(-0.2, 0.2, (N, n)) * eff).astype(np.float32)
    dof = np.stack((q, qd), axis=-1).reshape(N * n, 2).astype(np.float32)
    dm = np.zeros(N, np.float32)
    h.put("root_states", root); h.put("dof_state", dof); h.put("base_mass_delta", dm)
    h.put("contact_forces", np.zeros_like(h.get("contact_forces")))
    h.substep(tau)
    root1, dof1, cf = h.get("root_states"), h.get("dof_state"), h.get("contact_forces").astype(np.float64)
    dt = float(p.sim_dt)
    loaded, exact, worst_exact = 0, 0, 0.0
    vmax = np.where(robot.dof_velocity > 0, robot.dof_velocity, 20.0)
    free = (np.abs(dof1.reshape(N, n, 2)[..., 1]) < 0.85 * vmax).all(axis=1)       # (the joint-speed limit is the engine's own business)
    assert free.sum() >= N // 2
    mg = robot.total_mass * 9.81
    for e in np.nonzero(free)[0]:
        qe, qde = dof.reshape(N, n, 2)[e, :, 0].astype(np.float64), dof.reshape(N, n, 2)[e, :, 1].astype(np.float64)
        qd1 = dof1.reshape(N, n, 2)[e, :, 1].astype(np.float64)
        v0, w0 = root[e, 7:10].astype(np.float64), root[e, 10:13].astype(np.float64)
        a0, al0 = (root1[e, 7:10].astype(np.float64) - v0) / dt, (root1[e, 10:13].astype(np.float64) - w0) / dt
        t, f, m, scale = eom.inverse_dynamics(robot, root[e, 3:7], v0, w0, qe, qde, a0, al0, (qd1 - qde) / dt, gravity=tuple(p.gravity))
        total = cf[e].sum(axis=0)
        ref = max(mg, float(np.linalg.norm(total)))
        miss = (f - total) / ref
        loaded += int(total[2] > 0.2 * mg)
        if np.linalg.norm(miss) < 2e-3:                  # forces of hundreds of newtons from fp32 velocity differences over 5 ms
            exact += 1
            worst_exact = max(worst_exact, float(np.linalg.norm(miss)))
            continue
        # The one way the two may differ ("lift-off adhesion"): a point the last contact pass still held active
        # that turns out to separate (f_n <= 0 at the end-of-step velocity) is exported as 0, but its spring-damper did pull during
        # this one sub-step.  So the force that acted is never MORE than the exported one, the gap is along the plane's normal and it is
        # bounded by the damper over one pass's velocity change.
        assert abs(miss[0]) < 2e-3 and abs(miss[1]) < 2e-3, (e, miss)
        assert -0.5 < miss[2] < 0.0, (e, miss)
    assert loaded >= free.sum() // 2, loaded             # the robots did stand on something
    assert exact >= min_exact * free.sum(), (exact, int(free.sum()))
    return worst_exact


# share of the robots whose exported forces must equal the acting ones to 2e-3: all but a few for the sphere-footed quadrupeds; Cassie's
# two-point toes rock (one end lifts while the other presses), so a point separating inside the sub-step is the common case there
MIN_EXACT = {"anymal_c_flat": 0.9, "anymal_b": 0.9, "a1": 0.9, "cassie": 0.4}


@pytest.mark.parametrize("task", ROBOTS)
def test_exported_contact_forces_add_up_to_the_external_force_oracle(task, oracle_lib):
    h = OracleHandle(task, 24, no_rng)
    check_contact_forces_are_the_external_force(h, 24, seed=20 + ROBOTS.index(task), min_exact=MIN_EXACT[task])


@pytest.mark.gpu
@pytest.mark.parametrize("task", ROBOTS)
def test_exported_contact_forces_add_up_to_the_external_force_hip(task):
    h = DeviceHandle(task, 200, no_rng)
    check_contact_forces_are_the_external_force(h, 200, seed=30 + ROBOTS.index(task), min_exact=MIN_EXACT[task])
