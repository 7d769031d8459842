"""-m gpu: BASELINE.json configs 3 and 5 at their full size, against the CPU oracle and against size-independent invariants.

config 3: anymal_c_rough, 4096 envs, the full 1300 x 2100 int16 curriculum height field, actuator-net torques
          (kernel k_step<AnymalTraits, NET, HF>, four waves per workgroup);
config 5: cassie, 8192 envs/GPU on the same kind of terrain, friction / base-mass randomisation, pushes
          (kernel k_step<CassieTraits, PD, HF, NW = 4>: 8192 envs x 2 limbs = 256 workgroups, one per CU; the NW = 2 and
          NW = 1 height-field instantiations, used beyond 8192 Cassie envs, are covered by test_cassie_heightfield_step_parity).

Protocol of the at-size parity check: the robots first settle on the terrain for a few policy steps on the device
(so feet are in contact with slopes / stairs / obstacles, not in free fall), the COMPLETE device state is copied into the
oracle's host buffers, then both sides take the same policy step -- the push step (common_step_counter % 750 == 0) --
with the same actions, and EVERY env is compared.  Tolerances: integer / boolean outputs bit-equal; floating point within
the fp32 budget of one policy step with stiff contacts (see test_gpu_parity.py), stated per assert.
"""
import numpy as np
import pytest
import torch

from tests.common import make_setup

pytestmark = pytest.mark.gpu


def _full_terrain(task, N, seed=11):
    from legged_games_gym_amd.utils.terrain import Terrain
    from tests.common import TASK_CFG
    tc = TASK_CFG[task]().terrain
    tc.mesh_type = "heightfield"                       # BASELINE.json configs 3 / 5: height-field contact
    np.random.seed(seed)
    terr = Terrain(tc, N)
    assert terr.heightsamples.shape == (1300, 2100)
    return terr


def _pair_on_terrain(task, N, terr, tweak=None, seed=1):
    from oracle.oracle import OracleSim
    from legged_games_gym_amd.device_sim import DeviceSim

    def tw(cfg):
        cfg.terrain.mesh_type = "heightfield"
        if tweak:
            tweak(cfg)
    cfg, robot, p, names, model, w = make_setup(task, N, seed=seed, tweak=tw, terrain=terr, plane=False)
    o = OracleSim(p, model, robot, w, threads=16)
    d = DeviceSim(p, model, robot, torch.device("cuda:0"), w)
    o.set_terrain(terr.heightsamples, terr.env_origins)
    d.set_terrain(terr.heightsamples, terr.env_origins)
    return cfg, robot, p, names, o, d


def _spawn(d, N, terr, cfg, rng, friction_range, mass_range):
    """Creation-time randomisation (legged_robot.py:261-285, 316-327), terrain levels / types (:752-769), reset of every env."""
    buckets = rng.uniform(friction_range[0], friction_range[1], 64).astype(np.float32)
    fr = buckets[rng.integers(0, 64, N)]
    dm = rng.uniform(mass_range[0], mass_range[1], N).astype(np.float32)
    lv = rng.integers(0, cfg.terrain.max_init_terrain_level + 1, N).astype(np.int32)
    ty = np.floor(np.arange(N) / (N / cfg.terrain.num_cols)).astype(np.int32)
    d.buf["friction_coeffs"].copy_(torch.from_numpy(fr)); d.buf["base_mass_delta"].copy_(torch.from_numpy(dm))
    d.buf["terrain_levels"].copy_(torch.from_numpy(lv)); d.buf["terrain_types"].copy_(torch.from_numpy(ty))
    d.buf["env_origins"].copy_(torch.from_numpy(terr.env_origins[lv, ty].astype(np.float32)))
    d.reset_idx(torch.arange(N, dtype=torch.int32), 0)
    return dm


def _device_to_oracle(d, o):
    torch.cuda.synchronize()
    for name, dst in o.buf.items():
        t = d.buf[name]
        dst[...] = (t.to(torch.uint8) if t.dtype == torch.bool else t).cpu().numpy().astype(dst.dtype).reshape(dst.shape)


def _get(d, name):
    torch.cuda.synchronize()
    t = d.buf[name]
    return (t.to(torch.uint8) if t.dtype == torch.bool else t).cpu().numpy()


def _step_parity_every_env(o, d, N, act, step, vel_tol, pos_tol, obs_tol, rew_tol, min_contact_frac):
    o.step(act.numpy(), step); d.step(act.cuda(), step)
    # --- integer / boolean outputs: bit-equal on every env
    for k in ("reset_buf", "time_out_buf", "episode_length_buf", "terrain_levels"):
        assert np.array_equal(o.buf[k], _get(d, k)), k
    survivors = o.buf["reset_buf"] == 0
    # --- the contacts were real: most robots carry weight on the height field at the end of the step
    cf_o, cf_d = o.buf["contact_forces"], _get(d, "contact_forces")
    assert (np.abs(cf_o[..., 2]).sum(axis=1) > 50.0).mean() > min_contact_frac
    q_o, q_d = o.buf["dof_state"].reshape(N, 12, 2), _get(d, "dof_state").reshape(N, 12, 2)
    r_o, r_d = o.buf["root_states"], _get(d, "root_states")
    e_pos = np.abs(q_o[..., 0] - q_d[..., 0]).max(axis=1)
    e_vel = np.abs(q_o[..., 1] - q_d[..., 1]).max(axis=1)
    e_root = np.abs(r_o[:, :7] - r_d[:, :7]).max(axis=1)
    e_rv = np.abs(r_o[:, 7:] - r_d[:, 7:]).max(axis=1)
    # 99.9 % of the envs within the one-step fp32 budget (stiff implicit contacts amplify rounding of the 1e6 N/m springs);
    # the rest may have taken a different BRANCH on a 1-ulp difference (contact on / off at the margin, stick / slide at
    # the cone, speed-limit damper on / off): bounded, an order of magnitude looser
    q999 = lambda x: float(np.quantile(x, 0.999))
    assert q999(e_pos) < pos_tol and q999(e_root) < pos_tol, (q999(e_pos), q999(e_root))
    assert q999(e_vel) < vel_tol and q999(e_rv) < vel_tol, (q999(e_vel), q999(e_rv))
    assert e_pos.max() < 20 * pos_tol and e_root.max() < 20 * pos_tol and e_vel.max() < 20 * vel_tol, (e_pos.max(), e_root.max(), e_vel.max())
    # ... and the bulk two orders tighter
    assert np.median(e_pos) < pos_tol / 50 and np.median(e_vel) < vel_tol / 50, (np.median(e_pos), np.median(e_vel))
    mh_o, mh_d = o.buf["measured_heights"], _get(d, "measured_heights")
    assert (np.abs(mh_o - mh_d) > 1e-6).mean() < 2e-3          # a sample point within 1 ulp of a cell edge may truncate differently
    bulk = (e_pos < pos_tol) & (e_vel < vel_tol)                 # envs that took the same branches
    assert bulk.mean() > 0.998
    assert np.abs(o.buf["rew_buf"] - _get(d, "rew_buf"))[bulk].max() < rew_tol
    same_cells = ~(np.abs(mh_o - mh_d) > 1e-6).any(axis=1) & bulk
    assert np.abs(o.buf["obs_buf"][same_cells] - _get(d, "obs_buf")[same_cells]).max() < obs_tol
    assert np.abs(o.buf["episode_sums"] - _get(d, "episode_sums"))[:, bulk].max() < rew_tol
    f_scale = max(1.0, float(np.abs(cf_o).max()))
    assert np.abs(cf_o - cf_d)[bulk].max() < 2e-3 * f_scale, (np.abs(cf_o - cf_d)[bulk].max(), f_scale)
    # push step: every surviving env got a new xy velocity within +-max_push_vel, identical on both sides (Philox key)
    if step % 750 == 0:
        assert np.abs(r_o[survivors][:, 7:9]).max() <= 1.0 + 1e-6 and np.abs(r_o[:, 7:9] - r_d[:, 7:9]).max() < 1e-6
    return survivors


def _invariants(d, N, robot, dm, steps, action_std, vel_limit, first_step, upright_frac):
    g = torch.Generator(device="cuda").manual_seed(0)
    resets = 0
    for it in range(first_step, first_step + steps):
        a = torch.randn(N, 12, device="cuda", generator=g) * (action_std if it < first_step + steps - 80 else 0.0)   # flail, then settle
        d.step(a, it)
        resets += int(d.buf["reset_buf"].sum())
    root, dof = _get(d, "root_states"), _get(d, "dof_state").reshape(N, 12, 2)
    assert np.isfinite(root).all() and np.isfinite(dof).all() and np.isfinite(_get(d, "obs_buf")).all()
    np.testing.assert_allclose(np.linalg.norm(root[:, 3:7], axis=1), 1.0, atol=1e-5)
    assert np.abs(dof[..., 1]).max() <= vel_limit + 1e-3                    # URDF joint velocity limits
    assert np.abs(_get(d, "obs_buf")).max() <= 100.0                        # clip_observations
    lv = _get(d, "terrain_levels")
    assert lv.min() >= 0 and lv.max() <= 9 and resets > 0
    # envs sit on their own tile: within the tile grid, and above the local ground
    assert root[:, 0].min() > -5 and root[:, 0].max() < 85 and root[:, 1].min() > -5 and root[:, 1].max() < 165
    mh = _get(d, "measured_heights")
    assert ((root[:, 2:3] - mh).mean(axis=1) > 0.05).mean() > 0.99
    if robot.dof_has_limits.any():                                          # implicit joint-limit springs hold (Cassie)
        lim = robot.dof_has_limits
        q = dof[..., 0][:, lim]
        assert (q > robot.dof_lower[lim] - 0.15).all() and (q < robot.dof_upper[lim] + 0.15).all()
    pg = _get(d, "projected_gravity")
    up = (pg[:, 2] < -0.95) & (np.abs(root[:, 9]) < 0.05) & (_get(d, "episode_length_buf") > 60)
    if upright_frac is not None:
        assert up.mean() > upright_frac, up.mean()
    fz = _get(d, "contact_forces")[:, :, 2].sum(axis=1)
    mass = robot.total_mass + dm
    # standing robots: the terrain carries the weight (normal not vertical on slopes: sum of f_z still balances gravity)
    if up.sum() >= 20:
        assert np.median(np.abs(fz[up] / (mass[up] * 9.81) - 1.0)) < 0.03
    # nobody is pressed into / launched off the ground: net vertical contact force stays within a few body weights
    # (landings after a fall peak at a few body weights for a step; friction on a slope has a small downward z component)
    assert (fz < 8.0 * mass * 9.81).mean() > 0.97 and (fz < 40.0 * mass * 9.81).mean() > 0.999 and fz.min() > -0.2 * mass.min() * 9.81


@pytest.mark.parametrize("N", [200, 16400, 33000])
def test_cassie_heightfield_step_parity(N):
    """Cassie (2 x 6 chains, PD control, joint limits) on a height field against the oracle, every env, toes on the ground:
    N = 200 -> NW = 4 waves per workgroup, 16400 -> NW = 2 (513 workgroups), 33000 -> NW = 1."""
    from tests.test_gpu_parity import _rough_terrain
    terr = _rough_terrain(N)

    def tweak(cfg):
        cfg.terrain.mesh_type, cfg.terrain.num_rows, cfg.terrain.num_cols, cfg.terrain.border_size = "heightfield", 4, 5, 5
        cfg.terrain.max_init_terrain_level = 3
    from oracle.oracle import OracleSim
    from legged_games_gym_amd.device_sim import DeviceSim
    cfg, robot, p, names, model, w = make_setup("cassie", N, tweak=tweak, terrain=terr, plane=False)
    o = OracleSim(p, model, robot, w, threads=16)
    d = DeviceSim(p, model, robot, torch.device("cuda:0"), w)
    o.set_terrain(terr.heightsamples, terr.env_origins); d.set_terrain(terr.heightsamples, terr.env_origins)
    _spawn(d, N, terr, cfg, np.random.default_rng(0), (0.5, 1.25), (-1.0, 1.0))
    z = torch.zeros(N, 12, device="cuda")
    for it in range(1, 16):                              # 0.3 s: the 21 cm drop onto the terrain and the first contact transients
        d.step(z, it)
    _device_to_oracle(d, o)
    act = (torch.randn(N, 12, generator=torch.Generator().manual_seed(1)) * 0.3).float()
    _step_parity_every_env(o, d, N, act, 16, vel_tol=0.3, pos_tol=2e-3, obs_tol=2e-2, rew_tol=2e-3, min_contact_frac=0.7)


def test_config5_cassie_8192_full_terrain():
    """BASELINE.json configs[4]: cassie rough, 8192 envs/GPU, friction / mass randomisation, random pushes."""
    N = 8192
    terr = _full_terrain("cassie", N)

    def tweak(cfg):
        cfg.domain_rand.randomize_base_mass = True      # base mass +-1 kg
    cfg, robot, p, names, o, d = _pair_on_terrain("cassie", N, terr, tweak)
    assert p.push_interval == 750 and p.terrain_curriculum == 1 and p.measure_heights == 1 and p.num_obs == 169
    rng = np.random.default_rng(5)
    dm = _spawn(d, N, terr, cfg, rng, cfg.domain_rand.friction_range, cfg.domain_rand.added_mass_range)
    z = torch.zeros(N, 12, device="cuda")
    for it in range(730, 750):                           # settle onto the terrain (PD holds the default pose)
        d.step(z, it)
    _device_to_oracle(d, o)
    act = (torch.randn(N, 12, generator=torch.Generator().manual_seed(5)) * 0.3).float()
    _step_parity_every_env(o, d, N, act, 750, vel_tol=0.3, pos_tol=2e-3, obs_tol=2e-2, rew_tol=2e-3, min_contact_frac=0.7)
    # a biped under a random policy falls within a second: no "most robots stand" property here (the quadruped test has it)
    _invariants(d, N, robot, dm, steps=200, action_std=0.5, vel_limit=float(np.max(robot.dof_velocity)), first_step=751, upright_frac=None)


def test_config3_anymal_rough_4096_full_terrain():
    """BASELINE.json configs[2]: anymal_c_rough, 4096 envs, height-field terrain + actuator-net torques."""
    N = 4096
    terr = _full_terrain("anymal_c_rough", N)
    cfg, robot, p, names, o, d = _pair_on_terrain("anymal_c_rough", N, terr)
    assert p.terrain_curriculum == 1 and p.num_obs == 235 and "sea_hidden_state" in d.buf
    rng = np.random.default_rng(3)
    dm = _spawn(d, N, terr, cfg, rng, cfg.domain_rand.friction_range, cfg.domain_rand.added_mass_range)
    z = torch.zeros(N, 12, device="cuda")
    for it in range(735, 750):
        d.step(z, it)
    _device_to_oracle(d, o)
    act = (torch.randn(N, 12, generator=torch.Generator().manual_seed(3)) * 0.3).float()
    _step_parity_every_env(o, d, N, act, 750, vel_tol=0.1, pos_tol=1e-3, obs_tol=1e-2, rew_tol=1e-3, min_contact_frac=0.8)
    assert np.abs(o.buf["sea_hidden_state"] - _get(d, "sea_hidden_state")).max() < 5e-3
    _invariants(d, N, robot, dm, steps=200, action_std=1.0, vel_limit=20.0, first_step=751, upright_frac=0.3)


def test_trimesh_vertical_faces_parity():
    """mesh_type 'trimesh' (the registered anymal_c_rough / cassie default): stair risers and obstacle edges beyond slope_treshold are
    vertical faces (hf_step_threshold).  (i) the riser scenario of tests/test_oracle_physics.py, HIP against oracle, with the
    _reward_stumble condition on the front feet; (ii) 1000 envs on curriculum stairs / obstacles, settled, one policy step."""
    from tests.test_oracle_physics import _riser_setup, riser_state
    from oracle.oracle import OracleSim
    from legged_games_gym_amd.device_sim import DeviceSim
    terr, cfg, robot, p, names, model, w, x_face = _riser_setup("trimesh", N=64)
    N = 64
    o = OracleSim(p, model, robot, w)
    o.set_terrain(terr.heightsamples, terr.env_origins)
    d = DeviceSim(p, model, robot, torch.device("cuda:0"), w, terr.heightsamples, terr.env_origins)
    riser_state(o, p, robot, N, x_face)
    rng = np.random.default_rng(0)
    o.buf["root_states"][:, 0] += rng.uniform(-0.03, 0.02, N).astype(np.float32)      # from 2 cm clear of the face to 4 cm inside
    for k in ("root_states", "dof_state", "contact_forces", "env_origins"):
        d.buf[k].copy_(torch.from_numpy(o.buf[k]).view(d.buf[k].shape))
    tau = rng.normal(0, 5.0, (N, 12)).astype(np.float32)
    o.physics_substep(tau, True); d.physics_substep(torch.from_numpy(tau), True)
    cf_o, cf_d = o.buf["contact_forces"], _get(d, "contact_forces")
    feet = robot.bodies_matching("FOOT")
    stum_o = np.linalg.norm(cf_o[:, feet, :2], axis=-1) > 5.0 * np.abs(cf_o[:, feet, 2])
    stum_d = np.linalg.norm(cf_d[:, feet, :2], axis=-1) > 5.0 * np.abs(cf_d[:, feet, 2])
    assert stum_o[:, [0, 2]].any(axis=1).mean() > 0.3 and np.array_equal(stum_o, stum_d)          # LF / RF against the riser
    assert np.abs(cf_o - cf_d).max() < 2e-3 * max(1.0, np.abs(cf_o).max())
    assert np.abs(o.buf["dof_state"] - _get(d, "dof_state")).max() < 2e-3 and np.abs(o.buf["root_states"] - _get(d, "root_states")).max() < 1e-3
    # (ii) curriculum terrain with stairs / discrete obstacles
    from tests.test_gpu_parity import _rough_terrain
    N = 1000
    terr = _rough_terrain(N)

    def tweak(cfg):
        cfg.terrain.mesh_type, cfg.terrain.num_rows, cfg.terrain.num_cols, cfg.terrain.border_size = "trimesh", 4, 5, 5
        cfg.terrain.max_init_terrain_level = 3
    cfg, robot, p, names, model, w = make_setup("anymal_c_rough", N, tweak=tweak, terrain=terr, plane=False)
    assert p.hf_step_threshold > 0
    o = OracleSim(p, model, robot, w, threads=16)
    d = DeviceSim(p, model, robot, torch.device("cuda:0"), w)
    o.set_terrain(terr.heightsamples, terr.env_origins); d.set_terrain(terr.heightsamples, terr.env_origins)
    _spawn(d, N, terr, cfg, np.random.default_rng(0), (0.5, 1.25), (-5.0, 5.0))
    g = torch.Generator(device="cuda").manual_seed(0)
    for it in range(1, 41):                              # wander a little so that feet meet risers
        d.step(torch.randn(N, 12, device="cuda", generator=g) * 0.5, it)
    _device_to_oracle(d, o)
    act = (torch.randn(N, 12, generator=torch.Generator().manual_seed(1)) * 0.3).float()
    _step_parity_every_env(o, d, N, act, 41, vel_tol=0.1, pos_tol=1e-3, obs_tol=1e-2, rew_tol=1e-3, min_contact_frac=0.7)
