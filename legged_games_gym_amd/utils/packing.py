"""LeggedRobotCfg -> ``lg_params`` and the buffer table of the C-ABI.

Host-side restatement of what the reference derives from its config in
``legged_gym/envs/base/legged_robot.py``: ``_parse_cfg`` (:781-791),
``_prepare_reward_function`` (:583-607), ``_get_noise_scale_vec`` (:485-508),
``_init_buffers`` gains / default pose by DOF-name substring (:564-581),
``_process_dof_props`` soft limits (:299-314) and ``_init_height_points``
(:815-829).  Pure numpy; used by the env class and by the parity tests.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

from .. import capi
from .helpers import class_to_dict


@dataclass
class EngineOptions:
    """Parameters of the built-in rigid-body engine that have no counterpart in
    the reference config (PhysX hides them).  See DESIGN.md "Physics step"."""
    contact_stiffness: float = 1.0e6     # N/m, implicit
    contact_damping: float = 2.0e4       # N s/m normal damping (depenetration time constant b/k = 20 ms)
    friction_damping: float = 2.5e4      # N s/m tangential stick impedance (upper bound of the Coulomb secant)
    stick_velocity: float = 1.0e-3       # m/s, regularisation speed of the Coulomb law
    limit_stiffness: float = 2.0e4       # N m/rad, implicit joint-limit spring
    limit_damping: float = 2.0e2         # N m s/rad
    armature: float = 0.0


# Robot-specific engine options (keyed by ``asset.name``), used when the config carries no ``sim.engine`` of its own.
# A1 (12.5 kg, P-controlled legs with Kp 20 / Kd 0.5): on the 1e6 N/m ground spring of the 30-50 kg robots its feet chatter -- joint-speed rms
# of the STANDING robot 0.25 rad/s against 0.08 at 1e5 N/m, foot normal force 33 % above the weight under exploration noise -- and PPO with
# the reference's defaults does not ignite (action std 1 -> 3, reward -> 0) for any stiffness above ~1.2e5 N/m, while it trains for every
# value from 1e4 to 1e5 (tools/a1_probe.py, DESIGN.md "A1").  Normal / tangential damping do not matter; ANYmal-C / -B are insensitive to the
# stiffness over 1e5 .. 1e6.  Static foot penetration at 5e4 N/m: 0.6 mm.
ROBOT_ENGINE_OPTIONS = {"a1": EngineOptions(contact_stiffness=5.0e4)}


def reward_layout(cfg, dt: float) -> Tuple[np.ndarray, np.ndarray, List[str]]:
    """scale*dt per term id, slot per term id (-1 absent), slot names.
    Mirrors _prepare_reward_function: zero scales dropped, rest multiplied by dt,
    names in class_to_dict (alphabetical) order."""
    scales = class_to_dict(cfg.rewards.scales)
    scale = np.zeros(capi.LG_NUM_REWARD_TERMS, dtype=np.float32)
    slot = -np.ones(capi.LG_NUM_REWARD_TERMS, dtype=np.int32)
    names: List[str] = []
    for name, val in scales.items():
        if val == 0:
            continue
        if name not in capi.REWARD_TERMS:
            # same failure mode as getattr(self, '_reward_' + name) at legged_robot.py:602
            raise AttributeError(f"no reward function '_reward_{name}'")
        tid = capi.REWARD_TERMS.index(name)
        scale[tid] = np.float32(val * dt)
        slot[tid] = len(names)
        names.append(name)
    return scale, slot, names


def build_params(cfg, model, sim_dt: float, num_envs: int, seed: int, gravity=(0.0, 0.0, -9.81),
                 engine: EngineOptions = None, terrain=None, contact_offset: float = 0.01):
    """Returns (lg_params, reward_slot_names)."""
    engine = engine or getattr(cfg.sim, "engine", None) or ROBOT_ENGINE_OPTIONS.get(getattr(cfg.asset, "name", None)) or EngineOptions()
    p = capi.lg_params()
    n = model.num_dof
    dt = cfg.control.decimation * sim_dt
    p.abi_version, p.num_envs, p.decimation = capi.LG_ABI_VERSION, num_envs, cfg.control.decimation
    use_net = bool(getattr(cfg.control, "use_actuator_network", False))
    ctype = cfg.control.control_type
    if not use_net and ctype not in ("P", "V", "T"):
        raise NameError(f"Unknown controller type: {ctype}")     # legged_robot.py:394
    p.control_type = capi.CTRL["actuator_net"] if use_net else capi.CTRL[ctype]
    p.sim_dt = sim_dt
    capi._fill(p.gravity, gravity)
    p.contact_stiffness, p.contact_damping = engine.contact_stiffness, engine.contact_damping
    p.friction_damping, p.contact_margin = engine.friction_damping, contact_offset
    p.ground_friction = cfg.terrain.static_friction
    p.limit_stiffness, p.limit_damping = engine.limit_stiffness, engine.limit_damping
    p.stick_velocity = engine.stick_velocity
    p.action_scale = cfg.control.action_scale
    p.clip_actions, p.clip_observations = cfg.normalization.clip_actions, cfg.normalization.clip_observations

    kp, kd, q0 = np.zeros(n), np.zeros(n), np.zeros(n)
    for i, name in enumerate(model.dof_names):
        q0[i] = cfg.init_state.default_joint_angles[name]
        for key in cfg.control.stiffness.keys():
            if key in name:
                kp[i], kd[i] = cfg.control.stiffness[key], cfg.control.damping[key]
    capi._fill(p.p_gains, kp)
    capi._fill(p.d_gains, kd)
    capi._fill(p.default_dof_pos, q0)
    capi._fill(p.torque_limits, model.dof_effort)
    lo, hi = model.dof_lower.copy(), model.dof_upper.copy()      # 0,0 when the URDF has none (Isaac Gym reports 0)
    mid, rng = (lo + hi) / 2, hi - lo
    capi._fill(p.soft_pos_lower, mid - 0.5 * rng * cfg.rewards.soft_dof_pos_limit)
    capi._fill(p.soft_pos_upper, mid + 0.5 * rng * cfg.rewards.soft_dof_pos_limit)
    capi._fill(p.dof_vel_limits, model.dof_velocity)
    p.soft_dof_vel_limit, p.soft_torque_limit = cfg.rewards.soft_dof_vel_limit, cfg.rewards.soft_torque_limit
    p.tracking_sigma, p.base_height_target = cfg.rewards.tracking_sigma, cfg.rewards.base_height_target
    p.max_contact_force, p.dt_policy = cfg.rewards.max_contact_force, dt
    p.max_push_vel = cfg.domain_rand.max_push_vel_xy

    p.max_episode_length = int(np.ceil(cfg.env.episode_length_s / dt))
    p.max_episode_length_s = cfg.env.episode_length_s
    p.push_interval = int(np.ceil(cfg.domain_rand.push_interval_s / dt)) if cfg.domain_rand.push_robots else 0
    p.resample_interval = int(cfg.commands.resampling_time / dt)
    p.heading_command = int(bool(cfg.commands.heading_command))
    r = cfg.commands.ranges
    capi._fill(p.cmd_lin_vel_x, r.lin_vel_x)
    capi._fill(p.cmd_lin_vel_y, r.lin_vel_y)
    capi._fill(p.cmd_ang_vel_yaw, r.ang_vel_yaw)
    capi._fill(p.cmd_heading, r.heading)

    s, ns, lvl = cfg.normalization.obs_scales, cfg.noise.noise_scales, cfg.noise.noise_level
    p.obs_scale_lin_vel, p.obs_scale_ang_vel = s.lin_vel, s.ang_vel
    p.obs_scale_dof_pos, p.obs_scale_dof_vel, p.obs_scale_height = s.dof_pos, s.dof_vel, s.height_measurements
    p.noise_lin_vel = ns.lin_vel * lvl * s.lin_vel
    p.noise_ang_vel = ns.ang_vel * lvl * s.ang_vel
    p.noise_gravity = ns.gravity * lvl
    p.noise_dof_pos = ns.dof_pos * lvl * s.dof_pos
    p.noise_dof_vel = ns.dof_vel * lvl * s.dof_vel
    p.noise_height = ns.height_measurements * lvl * s.height_measurements
    p.add_noise = int(bool(cfg.noise.add_noise))
    p.measure_heights = int(bool(cfg.terrain.measure_heights))
    p.num_obs = cfg.env.num_observations
    if p.measure_heights:
        xs, ys = cfg.terrain.measured_points_x, cfg.terrain.measured_points_y
        pts = np.array([[x, y] for x in xs for y in ys], dtype=np.float64)   # meshgrid 'ij' flattened (:823-828)
        if len(pts) > capi.LG_MAX_HEIGHT_POINTS:
            raise ValueError("too many height points")
        p.num_height_points = len(pts)
        capi._fill(p.height_points, pts)
    if p.num_obs != 48 + p.num_height_points:
        raise ValueError(f"num_observations={p.num_obs} but the observation vector has {48 + p.num_height_points} entries")

    scale, slot, names = reward_layout(cfg, dt)
    capi._fill(p.reward_scale, scale)
    for i in range(capi.LG_NUM_REWARD_TERMS):
        p.reward_slot[i] = int(slot[i])
    p.num_reward_slots = len(names)
    p.only_positive_rewards = int(bool(cfg.rewards.only_positive_rewards))

    mesh = cfg.terrain.mesh_type
    if mesh in ("heightfield", "trimesh") and terrain is not None:
        p.terrain_type = capi.TERRAIN_HEIGHTFIELD
        p.hf_rows, p.hf_cols = terrain.tot_rows, terrain.tot_cols
        p.hf_horizontal_scale, p.hf_vertical_scale = cfg.terrain.horizontal_scale, cfg.terrain.vertical_scale
        p.hf_border = cfg.terrain.border_size
        # 'trimesh': the reference corrects slopes above slope_treshold to vertical faces when it triangulates the samples
        # (terrain.py:69-73, legged_robot_config.py:66); 'heightfield' collides against the plain (bilinear) samples
        thr = getattr(cfg.terrain, "slope_treshold", None)
        p.hf_step_threshold = float(thr) * cfg.terrain.horizontal_scale if (mesh == "trimesh" and thr is not None) else 0.0
        p.custom_origins = 1
        p.terrain_curriculum = int(bool(cfg.terrain.curriculum))
        p.terrain_num_rows, p.terrain_num_cols = cfg.terrain.num_rows, cfg.terrain.num_cols
        p.terrain_env_length = cfg.terrain.terrain_length
    else:
        p.terrain_type, p.custom_origins, p.terrain_curriculum = capi.TERRAIN_PLANE, 0, 0
    # asset.self_collisions is Isaac Gym's collision-filter bitmask: 0 = links of the robot collide with each other (legged_robot.py:683)
    p.self_collision = int(int(getattr(cfg.asset, "self_collisions", 1)) == 0)
    init = cfg.init_state
    capi._fill(p.base_init_state, list(init.pos) + list(init.rot) + list(init.lin_vel) + list(init.ang_vel))
    p.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return p, names


def buffer_spec(p, model) -> Dict[str, Tuple[Tuple[int, ...], str]]:
    """name -> (shape, dtype string) of every array of ``lg_buffers`` the caller allocates."""
    N, n, K, nb = p.num_envs, model.num_dof, model.num_limbs, model.num_bodies
    R = max(p.num_reward_slots, 1)
    spec = {
        "root_states": ((N, 13), "float32"), "dof_state": ((N * n, 2), "float32"),
        "contact_forces": ((N, nb, 3), "float32"), "obs_buf": ((N, p.num_obs), "float32"),
        "rew_buf": ((N,), "float32"), "reset_buf": ((N,), "bool"), "time_out_buf": ((N,), "bool"),
        "episode_length_buf": ((N,), "int64"),
        "torques": ((N, n), "float32"), "actions": ((N, n), "float32"), "last_actions": ((N, n), "float32"),
        "last_dof_vel": ((N, n), "float32"), "last_root_vel": ((N, 6), "float32"), "commands": ((N, 4), "float32"),
        "feet_air_time": ((N, K), "float32"), "last_contacts": ((N, K), "bool"),
        "base_lin_vel": ((N, 3), "float32"), "base_ang_vel": ((N, 3), "float32"), "projected_gravity": ((N, 3), "float32"),
        "episode_sums": ((R, N), "float32"), "episode_means": ((R + 1,), "float32"), "extras_accum": ((R + 1,), "float32"), "step_counter": ((1,), "int64"),
        "env_origins": ((N, 3), "float32"), "friction_coeffs": ((N,), "float32"), "base_mass_delta": ((N,), "float32"),
    }
    if p.measure_heights:
        spec["measured_heights"] = ((N, p.num_height_points), "float32")
    if p.control_type == capi.CTRL["actuator_net"]:
        spec["sea_hidden_state"] = ((2, N * n, 8), "float32")
        spec["sea_cell_state"] = ((2, N * n, 8), "float32")
    if p.terrain_type == capi.TERRAIN_HEIGHTFIELD:
        spec["terrain_levels"] = ((N,), "int32")
        spec["terrain_types"] = ((N,), "int32")
    return spec


def load_actuator_weights(path: str = None) -> np.ndarray:
    """The 972-float blob extracted from the reference's anydrive_v3_lstm.pt
    (tools/compile_models.py; raw storages, nothing executed).

    ``path`` may be a ``.f32`` blob, ``None``, or the registered config's ``.../anydrive_v3_lstm.pt`` (for which the bundled
    extraction is used).  Any OTHER TorchScript file is refused: silently running the bundled net's dynamics in its place
    would be wrong."""
    import os
    here = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
    bundled = os.path.join(here, "resources", "actuator_nets", "anydrive_v3_lstm.f32")
    if path is None or os.path.basename(path) == "anydrive_v3_lstm.pt":
        path = bundled
    elif not path.endswith(".f32"):
        raise ValueError(f"control.actuator_net_file = {path!r}: only 972-float .f32 blobs are loaded (TorchScript archives are never "
                         "executed here); convert the file with tools/compile_models.py (static, no-code extraction) and point the "
                         "config at the resulting .f32")
    w = np.fromfile(path, dtype="<f4")
    if w.size != capi.LG_ACTUATOR_FLOATS:
        raise ValueError(f"{path}: expected {capi.LG_ACTUATOR_FLOATS} floats, found {w.size}")
    return w
