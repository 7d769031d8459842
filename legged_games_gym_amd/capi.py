"""ctypes mirror of ``include/legged_hip.h`` and the loader of the HIP library.

This is the "thin C-ABI" of the north star: Python hands raw device pointers
(``tensor.data_ptr()``) and plain structs to ``liblegged_hip.so``; it plays
the role ``gymapi``/``gymtorch`` play in the reference (call sites
``legged_gym/envs/base/legged_robot.py:92-96, 111-112, 410-444, 515-529``).

There is NO CPU fallback: ``load_library()`` raises if the HIP extension has
not been built (``python -c "import __graft_entry__ as g; g.build()"``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

LG_ABI_VERSION = 21
LG_ADAM_SCRATCH_FLOATS = 2050
LG_MAX_LIMBS, LG_MAX_CHAIN, LG_MAX_DOF = 4, 6, 12
LG_MAX_LIMB_POINTS, LG_MAX_BASE_POINTS, LG_MAX_BODIES = 8, 4, 20
LG_MAX_HEIGHT_POINTS, LG_ACTUATOR_FLOATS = 192, 972

# alphabetical == enum order in legged_hip.h
REWARD_TERMS = ["action_rate", "ang_vel_xy", "base_height", "collision", "dof_acc", "dof_pos_limits",
                "dof_vel", "dof_vel_limits", "feet_air_time", "feet_contact_forces", "lin_vel_z", "no_fly",
                "orientation", "stand_still", "stumble", "termination", "torque_limits", "torques",
                "tracking_ang_vel", "tracking_lin_vel"]
LG_NUM_REWARD_TERMS = len(REWARD_TERMS)
CTRL = {"P": 0, "V": 1, "T": 2, "actuator_net": 3}
TERRAIN_PLANE, TERRAIN_HEIGHTFIELD = 0, 1

f32, i32, u32, u8, i64, u64, i16 = C.c_float, C.c_int32, C.c_uint32, C.c_uint8, C.c_int64, C.c_uint64, C.c_int16


class lg_point(C.Structure):
    _fields_ = [("pos", f32 * 3), ("radius", f32), ("report_body", i32), ("joint", i32)]


class lg_robot_model(C.Structure):
    _fields_ = [
        ("num_limbs", i32), ("chain_len", i32), ("num_bodies", i32), ("_pad0", i32),
        ("base_mass", f32), ("base_com", f32 * 3), ("base_inertia", f32 * 6),
        ("joint_pos", (f32 * 3) * LG_MAX_DOF), ("joint_rot", (f32 * 9) * LG_MAX_DOF), ("joint_axis", (f32 * 3) * LG_MAX_DOF),
        ("body_mass", f32 * LG_MAX_DOF), ("body_com", (f32 * 3) * LG_MAX_DOF), ("body_inertia", (f32 * 6) * LG_MAX_DOF),
        ("dof_lower", f32 * LG_MAX_DOF), ("dof_upper", f32 * LG_MAX_DOF), ("dof_vel_limit", f32 * LG_MAX_DOF),
        ("dof_armature", f32 * LG_MAX_DOF), ("dof_damping", f32 * LG_MAX_DOF),
        ("num_base_points", i32), ("num_limb_points", i32 * LG_MAX_LIMBS), ("_pad1", i32 * 3),
        ("base_points", lg_point * LG_MAX_BASE_POINTS),
        ("limb_points", (lg_point * LG_MAX_LIMB_POINTS) * LG_MAX_LIMBS),
        ("foot_body", i32 * LG_MAX_LIMBS), ("penalised_mask", u32), ("termination_mask", u32),
    ]


class lg_params(C.Structure):
    _fields_ = [
        ("abi_version", i32), ("num_envs", i32), ("decimation", i32), ("control_type", i32),
        ("sim_dt", f32), ("gravity", f32 * 3),
        ("contact_stiffness", f32), ("contact_damping", f32), ("friction_damping", f32), ("contact_margin", f32),
        ("ground_friction", f32), ("limit_stiffness", f32), ("limit_damping", f32), ("stick_velocity", f32),
        ("action_scale", f32), ("clip_actions", f32), ("clip_observations", f32), ("_padf1", f32),
        ("p_gains", f32 * LG_MAX_DOF), ("d_gains", f32 * LG_MAX_DOF), ("default_dof_pos", f32 * LG_MAX_DOF),
        ("torque_limits", f32 * LG_MAX_DOF),
        ("soft_pos_lower", f32 * LG_MAX_DOF), ("soft_pos_upper", f32 * LG_MAX_DOF), ("dof_vel_limits", f32 * LG_MAX_DOF),
        ("soft_dof_vel_limit", f32), ("soft_torque_limit", f32), ("tracking_sigma", f32), ("base_height_target", f32),
        ("max_contact_force", f32), ("dt_policy", f32), ("max_push_vel", f32), ("_padf2", f32),
        ("max_episode_length", i32), ("push_interval", i32), ("resample_interval", i32), ("heading_command", i32),
        ("cmd_lin_vel_x", f32 * 2), ("cmd_lin_vel_y", f32 * 2), ("cmd_ang_vel_yaw", f32 * 2), ("cmd_heading", f32 * 2),
        ("obs_scale_lin_vel", f32), ("obs_scale_ang_vel", f32), ("obs_scale_dof_pos", f32), ("obs_scale_dof_vel", f32),
        ("obs_scale_height", f32),
        ("noise_lin_vel", f32), ("noise_ang_vel", f32), ("noise_gravity", f32), ("noise_dof_pos", f32),
        ("noise_dof_vel", f32), ("noise_height", f32),
        ("add_noise", i32), ("measure_heights", i32), ("num_height_points", i32), ("num_obs", i32),
        ("height_points", (f32 * 2) * LG_MAX_HEIGHT_POINTS),
        ("reward_scale", f32 * LG_NUM_REWARD_TERMS),
        ("only_positive_rewards", i32), ("reward_slot", i32 * LG_NUM_REWARD_TERMS), ("num_reward_slots", i32),
        ("terrain_type", i32), ("hf_rows", i32), ("hf_cols", i32), ("custom_origins", i32),
        ("hf_horizontal_scale", f32), ("hf_vertical_scale", f32), ("hf_border", f32), ("hf_step_threshold", f32),
        ("terrain_curriculum", i32), ("terrain_num_rows", i32), ("terrain_num_cols", i32), ("self_collision", i32),
        ("terrain_env_length", f32), ("max_episode_length_s", f32),
        ("base_init_state", f32 * 13), ("_padf4", f32),
        ("seed", u64),
    ]


_PF, _PU8, _PI64, _PI32, _PI16 = C.POINTER(f32), C.POINTER(u8), C.POINTER(i64), C.POINTER(i32), C.POINTER(i16)


class lg_rollout_buffers(C.Structure):
    """Rollout storage of ``lg_rollout_policy`` (include/legged_hip.h): [steps(+1)][N][...] device arrays owned by the caller."""
    _fields_ = [("steps", i32), ("obs", _PF), ("obs0", _PF), ("actions", _PF), ("mean", _PF), ("rew", _PF), ("dones", _PU8), ("time_outs", _PU8)]


LG_MAX_ROLL_STEPS = 256


class lg_buffers(C.Structure):
    _fields_ = [
        ("root_states", _PF), ("dof_state", _PF), ("contact_forces", _PF), ("obs_buf", _PF), ("rew_buf", _PF),
        ("reset_buf", _PU8), ("time_out_buf", _PU8), ("episode_length_buf", _PI64),
        ("torques", _PF), ("actions", _PF), ("last_actions", _PF), ("last_dof_vel", _PF),
        ("last_root_vel", _PF), ("commands", _PF), ("feet_air_time", _PF), ("last_contacts", _PU8),
        ("base_lin_vel", _PF), ("base_ang_vel", _PF), ("projected_gravity", _PF),
        ("measured_heights", _PF), ("sea_hidden_state", _PF), ("sea_cell_state", _PF),
        ("episode_sums", _PF), ("episode_means", _PF), ("extras_accum", _PF), ("step_counter", _PI64), ("env_origins", _PF),
        ("terrain_levels", _PI32), ("terrain_types", _PI32), ("terrain_origins", _PF),
        ("height_samples", _PI16), ("friction_coeffs", _PF), ("base_mass_delta", _PF),
    ]


BUFFER_FIELDS = [name for name, _ in lg_buffers._fields_]


# ----------------------------------------------------------------------------- packing
def _fill(dst, src):
    flat = np.ascontiguousarray(np.asarray(src, dtype=np.float64).ravel(), dtype=np.float32)   # keep alive during memmove
    if 4 * flat.size > C.sizeof(dst):
        raise ValueError("source larger than the destination field")
    C.memmove(dst, flat.ctypes.data, 4 * flat.size)


def _sym6(I):
    I = np.asarray(I)
    return [I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]]


def pack_model(model, foot_name: str, penalize_on, terminate_on, armature: float = 0.0) -> lg_robot_model:
    """``RobotModel`` (model compiler) -> ``lg_robot_model``; the body lookups by
    substring mirror legged_robot.py:696-702."""
    K, L, n = model.num_limbs, model.chain_len, model.num_dof
    if K * L != LG_MAX_DOF or K > LG_MAX_LIMBS or L > LG_MAX_CHAIN:
        raise ValueError(f"robot must be K limbs x L joints with K*L == 12, got K={K} L={L}")
    if model.num_bodies > LG_MAX_BODIES:
        raise ValueError("too many report bodies")
    m = lg_robot_model()
    m.num_limbs, m.chain_len, m.num_bodies = K, L, model.num_bodies
    m.base_mass = model.base_mass
    _fill(m.base_com, model.base_com)
    _fill(m.base_inertia, _sym6(model.base_inertia))
    _fill(m.joint_pos, model.joint_pos)
    _fill(m.joint_rot, model.joint_rot)
    _fill(m.joint_axis, model.joint_axis)
    _fill(m.body_mass, model.body_mass)
    _fill(m.body_com, model.body_com)
    _fill(m.body_inertia, [_sym6(I) for I in model.body_inertia])
    lo = np.where(model.dof_has_limits, model.dof_lower, 1.0)
    hi = np.where(model.dof_has_limits, model.dof_upper, -1.0)     # lower > upper encodes "no limit"
    _fill(m.dof_lower, lo)
    _fill(m.dof_upper, hi)
    _fill(m.dof_vel_limit, model.dof_velocity)
    _fill(m.dof_armature, np.full(n, armature))
    _fill(m.dof_damping, model.dof_damping)

    def put(dst, cp, joint):
        _fill(dst.pos, cp.pos)
        dst.radius, dst.report_body, dst.joint = cp.radius, cp.report_body, joint

    if len(model.base_points) > LG_MAX_BASE_POINTS:
        raise ValueError("too many base collision points")
    m.num_base_points = len(model.base_points)
    for i, cp in enumerate(model.base_points):
        put(m.base_points[i], cp, -1)
    for k in range(K):
        pts = model.limb_points[k]
        if len(pts) > LG_MAX_LIMB_POINTS:
            raise ValueError("too many limb collision points")
        m.num_limb_points[k] = len(pts)
        for i, cp in enumerate(pts):
            put(m.limb_points[k][i], cp, model.limb_point_joint[k][i])
    feet = model.bodies_matching(foot_name)
    if len(feet) != K:
        raise ValueError(f"expected one foot body per limb, found {len(feet)} for {foot_name!r}")
    for k, b in enumerate(feet):
        dyn = model.report_parent_dyn[b]
        if not (1 + k * L <= dyn <= (k + 1) * L):
            raise ValueError("foot bodies must be ordered like the limbs")
        m.foot_body[k] = b
    pen = sorted({b for s in penalize_on for b in model.bodies_matching(s)})
    ter = sorted({b for s in terminate_on for b in model.bodies_matching(s)})
    m.penalised_mask = sum(1 << b for b in pen)
    m.termination_mask = sum(1 << b for b in ter)
    return m


class lg_mlp_net(C.Structure):
    """include/legged_hip.h: lg_mlp_net (device pointers as integers)."""
    _fields_ = [("weights", C.c_void_p * 4), ("biases", C.c_void_p * 4), ("grad_weights", C.c_void_p * 4), ("grad_biases", C.c_void_p * 4),
                ("input", C.c_void_p), ("output", C.c_void_p), ("grad_output", C.c_void_p), ("dims", i32 * 5)]


class lg_ppo_batch(C.Structure):
    """include/legged_hip.h: lg_ppo_batch."""
    _fields_ = [(n, C.c_void_p) for n in ("actions", "old_log_prob", "old_mu", "old_sigma", "advantages", "old_values", "returns", "std")] + \
               [("clip", C.c_float), ("value_coef", C.c_float), ("entropy_coef", C.c_float), ("use_clipped_value", i32),
                ("d_std", C.c_void_p), ("stats", C.c_void_p), ("loss_acc", C.c_void_p)]


class lg_rollout_step(C.Structure):
    """include/legged_hip.h: lg_rollout_step."""
    _fields_ = [(n, C.c_void_p) for n in ("obs", "actions", "mean", "rewards", "dones", "time_outs", "storage_obs", "storage_actions",
                                          "storage_mu", "storage_rewards", "storage_dones", "storage_time_outs", "cur_return", "cur_length",
                                          "sums", "std", "storage_sigma", "storage_log_prob")] + [("num_envs", i32), ("num_obs", i32), ("num_actions", i32)]


class lg_rollout_post(C.Structure):
    """include/legged_hip.h: lg_rollout_post."""
    _fields_ = [(n, C.c_void_p) for n in ("actions", "mean", "rewards", "dones", "time_outs", "std", "sigma", "log_prob", "time_outs_f",
                                          "cur_return", "cur_length", "sums")] + [("steps", i32), ("num_envs", i32), ("num_actions", i32)]


class lg_adam_tensor(C.Structure):
    """include/legged_hip.h: lg_adam_tensor."""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("step", C.c_void_p),
                ("numel", i64)]


def struct_to_dict(s) -> Dict:
    out = {}
    for name, _ in s._fields_:
        v = getattr(s, name)
        out[name] = np.ctypeslib.as_array(v).copy() if hasattr(v, "_length_") else v
    return out


# ----------------------------------------------------------------------------- library
_LIB_NAME = "liblegged_hip.so"
_lib = None


def library_path() -> str:
    """In-tree liblegged_hip.so; LG_HIP_LIB selects another build of the same sources (tools/profile_sections.py)."""
    return os.environ.get("LG_HIP_LIB") or os.path.join(os.path.dirname(os.path.realpath(__file__)), "csrc", _LIB_NAME)


def bind_prototypes(lib, prefix: str):
    """Attach argtypes/restype for every entry point of legged_hip.h."""
    vp = C.c_void_p
    sig = {
        "create": ([C.POINTER(lg_params), C.POINTER(lg_robot_model), _PF, C.c_int, C.POINTER(vp)], C.c_int),
        "destroy": ([vp], None),
        "bind": ([vp, C.POINTER(lg_buffers)], C.c_int),
        "step": ([vp, vp, i64, vp], C.c_int),
        "reset_idx": ([vp, vp, i32, i64, vp], C.c_int),
        "actuator_forward": ([vp, vp, vp, vp, vp, vp, i32, vp], C.c_int),
        "physics_substep": ([vp, vp, i32, vp], C.c_int),
        "compute_observations_only": ([vp, i64, vp], C.c_int),
        "set_params": ([vp, C.POINTER(lg_params)], C.c_int),
        "set_obs_buffer": ([vp, vp], C.c_int),
        "last_error": ([], C.c_char_p),
        "abi_version": ([], C.c_int),
        "sizeof": ([C.c_int], C.c_int),
    }
    if prefix == "lg_":      # the fused actor is a product-only entry point (the oracle side of it is torch fp32)
        lib.lg_policy_create.argtypes = [C.POINTER(i32), C.POINTER(_PF), C.POINTER(_PF), _PF, C.c_int, C.POINTER(vp)]
        lib.lg_policy_create.restype = C.c_int
        lib.lg_policy_destroy.argtypes, lib.lg_policy_destroy.restype = [vp], None
        lib.lg_policy_act.argtypes = [vp, vp, vp, vp, i32, u64, i64, vp, i32, vp]
        lib.lg_policy_act.restype = C.c_int
        lib.lg_policy_load_device.argtypes = [vp, vp, vp, vp, vp]
        lib.lg_policy_load_device.restype = C.c_int
        lib.lg_step_policy.argtypes = [vp, vp, vp, vp, vp, u64, i32, i64, vp]
        lib.lg_step_policy.restype = C.c_int
        lib.lg_rollout_policy.argtypes = [vp, vp, C.POINTER(lg_rollout_buffers), u64, i32, i64, vp]
        lib.lg_rollout_policy.restype = C.c_int
        lib.lg_gae_returns.argtypes = [vp, vp, vp, vp, C.c_float, C.c_float, vp, vp, i32, i32, vp]
        lib.lg_gae_returns.restype = C.c_int
        lib.lg_ppo_loss.argtypes = [vp] * 11 + [C.c_float, C.c_float, C.c_float, i32, vp, vp, vp, vp, i32, i32, vp]
        lib.lg_ppo_loss.restype = C.c_int
        lib.lg_mlp_forward.argtypes = [C.POINTER(lg_mlp_net), i32, vp, i32, vp]
        lib.lg_mlp_forward.restype = C.c_int
        lib.lg_mlp_workspace_bytes.argtypes = [C.POINTER(lg_mlp_net), i32]
        lib.lg_mlp_workspace_bytes.restype = C.c_size_t
        lib.lg_mlp_backward.argtypes = [C.POINTER(lg_mlp_net), i32, vp, i32, vp, C.c_size_t, vp]
        lib.lg_mlp_backward.restype = C.c_int
        lib.lg_mlp_wide_workspace_bytes.argtypes = [C.POINTER(lg_mlp_net), i32, i32]
        lib.lg_mlp_wide_workspace_bytes.restype = C.c_size_t
        lib.lg_mlp_wide_set_precision.argtypes, lib.lg_mlp_wide_set_precision.restype = [C.c_int], C.c_int
        lib.lg_mlp_wide_forward.argtypes = [C.POINTER(lg_mlp_net), i32, vp, i32, vp, C.c_size_t, vp]
        lib.lg_mlp_wide_forward.restype = C.c_int
        lib.lg_mlp_wide_backward.argtypes = [C.POINTER(lg_mlp_net), i32, vp, i32, vp, C.c_size_t, vp]
        lib.lg_mlp_wide_backward.restype = C.c_int
        lib.lg_ppo_minibatch.argtypes = [C.POINTER(lg_mlp_net), vp, i32, C.POINTER(lg_ppo_batch), vp, C.c_size_t, vp]
        lib.lg_ppo_minibatch.restype = C.c_int
        lib.lg_adam_step.argtypes = [C.POINTER(lg_adam_tensor), i32, vp, C.c_float, C.c_float, C.c_float, C.c_float, vp, C.c_float, vp, vp]
        lib.lg_adam_step.restype = C.c_int
        lib.lg_rollout_finish.argtypes, lib.lg_rollout_finish.restype = [C.POINTER(lg_rollout_post), vp], C.c_int
        lib.lg_rollout_record.argtypes = [C.POINTER(lg_rollout_step), vp]
        lib.lg_rollout_record.restype = C.c_int
        lib.lg_mlp_trace.argtypes, lib.lg_mlp_trace.restype = [vp], None
        lib.lg_set_deferred_extras.argtypes, lib.lg_set_deferred_extras.restype = [vp, i32], C.c_int
        lib.lg_extras_flush.argtypes, lib.lg_extras_flush.restype = [vp, i64, vp], C.c_int
        lib.lg_resample_reset_commands.argtypes, lib.lg_resample_reset_commands.restype = [vp, i64, vp], C.c_int
        lib.lg_device_status.argtypes, lib.lg_device_status.restype = [vp, i32], C.c_int
        lib.lg_clear_device_status.argtypes, lib.lg_clear_device_status.restype = [vp], C.c_int
        lib.lg_debug_handover.argtypes, lib.lg_debug_handover.restype = [vp, i32, i32], C.c_int
    for name, (args, res) in sig.items():
        fn = getattr(lib, prefix + name)
        fn.argtypes, fn.restype = args, res
    sizeof = getattr(lib, prefix + "sizeof")
    structs = (lg_params, lg_robot_model, lg_buffers, lg_point) + ((lg_mlp_net, lg_adam_tensor, lg_rollout_step, lg_ppo_batch) if prefix == "lg_" else ())
    for which, st in enumerate(structs):
        if sizeof(which) != C.sizeof(st):
            raise RuntimeError(f"struct layout mismatch for {st.__name__}: C {sizeof(which)} vs ctypes {C.sizeof(st)}")
    return lib


EXPORTED_SYMBOLS = ["lg_create", "lg_destroy", "lg_bind", "lg_step", "lg_reset_idx", "lg_actuator_forward",
                    "lg_physics_substep", "lg_compute_observations_only", "lg_set_params", "lg_last_error",
                    "lg_abi_version", "lg_sizeof", "lg_set_obs_buffer", "lg_policy_create", "lg_policy_destroy", "lg_policy_act",
                    "lg_step_policy", "lg_gae_returns", "lg_ppo_loss", "lg_policy_load_device", "lg_mlp_forward",
                    "lg_mlp_workspace_bytes", "lg_mlp_backward", "lg_mlp_wide_workspace_bytes", "lg_mlp_wide_forward", "lg_mlp_wide_backward", "lg_mlp_wide_set_precision", "lg_adam_step", "lg_rollout_record", "lg_mlp_trace", "lg_ppo_minibatch", "lg_set_deferred_extras", "lg_extras_flush",
                    "lg_device_status", "lg_clear_device_status", "lg_debug_handover", "lg_rollout_policy", "lg_rollout_finish", "lg_resample_reset_commands"]


def load_library():
    """Load the HIP extension or fail loudly -- never a CPU substitute."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.isfile(path):
        raise RuntimeError(
            f"HIP extension {path} is not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
    _lib = bind_prototypes(C.CDLL(path), "lg_")
    if _lib.lg_abi_version() != LG_ABI_VERSION:
        raise RuntimeError("liblegged_hip.so ABI version mismatch; rebuild")
    return _lib


class Sim:
    """Owner of one ``lg_sim`` handle (one per GPU).  ``lib``/``prefix`` are
    parameters only so the tests can drive the CPU oracle through the same class."""

    def __init__(self, params: lg_params, model: lg_robot_model, actuator_weights: Optional[np.ndarray],
                 device_id: int = 0, lib=None, prefix: str = "lg_"):
        self.lib = lib if lib is not None else load_library()
        self.prefix = prefix
        self.params, self.model = params, model
        self._w = None
        wptr = None
        if actuator_weights is not None:
            self._w = np.ascontiguousarray(actuator_weights, dtype=np.float32)
            if self._w.size != LG_ACTUATOR_FLOATS:
                raise ValueError("actuator weight blob must hold 972 floats")
            wptr = self._w.ctypes.data_as(_PF)
        self.handle = C.c_void_p()
        self._check(self._fn("create")(C.byref(params), C.byref(model), wptr, int(device_id), C.byref(self.handle)))
        self.buffers = lg_buffers()

    def _fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"{self.prefix}* call failed ({rc}): {self._fn('last_error')().decode()}")

    def bind(self, pointers: Dict[str, int]):
        for name in BUFFER_FIELDS:
            ptr = pointers.get(name, 0) or 0
            setattr(self.buffers, name, C.cast(C.c_void_p(ptr), dict(lg_buffers._fields_)[name]))
        self._check(self._fn("bind")(self.handle, C.byref(self.buffers)))

    def step(self, actions_ptr: int, common_step_counter: int, stream: int = 0):
        self._check(self._fn("step")(self.handle, actions_ptr, int(common_step_counter), stream))

    def step_policy(self, policy_handle, obs_ptr: int, actions_ptr: int, mean_ptr, seed: int, deterministic: bool,
                    common_step_counter: int, stream: int = 0):
        self._check(self.lib.lg_step_policy(self.handle, policy_handle, obs_ptr, actions_ptr, mean_ptr, int(seed), int(bool(deterministic)),
                                            int(common_step_counter), stream))

    def rollout_policy(self, policy_handle, steps: int, obs_ptr: int, actions_ptr: int, mean_ptr, rew_ptr: int, dones_ptr: int, time_outs_ptr: int,
                       seed: int, deterministic: bool, common_step_counter: int, stream: int = 0, obs0_ptr=None):
        r = lg_rollout_buffers()
        r.steps = int(steps)
        for name, ptr in (("obs", obs_ptr), ("obs0", obs0_ptr), ("actions", actions_ptr), ("mean", mean_ptr), ("rew", rew_ptr), ("dones", dones_ptr), ("time_outs", time_outs_ptr)):
            setattr(r, name, C.cast(C.c_void_p(ptr or 0), dict(lg_rollout_buffers._fields_)[name]))
        self._check(self.lib.lg_rollout_policy(self.handle, policy_handle, C.byref(r), int(seed), int(bool(deterministic)), int(common_step_counter), stream))

    def set_deferred_extras(self, on: bool):
        self._check(self.lib.lg_set_deferred_extras(self.handle, int(bool(on))))

    def extras_flush(self, common_step_counter: int, stream: int = 0):
        self._check(self.lib.lg_extras_flush(self.handle, int(common_step_counter), stream))

    def reset_idx(self, ids_ptr: int, count: int, common_step_counter: int, stream: int = 0):
        self._check(self._fn("reset_idx")(self.handle, ids_ptr, int(count), int(common_step_counter), stream))

    def actuator_forward(self, pos_err, vel, torques, hidden, cell, rows, stream=0):
        self._check(self._fn("actuator_forward")(self.handle, pos_err, vel, torques, hidden, cell, int(rows), stream))

    def physics_substep(self, torques_ptr, write_contacts=1, stream=0):
        self._check(self._fn("physics_substep")(self.handle, torques_ptr, int(write_contacts), stream))

    def compute_observations_only(self, common_step_counter, stream=0):
        self._check(self._fn("compute_observations_only")(self.handle, int(common_step_counter), stream))

    def set_obs_buffer(self, ptr: int):
        self._check(self._fn("set_obs_buffer")(self.handle, ptr))

    def set_params(self, params: lg_params):
        self.params = params
        self._check(self._fn("set_params")(self.handle, C.byref(params)))

    def resample_reset_commands(self, common_step_counter: int, stream: int = 0):
        self._check(self.lib.lg_resample_reset_commands(self.handle, int(common_step_counter), stream))

    def device_status(self, synchronize: bool = True) -> int:
        """``lg_device_status``: the sticky status word of the handle (0 = clean); see include/legged_hip.h."""
        return int(self.lib.lg_device_status(self.handle, int(bool(synchronize))))

    def clear_device_status(self):
        self._check(self.lib.lg_clear_device_status(self.handle))

    def debug_handover(self, skip: int, spin_limit: int = 0):
        self._check(self.lib.lg_debug_handover(self.handle, int(skip), int(spin_limit)))

    def close(self):
        if self.handle:
            self._fn("destroy")(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
