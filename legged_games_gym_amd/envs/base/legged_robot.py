"""``LeggedRobot`` -- the VecEnv the reference exposes, on top of the fused HIP step.

Surface mirror of reference ``legged_gym/envs/base/legged_robot.py``: same
constructor signature, same attribute names / shapes / dtypes (SURVEY.md 8b
"outer boundary"), same ``step`` / ``reset`` / ``reset_idx`` /
``get_observations`` contract, so ``rsl_rl.OnPolicyRunner`` (or the bundled
``legged_games_gym_amd.rl`` runner) drops in unchanged.

What differs, by design: the reference runs ``step`` as a Python loop of ~150
eager torch kernels around 4 PhysX calls (:80-137).  Here ``step`` is ONE call
through the C-ABI (``lg_step``); every per-step quantity below is a view of a
buffer the kernel wrote.  Python keeps only what is genuinely host-side:
config parsing (:781-791), reward bookkeeping names (:583-607), creation-time
domain randomisation (:261-327), env origins (:752-779) and the ``extras``
dictionary (:179-191).

Deliberate deviations from reference behaviour are listed in DESIGN.md
("Quirks"): ``reset_buf``/``time_out_buf`` are persistent bool tensors (Q1,
Q4), RNG is counter-based Philox rather than torch's global stream.
"""
import os

import numpy as np
import torch

from legged_games_gym_amd import LEGGED_GYM_ROOT_DIR, capi
from legged_games_gym_amd.device_sim import DeviceSim
from legged_games_gym_amd.envs.base.base_task import BaseTask
from legged_games_gym_amd.utils import packing
from legged_games_gym_amd.utils.helpers import class_to_dict

# Captured rollouts leave extras["episode"] of a step to the NEXT step's launch and end with one flush node ("deferred
# extras"); LG_DEFER_EXTRAS=0 keeps the in-launch finisher in graphs too.
_DEFER_EXTRAS = os.environ.get("LG_DEFER_EXTRAS", "1") != "0"
from legged_games_gym_amd.utils.model_compiler import load_model
from legged_games_gym_amd.utils.terrain import Terrain


def command_curriculum_update(mean_episode_sum, max_episode_length, tracking_scale_dt, lin_vel_x, max_curriculum):
    """``update_command_curriculum`` (reference :471-483): widen the ``lin_vel_x`` range by 0.5 on both sides, clipped at
    ``max_curriculum``, when the mean tracking reward of the reset envs exceeds 80 % of its maximum.  Pure host arithmetic;
    pinned by tests/golden/command_curriculum.npz (the reference's own method)."""
    lo, hi = float(lin_vel_x[0]), float(lin_vel_x[1])
    if mean_episode_sum / max_episode_length > 0.8 * tracking_scale_dt:
        lo = float(np.clip(lo - 0.5, -max_curriculum, 0.0))
        hi = float(np.clip(hi + 0.5, 0.0, max_curriculum))
    return [lo, hi]


class LeggedRobot(BaseTask):
    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless):
        self.cfg = cfg
        self.sim_params = sim_params
        self.height_samples = None
        self.debug_viz = False
        self.init_done = False
        self._parse_cfg(self.cfg)
        super().__init__(self.cfg, sim_params, physics_engine, sim_device, headless)
        self._init_buffers()
        self._prepare_reward_function()
        self.init_done = True

    # ------------------------------------------------------------------ hot path
    def step(self, actions):
        """One policy step for all envs (reference :80-104), a single fused launch."""
        self.common_step_counter += 1                     # :115 (the kernel needs the incremented value)
        # The reference builds a NEW obs_buf every step (:215) and rsl_rl's PPO.act keeps a reference to the previous one
        # until process_env_step: alternate between two buffers so the tensor returned last step is not overwritten.
        self._obs_flip ^= 1
        self.obs_buf = self._obs_pair[self._obs_flip]
        self._sim.set_obs_output(self.obs_buf)
        # while a caller captures a multi-step HIP graph (rl/runner.py) the counter must come from the device
        self._sim.step(actions, -1 if self._capturing else self.common_step_counter)
        if self.cfg.commands.curriculum:
            self._command_curriculum_tick()
        return self.obs_buf, self.privileged_obs_buf, self.rew_buf, self.reset_buf, self.extras

    def step_policy(self, fused_actor, deterministic=False):
        """Rollout step with the actor fused in: ``actions = fused_actor(obs_buf)`` and ``step(actions)`` as ONE launch
        (``lg_step_policy``).  Returns ``(actions, mean), (obs, privileged_obs, rew, dones, extras)``.  Only for the
        compiled fused shape (flat ANYmal actor on the plane); raises RuntimeError otherwise -- use ``step``."""
        if self.cfg.commands.curriculum and self._capturing:
            raise NotImplementedError("commands.curriculum needs eager steps (host-side rule between steps)")
        self.common_step_counter += 1
        prev_obs = self.obs_buf
        self._obs_flip ^= 1
        self.obs_buf = self._obs_pair[self._obs_flip]
        self._sim.set_obs_output(self.obs_buf)
        try:
            am = self._sim.step_policy(fused_actor, prev_obs, -1 if self._capturing else self.common_step_counter, deterministic)
        except Exception:
            self.common_step_counter -= 1
            self._obs_flip ^= 1
            self.obs_buf = prev_obs
            self._sim.set_obs_output(self.obs_buf)
            raise
        if self.cfg.commands.curriculum:
            self._command_curriculum_tick()
        return am, (self.obs_buf, self.privileged_obs_buf, self.rew_buf, self.reset_buf, self.extras)

    def post_physics_step(self):
        raise RuntimeError("post_physics_step is fused into lg_step; call step()")

    def _command_curriculum_tick(self):
        """reset_idx :159-168 for resets that happen INSIDE the fused step: every ``max_episode_length`` policy steps, if envs
        were reset by this step, apply ``update_command_curriculum`` to them (host rule, one device read per 1000 steps).
        The kernel has already zeroed the episode sums of those envs; their mean is what it published in ``episode_means``
        (= mean over the reset envs / max_episode_length_s, :179-183)."""
        if self._capturing:
            raise NotImplementedError("commands.curriculum is a host-side rule evaluated between steps: it cannot be captured into a "
                                      "multi-step HIP graph (use eager steps; the bundled runner falls back by itself)")
        if self.common_step_counter % self.max_episode_length != 0 or "tracking_lin_vel" not in self.episode_sums:
            return
        if not bool(self.reset_buf.any()):
            return
        i = self.reward_names_all.index("tracking_lin_vel")
        mean_sum = float(self._episode_means[i]) * self.max_episode_length_s
        new = command_curriculum_update(mean_sum, self.max_episode_length, self.reward_scales["tracking_lin_vel"],
                                        self.command_ranges["lin_vel_x"], self.cfg.commands.max_curriculum)
        if new != list(self.command_ranges["lin_vel_x"]):
            self.command_ranges["lin_vel_x"][:] = new
            self.set_command_ranges()
            # reference order (:159-176): the envs this step reset draw their commands AFTER the widening
            self._sim.sim.resample_reset_commands(self.common_step_counter, torch.cuda.current_stream(self.device).cuda_stream)
        self.extras["episode"]["max_command_x"] = self.command_ranges["lin_vel_x"][1]

    def begin_graph_capture(self):
        """Prepare for capturing several step() calls into one HIP graph: the step counter moves to the device."""
        if self.cfg.commands.curriculum:
            raise NotImplementedError("commands.curriculum needs eager steps (host-side rule between steps)")
        self._sim.buf["step_counter"].fill_(self.common_step_counter)
        self._capturing = True
        self._sim.set_deferred_extras(_DEFER_EXTRAS)      # captured steps leave extras["episode"] to the next launch ...

    def capture_extras_flush(self):
        """Last node of a captured rollout: publish the extras["episode"] of its last step (``lg_extras_flush``)."""
        if _DEFER_EXTRAS:
            self._sim.flush_extras(-1)

    def end_graph_capture(self, steps_captured: int):
        self._capturing = False
        self._sim.set_deferred_extras(False)               # ... eager steps publish them before they return, as the reference does
        self.common_step_counter -= steps_captured      # capture does not execute; replays are accounted by the caller

    def make_graphed_step(self, policy_act, warmup=3, steps_per_replay=1):
        """Capture ``actions = policy_act(obs_buf); step(actions)`` into one HIP graph and return a
        zero-argument callable that replays it (launch-bound inner loop -> one hipGraphLaunch).
        The step counter lives on the device while replaying (``lg_step(..., -1)``), the host copy is
        advanced alongside.  ``policy_act`` must be capturable (no host syncs) and read ``self.obs_buf``."""
        if self.cfg.commands.curriculum:
            raise NotImplementedError("commands.curriculum needs eager steps (host-side rule between steps)")
        sim = self._sim
        sim.set_obs_output(self.obs_buf)                  # single fixed buffer while replaying (the policy reads it inside the graph)
        sim.buf["step_counter"].fill_(self.common_step_counter)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                sim.step(policy_act(self.obs_buf), -1)
                self.common_step_counter += 1
        torch.cuda.current_stream(self.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        sim.set_deferred_extras(_DEFER_EXTRAS)
        try:
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                for _ in range(steps_per_replay):                # several policy steps per hipGraphLaunch: no host in between
                    sim.step(policy_act(self.obs_buf), -1)
                if _DEFER_EXTRAS:
                    sim.flush_extras(-1)                         # extras["episode"] of the replay's last step
        finally:
            sim.set_deferred_extras(False)
        self._step_graph = graph

        def replay():
            graph.replay()
            self.common_step_counter += steps_per_replay
            return self.obs_buf, self.privileged_obs_buf, self.rew_buf, self.reset_buf, self.extras
        return replay

    def make_graphed_policy_step(self, fused_actor, warmup=3, steps_per_replay=1):
        """Like ``make_graphed_step`` with the actor fused into the step kernel: the graph is ONE ``lg_step_policy`` launch
        (obs_buf -> actions -> next obs_buf, in place).  Raises RuntimeError when the sim / actor pair has no fused kernel."""
        if self.cfg.commands.curriculum:
            raise NotImplementedError("commands.curriculum needs eager steps (host-side rule between steps)")
        sim = self._sim
        sim.set_obs_output(self.obs_buf)
        sim.buf["step_counter"].fill_(self.common_step_counter)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                sim.step_policy(fused_actor, self.obs_buf, -1)
                self.common_step_counter += 1
        torch.cuda.current_stream(self.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        sim.set_deferred_extras(_DEFER_EXTRAS)
        try:
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                for _ in range(steps_per_replay):
                    sim.step_policy(fused_actor, self.obs_buf, -1)
                if _DEFER_EXTRAS:
                    sim.flush_extras(-1)
        finally:
            sim.set_deferred_extras(False)
        self._step_graph = graph

        def replay():
            graph.replay()
            self.common_step_counter += steps_per_replay
            return self.obs_buf, self.privileged_obs_buf, self.rew_buf, self.reset_buf, self.extras
        return replay

    def rollout_policy(self, fused_actor, steps, storage=None, deterministic=False):
        """``steps`` rollout steps (``actions = actor(obs) + std * eps``; ``step(actions)``) as ONE launch (``lg_rollout_policy``): what
        the caller's ``for i in range(num_steps_per_env)`` loop does, with identical results, but every workgroup walks through the
        steps of its own envs without meeting the others at each step boundary.  Returns the rollout storage: ``obs`` [T+1, N, num_obs]
        (``obs[0]`` = the observations before the first step), ``actions`` / ``mean`` [T, N, num_actions], ``rew`` [T, N], bool ``dones`` /
        ``time_outs`` [T, N].  ``self.obs_buf`` becomes ``obs[T]``; ``rew_buf`` / ``reset_buf`` / ``time_out_buf`` hold the last step's
        values as after ``step()``.  Raises RuntimeError when the sim / actor pair is not the compiled fused shape."""
        if self.cfg.commands.curriculum:
            raise NotImplementedError("commands.curriculum is a host-side rule evaluated between steps: use step()")
        N, T = self.num_envs, int(steps)
        if storage is None:
            dev, f32 = self.device, torch.float32
            with torch.inference_mode(False):         # obs_buf becomes a view of this storage: keep it usable by autograd modules
                storage = {"obs": torch.empty(T + 1, N, self.num_obs, device=dev, dtype=f32), "actions": torch.empty(T, N, self.num_actions, device=dev, dtype=f32),
                           "mean": torch.empty(T, N, self.num_actions, device=dev, dtype=f32), "rew": torch.empty(T, N, device=dev, dtype=f32),
                           "dones": torch.empty(T, N, device=dev, dtype=torch.bool), "time_outs": torch.empty(T, N, device=dev, dtype=torch.bool)}
        # the first step starts from the current observations: the kernel reads them where they are (and copies them to obs[0])
        obs0 = self.obs_buf if storage["obs"][0].data_ptr() != self.obs_buf.data_ptr() else None
        if obs0 is not None and not (obs0.is_contiguous() and obs0.dtype == torch.float32):
            storage["obs"][0].copy_(obs0); obs0 = None
        self._sim.rollout_policy(fused_actor, storage, -1 if self._capturing else self.common_step_counter + 1, deterministic, obs0=obs0)
        self.common_step_counter += T
        with torch.inference_mode(False):
            self.obs_buf = storage["obs"][T]
        return storage

    def make_graphed_rollout(self, fused_actor, steps, warmup=1):
        """``rollout_policy(fused_actor, steps)`` on a fixed storage captured into one HIP graph (ONE kernel node: the multi-step kernel starts from
        the previous replay's ``obs[steps]``, copies it to ``obs[0]``, and its last workgroup publishes ``extras["episode"]``): returns ``(replay, storage)``."""
        if self.cfg.commands.curriculum:
            raise NotImplementedError("commands.curriculum needs eager steps (host-side rule between steps)")
        sim = self._sim
        storage = None
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):               # (the first call allocates the library's workspace: outside capture)
                storage = self.rollout_policy(fused_actor, steps, storage=storage)
        torch.cuda.current_stream(self.device).wait_stream(side)
        sim.buf["step_counter"].fill_(self.common_step_counter)
        graph = torch.cuda.CUDAGraph()
        T = int(steps)
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            sim.rollout_policy(fused_actor, storage, -1, False, obs0=storage["obs"][T])     # continues from the previous replay's last observations
        self._rollout_graph = graph
        self.obs_buf = storage["obs"][T]

        def replay():
            graph.replay()
            self.common_step_counter += T
            return storage
        return replay, storage

    def reset_idx(self, env_ids):
        """Reset the listed envs (reference :147-191) through ``lg_reset_idx``."""
        if len(env_ids) == 0:
            return
        if self.cfg.commands.curriculum and (self.common_step_counter % self.max_episode_length == 0):
            self.update_command_curriculum(env_ids)
        if not self.init_done and self._params.terrain_curriculum:
            # "don't change on initial reset" (:453-455)
            self._params.terrain_curriculum = 0
            self._sim.sim.set_params(self._params)
            self._sim.reset_idx(env_ids, self.common_step_counter)
            self._params.terrain_curriculum = 1
            self._sim.sim.set_params(self._params)
        else:
            self._sim.reset_idx(env_ids, self.common_step_counter)
        if self.cfg.commands.curriculum:
            self.extras["episode"]["max_command_x"] = self.command_ranges["lin_vel_x"][1]

    def _init_extras(self):
        """extras["episode"] / extras["time_outs"] (:179-191).  The means are computed by the library (the last workgroup
        of k_step / k_extras) into ``episode_means``; the dict holds 0-dim views, so step() issues no torch kernel and
        no host sync for them.  They stay stale on steps without resets (quirk Q4)."""
        ep = self.extras.setdefault("episode", {})
        for i, name in enumerate(self.reward_names_all):
            ep["rew_" + name] = self._episode_means[i]
        if self.cfg.terrain.curriculum:
            ep["terrain_level"] = self._episode_means[len(self.reward_names_all)]
        if self.cfg.commands.curriculum:
            ep["max_command_x"] = self.command_ranges["lin_vel_x"][1]
        if self.cfg.env.send_timeouts:
            self.extras["time_outs"] = self.time_out_buf

    def compute_observations(self):
        """Recompute obs_buf from the current state (reference :212-230)."""
        self._sim.set_obs_output(self.obs_buf)
        self._sim.compute_observations_only(self.common_step_counter)

    # ------------------------------------------------------------------ creation
    def create_sim(self):
        """Terrain + robot model + device buffers (reference :232-251, :657-750)."""
        self.up_axis_idx = 2
        mesh_type = self.cfg.terrain.mesh_type
        if mesh_type not in (None, "none", "plane", "heightfield", "trimesh"):
            raise ValueError("Terrain mesh type not recognised. Allowed types are [None, plane, heightfield, trimesh]")
        self.terrain = None
        if mesh_type in ("heightfield", "trimesh"):
            # 'trimesh' collides against the same height samples: the built-in engine has one
            # terrain representation, the int16 height field the reference also feeds to _get_heights.
            self.terrain = Terrain(self.cfg.terrain, self.num_envs)
        self._create_envs()

    def _create_envs(self):
        cfg = self.cfg
        asset_path = cfg.asset.file.format(LEGGED_GYM_ROOT_DIR=LEGGED_GYM_ROOT_DIR)
        self.robot_model = load_model(asset_path, collapse_fixed_joints=cfg.asset.collapse_fixed_joints,
                                      replace_cylinder_with_capsule=cfg.asset.replace_cylinder_with_capsule) \
            if os.path.isfile(asset_path) else load_model(asset_path)
        rm = self.robot_model
        self.num_dof = self.num_dofs = rm.num_dof
        self.num_bodies = rm.num_bodies
        self.dof_names = list(rm.dof_names)
        body_names = list(rm.body_names)
        if self.num_actions != self.num_dof:
            raise ValueError("num_actions must equal the number of DOFs")

        seed = getattr(cfg, "seed", 1)
        sim_dt = self.sim_params.dt
        gravity = getattr(self.sim_params, "gravity", [0.0, 0.0, -9.81])
        physx = getattr(self.sim_params, "physx", None)
        contact_offset = getattr(physx, "contact_offset", 0.01) if physx is not None else 0.01
        self._params, self.reward_names_all = packing.build_params(
            cfg, rm, sim_dt, self.num_envs, seed if seed is not None and seed >= 0 else 1, gravity=gravity,
            terrain=self.terrain, contact_offset=contact_offset)
        # asset.self_collisions = 0 (e.g. anymal_c_flat, anymal_c_flat_config.py:42): links of one robot collide (DESIGN.md "Self-collision")
        self.self_collision_modelled = bool(self._params.self_collision)
        self._model = capi.pack_model(rm, cfg.asset.foot_name, cfg.asset.penalize_contacts_on,
                                      cfg.asset.terminate_after_contacts_on, armature=cfg.asset.armature)
        weights = None
        if self._params.control_type == capi.CTRL["actuator_net"]:
            weights = packing.load_actuator_weights(
                cfg.control.actuator_net_file.format(LEGGED_GYM_ROOT_DIR=LEGGED_GYM_ROOT_DIR))
        hs = self.terrain.heightsamples if self.terrain is not None else None
        to = self.terrain.env_origins if self.terrain is not None else None
        self._sim = DeviceSim(self._params, self._model, rm, torch.device(self.device), weights, hs, to)
        b = self._sim.buf

        # body index lookups by substring (:696-702, :733-750)
        def idx(names):
            out = []
            for n in names:
                out.extend(rm.bodies_matching(n))
            return torch.tensor(out, dtype=torch.long, device=self.device)
        self.feet_indices = torch.tensor(rm.bodies_matching(cfg.asset.foot_name), dtype=torch.long, device=self.device)
        self.penalised_contact_indices = idx(cfg.asset.penalize_contacts_on)
        self.termination_contact_indices = idx(cfg.asset.terminate_after_contacts_on)
        self.body_names = body_names

        # DOF limits (:299-314)
        lo = torch.tensor(rm.dof_lower, dtype=torch.float, device=self.device)
        hi = torch.tensor(rm.dof_upper, dtype=torch.float, device=self.device)
        m, r = (lo + hi) / 2, hi - lo
        self.dof_pos_limits = torch.stack((m - 0.5 * r * cfg.rewards.soft_dof_pos_limit,
                                           m + 0.5 * r * cfg.rewards.soft_dof_pos_limit), dim=1)
        self.dof_vel_limits = torch.tensor(rm.dof_velocity, dtype=torch.float, device=self.device)
        self.torque_limits = torch.tensor(rm.dof_effort, dtype=torch.float, device=self.device)

        init = cfg.init_state
        self.base_init_state = torch.tensor(list(init.pos) + list(init.rot) + list(init.lin_vel) + list(init.ang_vel),
                                            dtype=torch.float, device=self.device)
        self._get_env_origins()

        # creation-time domain randomisation (:261-285, :316-327): 64 friction buckets, per-env base mass
        if cfg.domain_rand.randomize_friction:
            fr = cfg.domain_rand.friction_range
            bucket_ids = torch.randint(0, 64, (self.num_envs, 1))
            buckets = (fr[1] - fr[0]) * torch.rand(64, 1) + fr[0]
            self.friction_coeffs = buckets[bucket_ids]                       # [N,1,1] on the host, as in the reference
            b["friction_coeffs"].copy_(self.friction_coeffs.view(-1).to(self.device))
        if cfg.domain_rand.randomize_base_mass:
            rng = cfg.domain_rand.added_mass_range
            dm = np.array([np.random.uniform(rng[0], rng[1]) for _ in range(self.num_envs)], dtype=np.float32)
            b["base_mass_delta"].copy_(torch.from_numpy(dm).to(self.device))
        if self.terrain is not None:
            self.height_samples = self._sim.buf["height_samples"].view(self.terrain.tot_rows, self.terrain.tot_cols)
            b["terrain_levels"].copy_(self.terrain_levels.to(torch.int32))
            b["terrain_types"].copy_(self.terrain_types.to(torch.int32))
            self.terrain_levels = b["terrain_levels"]
            self.terrain_types = b["terrain_types"]

        # initial actor poses (:714-719): origin + U(-1,1) xy (also on the plane, Q8); DOFs at the default pose
        root = b["root_states"]
        root[:] = self.base_init_state
        root[:, :3] += self.env_origins
        root[:, :2] += 2.0 * torch.rand(self.num_envs, 2, device=self.device) - 1.0
        q0 = torch.tensor([init.default_joint_angles[n] for n in self.dof_names], dtype=torch.float, device=self.device)
        b["dof_state"].view(self.num_envs, self.num_dof, 2)[..., 0] = q0

    def _get_env_origins(self):
        """Reference :752-779."""
        cfg = self.cfg
        origins = self._sim.buf["env_origins"]
        if cfg.terrain.mesh_type in ("heightfield", "trimesh"):
            self.custom_origins = True
            max_init_level = cfg.terrain.max_init_terrain_level
            if not cfg.terrain.curriculum:
                max_init_level = cfg.terrain.num_rows - 1
            self.terrain_levels = torch.randint(0, max_init_level + 1, (self.num_envs,), device=self.device)
            self.terrain_types = torch.div(torch.arange(self.num_envs, device=self.device),
                                           (self.num_envs / cfg.terrain.num_cols), rounding_mode="floor").to(torch.long)
            self.max_terrain_level = cfg.terrain.num_rows
            self.terrain_origins = torch.from_numpy(self.terrain.env_origins).to(self.device).to(torch.float)
            origins[:] = self.terrain_origins[self.terrain_levels, self.terrain_types]
        else:
            self.custom_origins = False
            num_cols = np.floor(np.sqrt(self.num_envs))
            num_rows = np.ceil(self.num_envs / num_cols)
            xx, yy = torch.meshgrid(torch.arange(num_rows), torch.arange(num_cols), indexing="ij")
            spacing = cfg.env.env_spacing
            origins[:, 0] = spacing * xx.flatten()[:self.num_envs].to(self.device)
            origins[:, 1] = spacing * yy.flatten()[:self.num_envs].to(self.device)
            origins[:, 2] = 0.0
        self.env_origins = origins

    # ------------------------------------------------------------------ buffers
    def _init_buffers(self):
        """Expose the kernel's buffers under the reference's names (:511-581)."""
        b = self._sim.buf
        N, n = self.num_envs, self.num_dof
        self.root_states = b["root_states"]
        self.dof_state = b["dof_state"]
        self.dof_pos = self.dof_state.view(N, n, 2)[..., 0]
        self.dof_vel = self.dof_state.view(N, n, 2)[..., 1]
        self.base_quat = self.root_states[:, 3:7]
        self.contact_forces = b["contact_forces"]
        self.obs_buf, self.rew_buf = b["obs_buf"], b["rew_buf"]
        self._obs_pair = (b["obs_buf"], torch.zeros_like(b["obs_buf"]))
        self._obs_flip = 0
        self._capturing = False
        self.reset_buf, self.time_out_buf = b["reset_buf"], b["time_out_buf"]
        self.reset_buf.fill_(True)                                   # base_task.py:73 starts at ones
        self.episode_length_buf = b["episode_length_buf"]
        self.common_step_counter = 0
        self.extras = {}
        self.noise_scale_vec = self._get_noise_scale_vec(self.cfg)
        self.gravity_vec = torch.tensor([0.0, 0.0, -1.0], device=self.device).repeat((N, 1))
        self.forward_vec = torch.tensor([1.0, 0.0, 0.0], device=self.device).repeat((N, 1))
        self.torques, self.actions = b["torques"], b["actions"]
        self.last_actions, self.last_dof_vel, self.last_root_vel = b["last_actions"], b["last_dof_vel"], b["last_root_vel"]
        self.commands = b["commands"]
        s = self.obs_scales
        self.commands_scale = torch.tensor([s.lin_vel, s.lin_vel, s.ang_vel], device=self.device, requires_grad=False)
        self.feet_air_time, self.last_contacts = b["feet_air_time"], b["last_contacts"]
        self.base_lin_vel, self.base_ang_vel, self.projected_gravity = b["base_lin_vel"], b["base_ang_vel"], b["projected_gravity"]
        self.projected_gravity[:] = self.gravity_vec
        if self.cfg.terrain.measure_heights:
            self.height_points = self._init_height_points()
            self.measured_heights = b["measured_heights"]
        else:
            self.measured_heights = 0
        p = self._params
        self.p_gains = torch.tensor(list(p.p_gains)[:n], dtype=torch.float, device=self.device)
        self.d_gains = torch.tensor(list(p.d_gains)[:n], dtype=torch.float, device=self.device)
        self.default_dof_pos = torch.tensor(list(p.default_dof_pos)[:n], dtype=torch.float, device=self.device).unsqueeze(0)
        for i, name in enumerate(self.dof_names):
            if not any(k in name for k in self.cfg.control.stiffness.keys()) and self.cfg.control.control_type in ["P", "V"]:
                print(f"PD gain of joint {name} were not defined, setting them to zero")

    def _prepare_reward_function(self):
        """Reference :583-607: zero scales dropped, the rest multiplied by dt; episode sums per name."""
        for key in list(self.reward_scales.keys()):
            if self.reward_scales[key] == 0:
                self.reward_scales.pop(key)
            else:
                self.reward_scales[key] *= self.dt
        self.reward_names = [n for n in self.reward_scales.keys() if n != "termination"]
        assert list(self.reward_scales.keys()) == self.reward_names_all
        sums = self._sim.buf["episode_sums"]
        self.episode_sums = {name: sums[i] for i, name in enumerate(self.reward_names_all)}
        self._episode_means = self._sim.buf["episode_means"]
        self._init_extras()

    def _parse_cfg(self, cfg):
        """Reference :781-791."""
        self.dt = self.cfg.control.decimation * self.sim_params.dt
        self.obs_scales = self.cfg.normalization.obs_scales
        self.reward_scales = class_to_dict(self.cfg.rewards.scales)
        self.command_ranges = class_to_dict(self.cfg.commands.ranges)
        if self.cfg.terrain.mesh_type not in ["heightfield", "trimesh"]:
            self.cfg.terrain.curriculum = False
        self.max_episode_length_s = self.cfg.env.episode_length_s
        self.max_episode_length = np.ceil(self.max_episode_length_s / self.dt)
        self.cfg.domain_rand.push_interval = np.ceil(self.cfg.domain_rand.push_interval_s / self.dt)

    def _get_noise_scale_vec(self, cfg):
        """Reference :485-508 (the kernel applies the same per-segment scales)."""
        noise_vec = torch.zeros(self.num_obs, device=self.device)
        self.add_noise = self.cfg.noise.add_noise
        ns, lvl, s = self.cfg.noise.noise_scales, self.cfg.noise.noise_level, self.obs_scales
        noise_vec[:3] = ns.lin_vel * lvl * s.lin_vel
        noise_vec[3:6] = ns.ang_vel * lvl * s.ang_vel
        noise_vec[6:9] = ns.gravity * lvl
        noise_vec[9:12] = 0.0
        noise_vec[12:24] = ns.dof_pos * lvl * s.dof_pos
        noise_vec[24:36] = ns.dof_vel * lvl * s.dof_vel
        noise_vec[36:48] = 0.0
        if self.cfg.terrain.measure_heights:
            noise_vec[48:] = ns.height_measurements * lvl * s.height_measurements
        return noise_vec

    def _init_height_points(self):
        """Reference :815-829."""
        y = torch.tensor(self.cfg.terrain.measured_points_y, device=self.device, requires_grad=False)
        x = torch.tensor(self.cfg.terrain.measured_points_x, device=self.device, requires_grad=False)
        grid_x, grid_y = torch.meshgrid(x, y, indexing="ij")
        self.num_height_points = grid_x.numel()
        points = torch.zeros(self.num_envs, self.num_height_points, 3, device=self.device, requires_grad=False)
        points[:, :, 0] = grid_x.flatten()
        points[:, :, 1] = grid_y.flatten()
        return points

    # ------------------------------------------------------------------ curricula (host side)
    def update_command_curriculum(self, env_ids):
        """Reference :471-483; pushes the widened range to the device."""
        if torch.mean(self.episode_sums["tracking_lin_vel"][env_ids]) / self.max_episode_length > 0.8 * \
                self.reward_scales["tracking_lin_vel"]:
            r = self.command_ranges["lin_vel_x"]
            r[0] = np.clip(r[0] - 0.5, -self.cfg.commands.max_curriculum, 0.0)
            r[1] = np.clip(r[1] + 0.5, 0.0, self.cfg.commands.max_curriculum)
            self.set_command_ranges()

    def set_command_ranges(self):
        """Re-upload ``self.command_ranges`` (tooling such as play.py edits them)."""
        capi._fill(self._params.cmd_lin_vel_x, self.command_ranges["lin_vel_x"])
        capi._fill(self._params.cmd_lin_vel_y, self.command_ranges["lin_vel_y"])
        capi._fill(self._params.cmd_ang_vel_yaw, self.command_ranges["ang_vel_yaw"])
        capi._fill(self._params.cmd_heading, self.command_ranges["heading"])
        self._sim.sim.set_params(self._params)

    def set_fixed_commands(self, vx, vy, yaw):
        """Benchmark helper (BASELINE.json: "fixed command"): pin the command and disable resampling."""
        self.command_ranges["lin_vel_x"] = [vx, vx]
        self.command_ranges["lin_vel_y"] = [vy, vy]
        self.command_ranges["ang_vel_yaw"] = [yaw, yaw]
        self.set_command_ranges()
        self.commands[:, 0], self.commands[:, 1], self.commands[:, 2] = vx, vy, yaw
