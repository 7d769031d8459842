"""G4: torch transcriptions of the reference's torch-side half, used to pin the C oracle.

Each function follows the cited lines of reference legged_gym/envs/base/legged_robot.py (and
cassie.py / utils/math.py) with the same torch ops on the same tensor shapes; the [EXTERNAL]
isaacgym.torch_utils helpers come from legged_games_gym_amd.utils.math (standard definitions).
RNG-consuming pieces (command resampling, resets, observation noise) are excluded here and
covered by distribution tests.
"""
import numpy as np
import torch

from legged_games_gym_amd.utils.math import quat_apply, quat_apply_yaw, quat_rotate_inverse, wrap_to_pi


class TorchSideRef:
    def __init__(self, cfg, robot, params, reward_scales, state):
        self.cfg, self.robot, self.p = cfg, robot, params
        self.dt = cfg.control.decimation * cfg.sim.dt
        self.reward_scales = reward_scales              # name -> scale*dt, alphabetical, zero scales removed
        for k, v in state.items():
            setattr(self, k, v.clone() if torch.is_tensor(v) else v)
        self.num_envs = self.root_states.shape[0]
        n = robot.num_dof
        self.dof_pos = self.dof_state.view(self.num_envs, n, 2)[..., 0]
        self.dof_vel = self.dof_state.view(self.num_envs, n, 2)[..., 1]
        self.feet_indices = torch.tensor(robot.bodies_matching(cfg.asset.foot_name))
        pen = [b for s in cfg.asset.penalize_contacts_on for b in robot.bodies_matching(s)]
        ter = [b for s in cfg.asset.terminate_after_contacts_on for b in robot.bodies_matching(s)]
        self.penalised_contact_indices = torch.tensor(pen, dtype=torch.long)
        self.termination_contact_indices = torch.tensor(ter, dtype=torch.long)
        self.max_episode_length = np.ceil(cfg.env.episode_length_s / self.dt)
        self.default_dof_pos = torch.tensor([cfg.init_state.default_joint_angles[k] for k in robot.dof_names], dtype=torch.float).unsqueeze(0)
        lo, hi = torch.tensor(robot.dof_lower, dtype=torch.float), torch.tensor(robot.dof_upper, dtype=torch.float)
        m, r = (lo + hi) / 2, hi - lo
        self.dof_pos_limits = torch.stack((m - 0.5 * r * cfg.rewards.soft_dof_pos_limit, m + 0.5 * r * cfg.rewards.soft_dof_pos_limit), dim=1)
        self.dof_vel_limits = torch.tensor(robot.dof_velocity, dtype=torch.float)
        self.torque_limits = torch.tensor(robot.dof_effort, dtype=torch.float)
        self.gravity_vec = torch.tensor([0.0, 0.0, -1.0]).repeat(self.num_envs, 1)
        self.forward_vec = torch.tensor([1.0, 0.0, 0.0]).repeat(self.num_envs, 1)
        s = cfg.normalization.obs_scales
        self.obs_scales = s
        self.commands_scale = torch.tensor([s.lin_vel, s.lin_vel, s.ang_vel])
        self.episode_sums = {k: torch.zeros(self.num_envs) for k in reward_scales}
        self.measured_heights = 0

    # ---- legged_robot.py:106-137 without the RNG consumers
    def post_physics_step(self):
        self.episode_length_buf += 1
        self.base_quat = self.root_states[:, 3:7]
        self.base_lin_vel = quat_rotate_inverse(self.base_quat, self.root_states[:, 7:10])
        self.base_ang_vel = quat_rotate_inverse(self.base_quat, self.root_states[:, 10:13])
        self.projected_gravity = quat_rotate_inverse(self.base_quat, self.gravity_vec)
        # _post_physics_step_callback :329-345 (no resample / push in this transcription)
        if self.cfg.commands.heading_command:
            forward = quat_apply(self.base_quat, self.forward_vec)
            heading = torch.atan2(forward[:, 1], forward[:, 0])
            self.commands[:, 2] = torch.clip(0.5 * wrap_to_pi(self.commands[:, 3] - heading), -1.0, 1.0)
        if self.cfg.terrain.measure_heights:
            self.measured_heights = self._get_heights()
        self.check_termination()
        self.compute_reward()
        self.compute_observations()
        self.last_actions = self.actions.clone()
        self.last_dof_vel = self.dof_vel.clone()
        self.last_root_vel = self.root_states[:, 7:13].clone()

    def check_termination(self):          # :139-145
        self.reset_buf = torch.any(torch.norm(self.contact_forces[:, self.termination_contact_indices, :], dim=-1) > 1.0, dim=1)
        self.time_out_buf = self.episode_length_buf > self.max_episode_length
        self.reset_buf |= self.time_out_buf

    def compute_reward(self):             # :193-210
        self.rew_buf = torch.zeros(self.num_envs)
        for name, scale in self.reward_scales.items():
            if name == "termination":
                continue
            rew = getattr(self, "_reward_" + name)() * scale
            self.rew_buf += rew
            self.episode_sums[name] += rew
        if self.cfg.rewards.only_positive_rewards:
            self.rew_buf[:] = torch.clip(self.rew_buf[:], min=0.0)
        if "termination" in self.reward_scales:
            rew = self._reward_termination() * self.reward_scales["termination"]
            self.rew_buf += rew
            self.episode_sums["termination"] += rew

    def compute_observations(self):       # :212-230 (noise excluded) + clip :100-101
        self.obs_buf = torch.cat((self.base_lin_vel * self.obs_scales.lin_vel, self.base_ang_vel * self.obs_scales.ang_vel,
                                  self.projected_gravity, self.commands[:, :3] * self.commands_scale,
                                  (self.dof_pos - self.default_dof_pos) * self.obs_scales.dof_pos,
                                  self.dof_vel * self.obs_scales.dof_vel, self.actions), dim=-1)
        if self.cfg.terrain.measure_heights:
            heights = torch.clip(self.root_states[:, 2].unsqueeze(1) - 0.5 - self.measured_heights, -1, 1.0) * self.obs_scales.height_measurements
            self.obs_buf = torch.cat((self.obs_buf, heights), dim=-1)
        c = self.cfg.normalization.clip_observations
        self.obs_buf = torch.clip(self.obs_buf, -c, c)

    def _get_heights(self):               # :831-869
        if self.cfg.terrain.mesh_type == "plane":
            return torch.zeros(self.num_envs, self.num_height_points)
        points = quat_apply_yaw(self.base_quat.repeat(1, self.num_height_points), self.height_points) + (self.root_states[:, :3]).unsqueeze(1)
        points += self.cfg.terrain.border_size
        points = (points / self.cfg.terrain.horizontal_scale).long()
        px = torch.clip(points[:, :, 0].view(-1), 0, self.height_samples.shape[0] - 2)
        py = torch.clip(points[:, :, 1].view(-1), 0, self.height_samples.shape[1] - 2)
        heights = torch.min(torch.min(self.height_samples[px, py], self.height_samples[px + 1, py]), self.height_samples[px, py + 1])
        return heights.view(self.num_envs, -1) * self.cfg.terrain.vertical_scale

    # ---- reward terms :872-969, cassie.py:43-46
    def _reward_lin_vel_z(self): return torch.square(self.base_lin_vel[:, 2])
    def _reward_ang_vel_xy(self): return torch.sum(torch.square(self.base_ang_vel[:, :2]), dim=1)
    def _reward_orientation(self): return torch.sum(torch.square(self.projected_gravity[:, :2]), dim=1)
    def _reward_base_height(self):
        base_height = torch.mean(self.root_states[:, 2].unsqueeze(1) - self.measured_heights, dim=1)
        return torch.square(base_height - self.cfg.rewards.base_height_target)
    def _reward_torques(self): return torch.sum(torch.square(self.torques), dim=1)
    def _reward_dof_vel(self): return torch.sum(torch.square(self.dof_vel), dim=1)
    def _reward_dof_acc(self): return torch.sum(torch.square((self.last_dof_vel - self.dof_vel) / self.dt), dim=1)
    def _reward_action_rate(self): return torch.sum(torch.square(self.last_actions - self.actions), dim=1)
    def _reward_collision(self):
        return torch.sum(1.0 * (torch.norm(self.contact_forces[:, self.penalised_contact_indices, :], dim=-1) > 0.1), dim=1)
    def _reward_termination(self): return self.reset_buf * ~self.time_out_buf
    def _reward_dof_pos_limits(self):
        out = -(self.dof_pos - self.dof_pos_limits[:, 0]).clip(max=0.0)
        out += (self.dof_pos - self.dof_pos_limits[:, 1]).clip(min=0.0)
        return torch.sum(out, dim=1)
    def _reward_dof_vel_limits(self):
        return torch.sum((torch.abs(self.dof_vel) - self.dof_vel_limits * self.cfg.rewards.soft_dof_vel_limit).clip(min=0.0, max=1.0), dim=1)
    def _reward_torque_limits(self):
        return torch.sum((torch.abs(self.torques) - self.torque_limits * self.cfg.rewards.soft_torque_limit).clip(min=0.0), dim=1)
    def _reward_tracking_lin_vel(self):
        err = torch.sum(torch.square(self.commands[:, :2] - self.base_lin_vel[:, :2]), dim=1)
        return torch.exp(-err / self.cfg.rewards.tracking_sigma)
    def _reward_tracking_ang_vel(self):
        err = torch.square(self.commands[:, 2] - self.base_ang_vel[:, 2])
        return torch.exp(-err / self.cfg.rewards.tracking_sigma)
    def _reward_feet_air_time(self):
        contact = self.contact_forces[:, self.feet_indices, 2] > 1.0
        contact_filt = torch.logical_or(contact, self.last_contacts)
        self.last_contacts = contact
        first_contact = (self.feet_air_time > 0.0) * contact_filt
        self.feet_air_time += self.dt
        rew = torch.sum((self.feet_air_time - 0.5) * first_contact, dim=1)
        rew *= torch.norm(self.commands[:, :2], dim=1) > 0.1
        self.feet_air_time *= ~contact_filt
        return rew
    def _reward_stumble(self):
        return torch.any(torch.norm(self.contact_forces[:, self.feet_indices, :2], dim=2) > 5 * torch.abs(self.contact_forces[:, self.feet_indices, 2]), dim=1)
    def _reward_stand_still(self):
        return torch.sum(torch.abs(self.dof_pos - self.default_dof_pos), dim=1) * (torch.norm(self.commands[:, :2], dim=1) < 0.1)
    def _reward_feet_contact_forces(self):
        return torch.sum((torch.norm(self.contact_forces[:, self.feet_indices, :], dim=-1) - self.cfg.rewards.max_contact_force).clip(min=0.0), dim=1)
    def _reward_no_fly(self):
        contacts = self.contact_forces[:, self.feet_indices, 2] > 0.1
        return 1.0 * (torch.sum(1.0 * contacts, dim=1) == 1)


def pd_torques(cfg, p_gains, d_gains, default_dof_pos, torque_limits, actions, dof_pos, dof_vel, last_dof_vel, sim_dt):
    """legged_robot.py:371-395."""
    actions_scaled = actions * cfg.control.action_scale
    ct = cfg.control.control_type
    if ct == "P":
        torques = p_gains * (actions_scaled + default_dof_pos - dof_pos) - d_gains * dof_vel
    elif ct == "V":
        torques = p_gains * (actions_scaled - dof_vel) - d_gains * (dof_vel - last_dof_vel) / sim_dt
    elif ct == "T":
        torques = actions_scaled
    else:
        raise NameError(f"Unknown controller type: {ct}")
    return torch.clip(torques, -torque_limits, torque_limits)
