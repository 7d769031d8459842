"""All in-scope configuration trees in one place.

API mirror (same class / attribute names, same values) of the reference's
``legged_gym/envs/base/legged_robot_config.py:33-244``,
``envs/anymal_c/mixed_terrains/anymal_c_rough_config.py:33-94``,
``envs/anymal_c/flat/anymal_c_flat_config.py:33-74`` and
``envs/cassie/cassie_config.py:33-112``.  The reference-path modules
(``envs/base/legged_robot_config.py`` ...) re-export from here.  Values are
pinned against the reference by ``tests/golden/configs.json`` (fixture G2).

Every value is data the env or the PPO runner consumes; nothing here touches
the simulator.
"""
from .base.base_config import BaseConfig

_ROOT = "{LEGGED_GYM_ROOT_DIR}"


def _sym_grid(half_extent_dm):
    """[-h, ..., h] in 0.1 m steps, written as the same literals as the
    reference (-0.8, -0.7, ...) so float equality holds."""
    return [round(0.1 * i, 1) + 0.0 for i in range(-half_extent_dm, half_extent_dm + 1)]


# --------------------------------------------------------------------------- #
#  generic legged robot (legged_robot_config.py:33-200)
# --------------------------------------------------------------------------- #
class LeggedRobotCfg(BaseConfig):
    class env:
        num_envs = 4096
        num_observations = 235
        num_privileged_obs = None   # None -> step() returns None for the critic obs
        num_actions = 12
        env_spacing = 3.0           # plane only
        send_timeouts = True
        episode_length_s = 20

    class terrain:
        mesh_type = "trimesh"       # none | plane | heightfield | trimesh
        horizontal_scale = 0.1
        vertical_scale = 0.005
        border_size = 25
        curriculum = True
        static_friction = 1.0
        dynamic_friction = 1.0
        restitution = 0.0
        measure_heights = True
        measured_points_x = _sym_grid(8)    # 17 points, 1.6 m
        measured_points_y = _sym_grid(5)    # 11 points, 1.0 m
        selected = False
        terrain_kwargs = None
        max_init_terrain_level = 5
        terrain_length = 8.0
        terrain_width = 8.0
        num_rows = 10
        num_cols = 20
        terrain_proportions = [0.1, 0.1, 0.35, 0.25, 0.2]
        slope_treshold = 0.75

    class commands:
        curriculum = False
        max_curriculum = 1.0
        num_commands = 4
        resampling_time = 10.0
        heading_command = True

        class ranges:
            lin_vel_x = [-1.0, 1.0]
            lin_vel_y = [-1.0, 1.0]
            ang_vel_yaw = [-1, 1]
            heading = [-3.14, 3.14]

    class init_state:
        pos = [0.0, 0.0, 0.42]
        rot = [0.0, 0.0, 0.0, 1.0]     # xyzw
        lin_vel = [0.0, 0.0, 0.0]
        ang_vel = [0.0, 0.0, 0.0]
        default_joint_angles = {"joint_a": 0.0, "joint_b": 0.0}

    class control:
        control_type = "P"             # P | V | T
        stiffness = {"joint_a": 10.0, "joint_b": 15.0}
        damping = {"joint_a": 1.0, "joint_b": 1.5}
        action_scale = 0.5
        decimation = 4

    class asset:
        file = ""
        name = "legged_robot"
        foot_name = "None"
        penalize_contacts_on = []
        terminate_after_contacts_on = []
        disable_gravity = False
        collapse_fixed_joints = True
        fix_base_link = False
        default_dof_drive_mode = 3
        self_collisions = 0
        replace_cylinder_with_capsule = True
        flip_visual_attachments = True
        density = 0.001
        angular_damping = 0.0
        linear_damping = 0.0
        max_angular_velocity = 1000.0
        max_linear_velocity = 1000.0
        armature = 0.0
        thickness = 0.01

    class domain_rand:
        randomize_friction = True
        friction_range = [0.5, 1.25]
        randomize_base_mass = False
        added_mass_range = [-1.0, 1.0]
        push_robots = True
        push_interval_s = 15
        max_push_vel_xy = 1.0

    class rewards:
        class scales:
            termination = -0.0
            tracking_lin_vel = 1.0
            tracking_ang_vel = 0.5
            lin_vel_z = -2.0
            ang_vel_xy = -0.05
            orientation = -0.0
            torques = -0.00001
            dof_vel = -0.0
            dof_acc = -2.5e-7
            base_height = -0.0
            feet_air_time = 1.0
            collision = -1.0
            feet_stumble = -0.0
            action_rate = -0.01
            stand_still = -0.0

        only_positive_rewards = True
        tracking_sigma = 0.25
        soft_dof_pos_limit = 1.0
        soft_dof_vel_limit = 1.0
        soft_torque_limit = 1.0
        base_height_target = 1.0
        max_contact_force = 100.0

    class normalization:
        class obs_scales:
            lin_vel = 2.0
            ang_vel = 0.25
            dof_pos = 1.0
            dof_vel = 0.05
            height_measurements = 5.0

        clip_observations = 100.0
        clip_actions = 100.0

    class noise:
        add_noise = True
        noise_level = 1.0

        class noise_scales:
            dof_pos = 0.01
            dof_vel = 1.5
            lin_vel = 0.1
            ang_vel = 0.2
            gravity = 0.05
            height_measurements = 0.1

    class viewer:
        ref_env = 0
        pos = [10, 0, 6]
        lookat = [11.0, 5, 3.0]

    class sim:
        dt = 0.005
        substeps = 1
        gravity = [0.0, 0.0, -9.81]
        up_axis = 1

        class physx:
            num_threads = 10
            solver_type = 1
            num_position_iterations = 4
            num_velocity_iterations = 0
            contact_offset = 0.01
            rest_offset = 0.0
            bounce_threshold_velocity = 0.5
            max_depenetration_velocity = 1.0
            max_gpu_contact_pairs = 2 ** 23
            default_buffer_size_multiplier = 5
            contact_collection = 2


class LeggedRobotCfgPPO(BaseConfig):
    seed = 1
    runner_class_name = "OnPolicyRunner"

    class policy:
        init_noise_std = 1.0
        actor_hidden_dims = [512, 256, 128]
        critic_hidden_dims = [512, 256, 128]
        activation = "elu"

    class algorithm:
        value_loss_coef = 1.0
        use_clipped_value_loss = True
        clip_param = 0.2
        entropy_coef = 0.01
        num_learning_epochs = 5
        num_mini_batches = 4
        learning_rate = 1.0e-3
        schedule = "adaptive"
        gamma = 0.99
        lam = 0.95
        desired_kl = 0.01
        max_grad_norm = 1.0

    class runner:
        policy_class_name = "ActorCritic"
        algorithm_class_name = "PPO"
        num_steps_per_env = 24
        max_iterations = 1500
        save_interval = 50
        experiment_name = "test"
        run_name = ""
        resume = False
        load_run = -1
        checkpoint = -1
        resume_path = None


# --------------------------------------------------------------------------- #
#  ANYmal-C rough / flat (anymal_c_rough_config.py, anymal_c_flat_config.py)
# --------------------------------------------------------------------------- #
def _anymal_default_angles():
    q = {}
    for leg, sgn in (("LF", 1.0), ("LH", -1.0), ("RF", 1.0), ("RH", -1.0)):
        q[f"{leg}_HAA"] = 0.0 if leg[0] == "L" else -0.0
        q[f"{leg}_HFE"] = 0.4 * sgn
        q[f"{leg}_KFE"] = -0.8 * sgn
    return q


class AnymalCRoughCfg(LeggedRobotCfg):
    class env(LeggedRobotCfg.env):
        num_envs = 4096
        num_actions = 12

    class terrain(LeggedRobotCfg.terrain):
        mesh_type = "trimesh"

    class init_state(LeggedRobotCfg.init_state):
        pos = [0.0, 0.0, 0.6]
        default_joint_angles = _anymal_default_angles()

    class control(LeggedRobotCfg.control):
        stiffness = {"HAA": 80.0, "HFE": 80.0, "KFE": 80.0}
        damping = {"HAA": 2.0, "HFE": 2.0, "KFE": 2.0}
        action_scale = 0.5
        decimation = 4
        use_actuator_network = True
        actuator_net_file = _ROOT + "/resources/actuator_nets/anydrive_v3_lstm.pt"

    class asset(LeggedRobotCfg.asset):
        file = _ROOT + "/resources/robots/anymal_c/urdf/anymal_c.urdf"
        name = "anymal_c"
        foot_name = "FOOT"
        penalize_contacts_on = ["SHANK", "THIGH"]
        terminate_after_contacts_on = ["base"]
        self_collisions = 1     # 1 = disabled (bitwise filter)

    class domain_rand(LeggedRobotCfg.domain_rand):
        randomize_base_mass = True
        added_mass_range = [-5.0, 5.0]

    class rewards(LeggedRobotCfg.rewards):
        base_height_target = 0.5
        max_contact_force = 500.0
        only_positive_rewards = True

        class scales(LeggedRobotCfg.rewards.scales):
            pass


class AnymalCRoughCfgPPO(LeggedRobotCfgPPO):
    class runner(LeggedRobotCfgPPO.runner):
        run_name = ""
        experiment_name = "rough_anymal_c"
        load_run = -1


class AnymalCFlatCfg(AnymalCRoughCfg):
    class env(AnymalCRoughCfg.env):
        num_observations = 48

    class terrain(AnymalCRoughCfg.terrain):
        mesh_type = "plane"
        measure_heights = False

    class asset(AnymalCRoughCfg.asset):
        self_collisions = 0     # 0 = enabled

    class rewards(AnymalCRoughCfg.rewards):
        max_contact_force = 350.0

        class scales(AnymalCRoughCfg.rewards.scales):
            orientation = -5.0
            torques = -0.000025
            feet_air_time = 2.0

    class commands(AnymalCRoughCfg.commands):
        heading_command = False
        resampling_time = 4.0

        class ranges(AnymalCRoughCfg.commands.ranges):
            ang_vel_yaw = [-1.5, 1.5]

    class domain_rand(AnymalCRoughCfg.domain_rand):
        # ground-plane friction is combined by averaging: mu = (mu_foot + 1)/2
        friction_range = [0.0, 1.5]


class AnymalCFlatCfgPPO(AnymalCRoughCfgPPO):
    class policy(AnymalCRoughCfgPPO.policy):
        actor_hidden_dims = [128, 64, 32]
        critic_hidden_dims = [128, 64, 32]
        activation = "elu"

    class algorithm(AnymalCRoughCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(AnymalCRoughCfgPPO.runner):
        run_name = ""
        experiment_name = "flat_anymal_c"
        load_run = -1
        max_iterations = 300


# --------------------------------------------------------------------------- #
#  Cassie (cassie_config.py:33-112)
# --------------------------------------------------------------------------- #
def _cassie_default_angles():
    q = {}
    for side, sgn in (("left", 1.0), ("right", -1.0)):
        q[f"hip_abduction_{side}"] = 0.1 * sgn
        q[f"hip_rotation_{side}"] = 0.0
        q[f"hip_flexion_{side}"] = 1.0
        q[f"thigh_joint_{side}"] = -1.8
        q[f"ankle_joint_{side}"] = 1.57
        q[f"toe_joint_{side}"] = -1.57
    return q


class CassieRoughCfg(LeggedRobotCfg):
    class env(LeggedRobotCfg.env):
        num_envs = 4096
        num_observations = 169
        num_actions = 12

    class terrain(LeggedRobotCfg.terrain):
        measured_points_x = _sym_grid(5)    # 11 x 11 = 121 points
        measured_points_y = _sym_grid(5)

    class init_state(LeggedRobotCfg.init_state):
        pos = [0.0, 0.0, 1.0]
        default_joint_angles = _cassie_default_angles()

    class control(LeggedRobotCfg.control):
        stiffness = {"hip_abduction": 100.0, "hip_rotation": 100.0, "hip_flexion": 200.0,
                     "thigh_joint": 200.0, "ankle_joint": 200.0, "toe_joint": 40.0}
        damping = {"hip_abduction": 3.0, "hip_rotation": 3.0, "hip_flexion": 6.0,
                   "thigh_joint": 6.0, "ankle_joint": 6.0, "toe_joint": 1.0}
        action_scale = 0.5
        decimation = 4

    class asset(LeggedRobotCfg.asset):
        file = _ROOT + "/resources/robots/cassie/urdf/cassie.urdf"
        name = "cassie"
        foot_name = "toe"
        terminate_after_contacts_on = ["pelvis"]
        flip_visual_attachments = False
        self_collisions = 1

    class rewards(LeggedRobotCfg.rewards):
        soft_dof_pos_limit = 0.95
        soft_dof_vel_limit = 0.9
        soft_torque_limit = 0.9
        max_contact_force = 300.0
        only_positive_rewards = False

        class scales(LeggedRobotCfg.rewards.scales):
            termination = -200.0
            tracking_ang_vel = 1.0
            torques = -5.0e-6
            dof_acc = -2.0e-7
            lin_vel_z = -0.5
            feet_air_time = 5.0
            dof_pos_limits = -1.0
            no_fly = 0.25
            dof_vel = -0.0
            ang_vel_xy = -0.0
            feet_contact_forces = -0.0


class CassieRoughCfgPPO(LeggedRobotCfgPPO):
    class runner(LeggedRobotCfgPPO.runner):
        run_name = ""
        experiment_name = "rough_cassie"

    class algorithm(LeggedRobotCfgPPO.algorithm):
        entropy_coef = 0.01
