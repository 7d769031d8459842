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
    class asset:
        angular_damping = 0.0
        armature = 0.0
        collapse_fixed_joints = True
        default_dof_drive_mode = 3
        density = 0.001
        disable_gravity = False
        file = ""
        fix_base_link = False
        flip_visual_attachments = True
        foot_name = "None"
        linear_damping = 0.0
        max_angular_velocity = 1000.0
        max_linear_velocity = 1000.0
        name = "legged_robot"
        penalize_contacts_on = []
        replace_cylinder_with_capsule = True
        self_collisions = 0
        terminate_after_contacts_on = []
        thickness = 0.01

    class commands:
        curriculum = False
        heading_command = True
        max_curriculum = 1.0
        num_commands = 4
        resampling_time = 10.0

        class ranges:
            ang_vel_yaw = [-1, 1]
            heading = [-3.14, 3.14]
            lin_vel_x = [-1.0, 1.0]
            lin_vel_y = [-1.0, 1.0]

    class control:
        action_scale = 0.5
        control_type = "P"             # P | V | T
        damping = {"joint_a": 1.0, "joint_b": 1.5}
        decimation = 4
        stiffness = {"joint_a": 10.0, "joint_b": 15.0}

    class domain_rand:
        added_mass_range = [-1.0, 1.0]
        friction_range = [0.5, 1.25]
        max_push_vel_xy = 1.0
        push_interval_s = 15
        push_robots = True
        randomize_base_mass = False
        randomize_friction = True

    class env:
        env_spacing = 3.0           # plane only
        episode_length_s = 20
        num_actions = 12
        num_envs = 4096
        num_observations = 235
        num_privileged_obs = None   # None -> step() returns None for the critic obs
        send_timeouts = True

    class init_state:
        ang_vel = [0.0, 0.0, 0.0]
        default_joint_angles = {"joint_a": 0.0, "joint_b": 0.0}
        lin_vel = [0.0, 0.0, 0.0]
        pos = [0.0, 0.0, 0.42]
        rot = [0.0, 0.0, 0.0, 1.0]     # xyzw

    class noise:
        add_noise = True
        noise_level = 1.0

        class noise_scales:
            ang_vel = 0.2
            dof_pos = 0.01
            dof_vel = 1.5
            gravity = 0.05
            height_measurements = 0.1
            lin_vel = 0.1

    class normalization:
        clip_actions = 100.0
        clip_observations = 100.0

        class obs_scales:
            ang_vel = 0.25
            dof_pos = 1.0
            dof_vel = 0.05
            height_measurements = 5.0
            lin_vel = 2.0

    class rewards:
        base_height_target = 1.0
        max_contact_force = 100.0
        only_positive_rewards = True
        soft_dof_pos_limit = 1.0
        soft_dof_vel_limit = 1.0
        soft_torque_limit = 1.0
        tracking_sigma = 0.25

        class scales:
            action_rate = -0.01
            ang_vel_xy = -0.05
            base_height = -0.0
            collision = -1.0
            dof_acc = -2.5e-7
            dof_vel = -0.0
            feet_air_time = 1.0
            feet_stumble = -0.0
            lin_vel_z = -2.0
            orientation = -0.0
            stand_still = -0.0
            termination = -0.0
            torques = -0.00001
            tracking_ang_vel = 0.5
            tracking_lin_vel = 1.0

    class sim:
        dt = 0.005
        gravity = [0.0, 0.0, -9.81]
        substeps = 1
        up_axis = 1

        class physx:
            bounce_threshold_velocity = 0.5
            contact_collection = 2
            contact_offset = 0.01
            default_buffer_size_multiplier = 5
            max_depenetration_velocity = 1.0
            max_gpu_contact_pairs = 2 ** 23
            num_position_iterations = 4
            num_threads = 10
            num_velocity_iterations = 0
            rest_offset = 0.0
            solver_type = 1

    class terrain:
        border_size = 25
        curriculum = True
        dynamic_friction = 1.0
        horizontal_scale = 0.1
        max_init_terrain_level = 5
        measure_heights = True
        measured_points_x = _sym_grid(8)    # 17 points, 1.6 m
        measured_points_y = _sym_grid(5)    # 11 points, 1.0 m
        mesh_type = "trimesh"       # none | plane | heightfield | trimesh
        num_cols = 20
        num_rows = 10
        restitution = 0.0
        selected = False
        slope_treshold = 0.75
        static_friction = 1.0
        terrain_kwargs = None
        terrain_length = 8.0
        terrain_proportions = [0.1, 0.1, 0.35, 0.25, 0.2]
        terrain_width = 8.0
        vertical_scale = 0.005

    class viewer:
        lookat = [11.0, 5, 3.0]
        pos = [10, 0, 6]
        ref_env = 0


class LeggedRobotCfgPPO(BaseConfig):
    runner_class_name = "OnPolicyRunner"
    seed = 1

    class algorithm:
        clip_param = 0.2
        desired_kl = 0.01
        entropy_coef = 0.01
        gamma = 0.99
        lam = 0.95
        learning_rate = 1.0e-3
        max_grad_norm = 1.0
        num_learning_epochs = 5
        num_mini_batches = 4
        schedule = "adaptive"
        use_clipped_value_loss = True
        value_loss_coef = 1.0

    class policy:
        activation = "elu"
        actor_hidden_dims = [512, 256, 128]
        critic_hidden_dims = [512, 256, 128]
        init_noise_std = 1.0

    class runner:
        algorithm_class_name = "PPO"
        checkpoint = -1
        experiment_name = "test"
        load_run = -1
        max_iterations = 1500
        num_steps_per_env = 24
        policy_class_name = "ActorCritic"
        resume = False
        resume_path = None
        run_name = ""
        save_interval = 50


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
    class asset(LeggedRobotCfg.asset):
        file = _ROOT + "/resources/robots/anymal_c/urdf/anymal_c.urdf"
        foot_name = "FOOT"
        name = "anymal_c"
        penalize_contacts_on = ["SHANK", "THIGH"]
        self_collisions = 1     # 1 = disabled (bitwise filter)
        terminate_after_contacts_on = ["base"]

    class control(LeggedRobotCfg.control):
        action_scale = 0.5
        actuator_net_file = _ROOT + "/resources/actuator_nets/anydrive_v3_lstm.pt"
        damping = {"HAA": 2.0, "HFE": 2.0, "KFE": 2.0}
        decimation = 4
        stiffness = {"HAA": 80.0, "HFE": 80.0, "KFE": 80.0}
        use_actuator_network = True

    class domain_rand(LeggedRobotCfg.domain_rand):
        added_mass_range = [-5.0, 5.0]
        randomize_base_mass = True

    class env(LeggedRobotCfg.env):
        num_actions = 12
        num_envs = 4096

    class init_state(LeggedRobotCfg.init_state):
        default_joint_angles = _anymal_default_angles()
        pos = [0.0, 0.0, 0.6]

    class rewards(LeggedRobotCfg.rewards):
        base_height_target = 0.5
        max_contact_force = 500.0
        only_positive_rewards = True

        class scales(LeggedRobotCfg.rewards.scales):
            pass

    class terrain(LeggedRobotCfg.terrain):
        mesh_type = "trimesh"


class AnymalCRoughCfgPPO(LeggedRobotCfgPPO):
    class runner(LeggedRobotCfgPPO.runner):
        experiment_name = "rough_anymal_c"
        load_run = -1
        run_name = ""


class AnymalCFlatCfg(AnymalCRoughCfg):
    class asset(AnymalCRoughCfg.asset):
        self_collisions = 0     # 0 = enabled

    class commands(AnymalCRoughCfg.commands):
        heading_command = False
        resampling_time = 4.0

        class ranges(AnymalCRoughCfg.commands.ranges):
            ang_vel_yaw = [-1.5, 1.5]

    class domain_rand(AnymalCRoughCfg.domain_rand):
        # ground-plane friction is combined by averaging: mu = (mu_foot + 1)/2
        friction_range = [0.0, 1.5]

    class env(AnymalCRoughCfg.env):
        num_observations = 48

    class rewards(AnymalCRoughCfg.rewards):
        max_contact_force = 350.0

        class scales(AnymalCRoughCfg.rewards.scales):
            feet_air_time = 2.0
            orientation = -5.0
            torques = -0.000025

    class terrain(AnymalCRoughCfg.terrain):
        measure_heights = False
        mesh_type = "plane"


class AnymalCFlatCfgPPO(AnymalCRoughCfgPPO):
    class algorithm(AnymalCRoughCfgPPO.algorithm):
        entropy_coef = 0.01

    class policy(AnymalCRoughCfgPPO.policy):
        activation = "elu"
        actor_hidden_dims = [128, 64, 32]
        critic_hidden_dims = [128, 64, 32]

    class runner(AnymalCRoughCfgPPO.runner):
        experiment_name = "flat_anymal_c"
        load_run = -1
        max_iterations = 300
        run_name = ""


# --------------------------------------------------------------------------- #
#  ANYmal-B (anymal_b_config.py:33-45) and Unitree A1 (a1_config.py:33-86)
# --------------------------------------------------------------------------- #
class AnymalBRoughCfg(AnymalCRoughCfg):
    class asset(AnymalCRoughCfg.asset):
        file = _ROOT + "/resources/robots/anymal_b/urdf/anymal_b.urdf"
        foot_name = "FOOT"
        name = "anymal_b"

    class rewards(AnymalCRoughCfg.rewards):
        class scales(AnymalCRoughCfg.rewards.scales):
            pass


class AnymalBRoughCfgPPO(AnymalCRoughCfgPPO):
    class runner(AnymalCRoughCfgPPO.runner):
        experiment_name = "rough_anymal_b"
        load_run = -1
        run_name = ""


def _a1_default_angles():
    q = {}
    for leg in ("FL", "RL", "FR", "RR"):
        q[f"{leg}_hip_joint"] = 0.1 if leg[1] == "L" else -0.1
        q[f"{leg}_thigh_joint"] = 0.8 if leg[0] == "F" else 1.0
        q[f"{leg}_calf_joint"] = -1.5
    return q


class A1RoughCfg(LeggedRobotCfg):
    class asset(LeggedRobotCfg.asset):
        file = _ROOT + "/resources/robots/a1/urdf/a1.urdf"
        foot_name = "foot"
        name = "a1"
        penalize_contacts_on = ["thigh", "calf"]
        self_collisions = 1
        terminate_after_contacts_on = ["base"]

    class control(LeggedRobotCfg.control):
        action_scale = 0.25
        control_type = "P"
        damping = {"joint": 0.5}
        decimation = 4
        stiffness = {"joint": 20.0}

    class init_state(LeggedRobotCfg.init_state):
        default_joint_angles = _a1_default_angles()
        pos = [0.0, 0.0, 0.42]

    class rewards(LeggedRobotCfg.rewards):
        base_height_target = 0.25
        soft_dof_pos_limit = 0.9

        class scales(LeggedRobotCfg.rewards.scales):
            dof_pos_limits = -10.0
            torques = -0.0002


class A1RoughCfgPPO(LeggedRobotCfgPPO):
    class algorithm(LeggedRobotCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(LeggedRobotCfgPPO.runner):
        experiment_name = "rough_a1"
        run_name = ""


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
    class asset(LeggedRobotCfg.asset):
        file = _ROOT + "/resources/robots/cassie/urdf/cassie.urdf"
        flip_visual_attachments = False
        foot_name = "toe"
        name = "cassie"
        self_collisions = 1
        terminate_after_contacts_on = ["pelvis"]

    class control(LeggedRobotCfg.control):
        action_scale = 0.5
        damping = {"hip_abduction": 3.0, "hip_rotation": 3.0, "hip_flexion": 6.0,
                   "thigh_joint": 6.0, "ankle_joint": 6.0, "toe_joint": 1.0}
        decimation = 4
        stiffness = {"hip_abduction": 100.0, "hip_rotation": 100.0, "hip_flexion": 200.0,
                     "thigh_joint": 200.0, "ankle_joint": 200.0, "toe_joint": 40.0}

    class env(LeggedRobotCfg.env):
        num_actions = 12
        num_envs = 4096
        num_observations = 169

    class init_state(LeggedRobotCfg.init_state):
        default_joint_angles = _cassie_default_angles()
        pos = [0.0, 0.0, 1.0]

    class rewards(LeggedRobotCfg.rewards):
        max_contact_force = 300.0
        only_positive_rewards = False
        soft_dof_pos_limit = 0.95
        soft_dof_vel_limit = 0.9
        soft_torque_limit = 0.9

        class scales(LeggedRobotCfg.rewards.scales):
            ang_vel_xy = -0.0
            dof_acc = -2.0e-7
            dof_pos_limits = -1.0
            dof_vel = -0.0
            feet_air_time = 5.0
            feet_contact_forces = -0.0
            lin_vel_z = -0.5
            no_fly = 0.25
            termination = -200.0
            torques = -5.0e-6
            tracking_ang_vel = 1.0

    class terrain(LeggedRobotCfg.terrain):
        measured_points_x = _sym_grid(5)    # 11 x 11 = 121 points
        measured_points_y = _sym_grid(5)


class CassieRoughCfgPPO(LeggedRobotCfgPPO):
    class algorithm(LeggedRobotCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(LeggedRobotCfgPPO.runner):
        experiment_name = "rough_cassie"
        run_name = ""
