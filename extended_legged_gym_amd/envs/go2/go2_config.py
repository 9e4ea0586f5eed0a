"""Unitree Go2 task configs (values of the reference's `envs/go2/flat/go2_rough_config.py:33-174` and
`go2_flat_config.py:33-97`)."""
from extended_legged_gym_amd.envs.base.legged_robot_config import LeggedRobotCfg, LeggedRobotCfgPPO


class Go2RoughCfg(LeggedRobotCfg):
    class env(LeggedRobotCfg.env):
        num_envs = 4096
        num_observations = 235
        num_actions = 12

    class terrain(LeggedRobotCfg.terrain):
        curriculum = True
        mesh_type = 'trimesh'
        measure_heights = True

    class commands(LeggedRobotCfg.commands):
        curriculum = False
        max_curriculum = 1.
        resampling_time = 10.
        heading_command = True

        class ranges:
            lin_vel_x = [-1.0, 1.0]
            lin_vel_y = [-1.0, 1.0]
            ang_vel_yaw = [-1, 1]
            heading = [-3.14, 3.14]

    class init_state(LeggedRobotCfg.init_state):
        pos = [0.0, 0.0, 0.33]
        default_joint_angles = {
            'FL_hip_joint': 0.1, 'RL_hip_joint': 0.1, 'FR_hip_joint': -0.1, 'RR_hip_joint': -0.1,
            'FL_thigh_joint': 0.8, 'RL_thigh_joint': 0.8, 'FR_thigh_joint': 0.8, 'RR_thigh_joint': 0.8,
            'FL_calf_joint': -1.5, 'RL_calf_joint': -1.5, 'FR_calf_joint': -1.5, 'RR_calf_joint': -1.5,
        }

    class control(LeggedRobotCfg.control):
        stiffness = {'joint': 30.0}
        damping = {'joint': 0.8}
        action_scale = 0.3
        decimation = 4
        use_actuator_network = False

    class asset(LeggedRobotCfg.asset):
        file = "{LEGGED_GYM_ROOT_DIR}/resources/robots/go2/urdf/go2_description.urdf"
        name = "go2"
        foot_name = "foot"
        penalize_contacts_on = ["thigh", "calf"]
        terminate_after_contacts_on = ["base", "Head_upper"]
        self_collisions = 1

    class rewards(LeggedRobotCfg.rewards):
        soft_dof_pos_limit = 0.9
        max_contact_force = 350.0
        base_height_target = 0.25

        class scales(LeggedRobotCfg.rewards.scales):
            orientation = -0.5
            torques = -0.00001
            action_rate = -0.001

    class domain_rand(LeggedRobotCfg.domain_rand):
        randomize_base_mass = True
        added_mass_range = [-1., 1.]


class Go2RoughCfgPPO(LeggedRobotCfgPPO):
    class algorithm(LeggedRobotCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(LeggedRobotCfgPPO.runner):
        run_name = ''
        experiment_name = 'rough_go2'
        max_iterations = 1000


class Go2FlatCfg(Go2RoughCfg):
    class env(Go2RoughCfg.env):
        num_observations = 48

    class terrain(Go2RoughCfg.terrain):
        mesh_type = 'plane'
        measure_heights = False

    class asset(Go2RoughCfg.asset):
        self_collisions = 0

    class rewards(Go2RoughCfg.rewards):
        max_contact_force = 350.

        class scales(Go2RoughCfg.rewards.scales):
            orientation = -5.0
            torques = -0.000025
            action_rate = -0.01
            feet_air_time = 1.0

    class commands(Go2RoughCfg.commands):
        heading_command = False
        resampling_time = 4.

        class ranges(Go2RoughCfg.commands.ranges):
            ang_vel_yaw = [-1.5, 1.5]

    class domain_rand(Go2RoughCfg.domain_rand):
        friction_range = [0.5, 1.5]


class Go2FlatCfgPPO(Go2RoughCfgPPO):
    class policy(Go2RoughCfgPPO.policy):
        actor_hidden_dims = [128, 64, 32]
        critic_hidden_dims = [128, 64, 32]
        activation = 'elu'

    class runner(Go2RoughCfgPPO.runner):
        experiment_name = 'flat_go2'
        max_iterations = 300


class LoadAdaptGo2FlatCfg(Go2FlatCfg):
    """`load_adapt_go2_flat` (values of the reference's `envs/go2/flat/load_adapt_go2_flat_config.py:4-60`)."""
    class env(Go2FlatCfg.env):
        num_observations = 48
        num_actions = 12

    class commands(Go2FlatCfg.commands):
        resampling_time = 10.

        class ranges:
            lin_vel_x = [-0.5, 0.5]
            lin_vel_y = [-0.5, 0.5]
            ang_vel_yaw = [-0.5, 0.5]
            heading = [-3.14, 3.14]

    class rewards(Go2FlatCfg.rewards):
        class scales(Go2FlatCfg.rewards.scales):
            termination = -0.0
            tracking_lin_vel = 1.0
            tracking_ang_vel = 0.5
            lin_vel_z = -2.0
            ang_vel_xy = -0.05
            orientation = -5.0
            torques = -0.00001
            dof_vel = -0.
            dof_acc = -2.5e-7
            base_height = -0.
            collision = -1.
            action_rate = -0.01
            stumble = -0.0
            stand_still = -0.

    class domain_rand(Go2FlatCfg.domain_rand):
        randomize_friction = True
        friction_range = [0.2, 1.25]
        randomize_base_mass = True
        added_mass_range = [-3., 3.]
        push_robots = True
        push_interval_s = 3
        max_push_vel_xy = 1.


class LoadAdaptGo2FlatCfgPPO(Go2FlatCfgPPO):
    class algorithm(Go2FlatCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(Go2FlatCfgPPO.runner):
        run_name = ''
        experiment_name = 'load_adapt_go2'
        max_iterations = 300



class StandGo2FlatCfg(Go2FlatCfg):
    """`stand_go2_flat` (values of the reference's `envs/go2/flat/stand_go2_flat_config.py:33-86`).  The reference scales
    `standing = -10`, a term whose function raises on its first evaluation (`go2.py:268-270`: `torch.sum(..., dim=1)` of a 1-D
    tensor), so the registered task cannot step there; the value is kept and `NativeSetup` raises the same `IndexError` until
    the scale is set to 0."""
    class env(Go2FlatCfg.env):
        num_observations = 48
        num_actions = 12
        episode_length_s = 5.

    class commands(Go2FlatCfg.commands):
        num_commands = 4
        resampling_time = 5.

        class ranges:
            lin_vel_x = [-0.1, 0.1]
            lin_vel_y = [-0.1, 0.1]
            ang_vel_yaw = [-0.2, 0.2]
            heading = [-0.0, 0.0]

    class rewards(Go2FlatCfg.rewards):
        class scales(Go2FlatCfg.rewards.scales):
            tracking_lin_vel = 1.0
            tracking_ang_vel = 1.0
            lin_vel_z = -2.0
            ang_vel_xy = -0.5
            orientation = -10.0
            torques = -0.000025
            action_rate = -0.01
            standing = -10.0
            penalty_in_the_air = -5.0

    class init_state(Go2FlatCfg.init_state):
        pos = [0.0, 0.0, 0.43]
        default_joint_angles = {
            'FL_hip_joint': 0.0, 'RL_hip_joint': 0.0, 'FR_hip_joint': 0.0, 'RR_hip_joint': 0.0,
            'FL_thigh_joint': 0.9, 'RL_thigh_joint': 0.9, 'FR_thigh_joint': 0.9, 'RR_thigh_joint': 0.9,
            'FL_calf_joint': -1.8, 'RL_calf_joint': -1.8, 'FR_calf_joint': -1.8, 'RR_calf_joint': -1.8,
        }


class StandGo2FlatCfgPPO(Go2FlatCfgPPO):
    class runner(Go2FlatCfgPPO.runner):
        run_name = ''
        experiment_name = 'stand_go2'
        max_iterations = 300


class PoseGo2FlatCfg(Go2FlatCfg):
    """`pose_go2_flat` (values of the reference's `envs/go2/flat/pose_go2_flat_config.py:33-70`, including its `num_observations
    = 60` for a row of 52 entries: see `PoseGo2`)."""
    class env(Go2FlatCfg.env):
        num_observations = 60
        num_actions = 12

    class commands(Go2FlatCfg.commands):
        num_commands = 8
        resampling_time = 5.

        class ranges:
            lin_vel_x = [-0.5, 0.5]
            lin_vel_y = [-0.5, 0.5]
            ang_vel_yaw = [-0.5, 0.5]
            heading = [-0.0, 0.0]
            base_yaw_shift = [-0.0, 0.0]
            base_pitch_shift = [-0.3, 0.3]
            base_roll_shift = [-0.3, 0.3]
            base_height = [0.25, 0.42]

    class rewards(Go2FlatCfg.rewards):
        class scales(Go2FlatCfg.rewards.scales):
            orientation = -5.0
            base_height = -5.0

    class init_state(Go2FlatCfg.init_state):
        reset_mode = 'reset_to_range'
        pos = [0.0, 0.0, 0.52]


class PoseGo2FlatCfgPPO(Go2FlatCfgPPO):
    class runner(Go2FlatCfgPPO.runner):
        run_name = ''
        experiment_name = 'pose_go2'
        load_run = -1
        max_iterations = 300
