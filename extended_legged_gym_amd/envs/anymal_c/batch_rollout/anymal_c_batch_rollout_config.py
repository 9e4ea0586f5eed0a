"""ANYmal-C main-rollout task configs (values of the reference's `envs/anymal_c/batch_rollout/anymal_c_batch_rollout_config.py:
34-262` and `anymal_c_batch_rollout_flat_config.py:36-209`): plane, 48 observations, PD actuators, no contact
termination (an upside-down base ends the episode instead), `only_positive_rewards` off."""
from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout_percept_config import (
    RobotBatchRolloutPerceptCfg, RobotBatchRolloutPerceptCfgPPO)
from extended_legged_gym_amd.utils.gait_scheduler import AsyncGaitSchedulerCfg


class AnymalCBatchRolloutCfg(RobotBatchRolloutPerceptCfg):
    class gait_scheduler:            # time-driven gait shaping of the class (zero-scaled in the shipped configs)
        period = 1.0
        duty = 0.5
        foot_phases = [0.0, 0.5, 0.0, 0.5]
        dt = 0.02
        swing_height = 0.04
        track_sigma = 0.25

    class async_gait_scheduler(AsyncGaitSchedulerCfg):
        dof_names = ['LF_HAA', 'LF_HFE', 'LF_KFE', 'RF_HAA', 'RF_HFE', 'RF_KFE',
                     'LH_HAA', 'LH_HFE', 'LH_KFE', 'RH_HAA', 'RH_HFE', 'RH_KFE']
        dof_align_sets = [['LF_HFE', 'RH_HFE'], ['RF_HFE', 'LH_HFE'], ['LF_KFE', 'RH_KFE'], ['RF_KFE', 'LH_KFE']]
        dof_nominal_pos = [0.0, 0.4, -0.8, 0.0, 0.4, -0.8, 0.0, -0.4, 0.8, 0.0, -0.4, 0.8]
        foot_names = ['LF_FOOT', 'RF_FOOT', 'LH_FOOT', 'RH_FOOT']
        foot_z_align_sets = [['LF_FOOT', 'RH_FOOT'], ['RF_FOOT', 'LH_FOOT']]

    class env(RobotBatchRolloutPerceptCfg.env):
        num_envs = 32            # main envs
        rollout_envs = 1
        num_observations = 48
        num_actions = 12
        episode_length_s = 20

    class terrain(RobotBatchRolloutPerceptCfg.terrain):
        use_terrain_obj = False
        mesh_type = 'plane'
        measure_heights = False
        curriculum = False
        max_init_terrain_level = 2
        terrain_length = 6.
        terrain_width = 6.
        num_rows = 2
        num_cols = 1
        terrain_proportions = [0.1, 0.1, 0.35, 0.3, 0.2]
        confined_terrain_proportions = [0.0, 0.0, 1.0, 0.0]
        random_origins = False
        origin_generation_max_attempts = 10000
        origins_x_range = [-20.0, 20.0]
        origins_y_range = [-20.0, 20.0]
        height_clearance_factor = 2.0

    class raycaster(RobotBatchRolloutPerceptCfg.raycaster):
        enable_raycast = False
        ray_pattern = "spherical"
        num_rays = 10
        ray_angle = 30.0
        terrain_file = ""
        max_distance = 10.0
        attach_yaw_only = False
        offset_pos = [0.0, 0.0, 0.0]
        spherical_num_azimuth = 16
        spherical_num_elevation = 8

    class sdf(RobotBatchRolloutPerceptCfg.sdf):
        enable_sdf = False
        mesh_paths = []
        max_distance = 10.0
        update_freq = 5
        query_bodies = ["base", "LF_SHANK", "RF_SHANK", "LH_SHANK", "RH_SHANK"]
        compute_gradients = True
        compute_nearest_points = True
        include_in_obs = True

    class commands(RobotBatchRolloutPerceptCfg.commands):
        curriculum = False
        max_curriculum = 1.
        num_commands = 4
        resampling_time = 4.
        heading_command = False

        class ranges(RobotBatchRolloutPerceptCfg.commands.ranges):
            lin_vel_x = [-1.0, 1.0]
            lin_vel_y = [-1.0, 1.0]
            ang_vel_yaw = [-1.0, 1.0]
            heading = [-3.14, 3.14]

    class init_state(RobotBatchRolloutPerceptCfg.init_state):
        pos = [0.0, 0.0, 0.5]
        rot = [0.0, 0.0, 0.0, 1.0]
        default_joint_angles = {
            'LF_HAA': 0.0, 'LF_HFE': 0.4, 'LF_KFE': -1.1,
            'RF_HAA': 0.0, 'RF_HFE': 0.4, 'RF_KFE': -1.1,
            'LH_HAA': 0.0, 'LH_HFE': -0.4, 'LH_KFE': 1.1,
            'RH_HAA': 0.0, 'RH_HFE': -0.4, 'RH_KFE': 1.1,
        }

    class control(RobotBatchRolloutPerceptCfg.control):
        control_type = 'P'
        jointpos_action_normalization = False
        stiffness = {'HAA': 80., 'HFE': 80., 'KFE': 80.}
        damping = {'HAA': 2., 'HFE': 2., 'KFE': 2.}
        action_scale = 0.5
        decimation = 4
        use_actuator_network = False
        actuator_net_file = "{LEGGED_GYM_ROOT_DIR}/resources/actuator_nets/anydrive_v3_lstm.pt"

    class asset(RobotBatchRolloutPerceptCfg.asset):
        file = "{LEGGED_GYM_ROOT_DIR}/resources/robots/anymal_c/urdf/anymal_c.urdf"
        name = "anymal_c"
        foot_name = "FOOT"
        penalize_contacts_on = ["SHANK", "THIGH", "base"]
        terminate_after_contacts_on = []
        self_collisions = 1

    class rewards(RobotBatchRolloutPerceptCfg.rewards):
        max_contact_force = 500.
        base_height_target = 0.5
        only_positive_rewards = False
        multi_stage_rewards = False
        reward_stage_threshold = 6.0
        reward_min_stage = 0
        reward_max_stage = 1

        class scales:
            termination = -0.0
            tracking_lin_vel = 2.0
            tracking_ang_vel = 0.5
            lin_vel_z = -1.0
            ang_vel_xy = -0.5
            orientation = -2.0
            torques = -0.00001
            dof_vel = -0.
            dof_acc = -2.5e-7
            feet_air_time = 0.4
            collision = -0.6
            feet_stumble = -0.8
            feet_stumble_liftup = 1.0
            action_rate = -0.001
            stand_still = -0.

        class async_gait_scheduler:  # weights of the three alignment terms per reward stage
            dof_align = 1.0
            dof_nominal_pos = [0.05, 0.2]
            reward_foot_z_align = [0.1, 0.6]

    class domain_rand(RobotBatchRolloutPerceptCfg.domain_rand):
        randomize_base_mass = True
        added_mass_range = [-5., 5.]

    class viewer(RobotBatchRolloutPerceptCfg.viewer):
        ref_env = 0
        pos = [2.0, 0.0, 2.0]
        lookat = [0.5, 0.0, 0.]


class AnymalCBatchRolloutCfgPPO(RobotBatchRolloutPerceptCfgPPO):
    class policy(RobotBatchRolloutPerceptCfgPPO.policy):
        actor_hidden_dims = [128, 64, 32]
        critic_hidden_dims = [128, 64, 32]
        activation = 'elu'

    class algorithm(RobotBatchRolloutPerceptCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(RobotBatchRolloutPerceptCfgPPO.runner):
        run_name = ''
        experiment_name = 'anymal_c_batch_rollout'
        load_run = -1
        max_iterations = 3000


class AnymalCBatchRolloutFlatCfg(AnymalCBatchRolloutCfg):
    """`anymal_c_batch_rollout_flat` (flat_config.py:36-209): training on the plane, two-stage reward scales."""
    class env(AnymalCBatchRolloutCfg.env):
        num_envs = 32
        rollout_envs = 0
        num_observations = 48
        num_actions = 12
        episode_length_s = 20

    class init_state(AnymalCBatchRolloutCfg.init_state):
        pos = [0.0, 0.0, 0.5]
        default_joint_angles = {
            'LF_HAA': 0.0, 'LF_HFE': 0.4, 'LF_KFE': -0.8,
            'RF_HAA': 0.0, 'RF_HFE': 0.4, 'RF_KFE': -0.8,
            'LH_HAA': 0.0, 'LH_HFE': -0.4, 'LH_KFE': 0.8,
            'RH_HAA': 0.0, 'RH_HFE': -0.4, 'RH_KFE': 0.8,
        }

    class asset(AnymalCBatchRolloutCfg.asset):
        penalize_contacts_on = ["SHANK", "THIGH"]
        terminate_after_contacts_on = ["base"]

    class rewards(AnymalCBatchRolloutCfg.rewards):
        max_contact_force = 500.
        base_height_target = 0.5
        only_positive_rewards = True
        multi_stage_rewards = True
        reward_stage_threshold = 6.0
        reward_min_stage = 0
        reward_max_stage = 1

        class scales:
            termination = -0.0
            tracking_lin_vel = 1.0
            tracking_ang_vel = 0.5
            lin_vel_z = -2.0
            ang_vel_xy = -0.05
            orientation = -0.
            torques = -0.00001
            dof_vel = -0.
            dof_acc = -2.5e-7
            base_height = -0.
            feet_air_time = 1.0
            collision = -1.
            feet_stumble = -0.0
            action_rate = -0.01
            stand_still = -0.


class AnymalCBatchRolloutFlatCfgPPO(AnymalCBatchRolloutCfgPPO):
    class runner(AnymalCBatchRolloutCfgPPO.runner):
        experiment_name = 'anymal_c_batch_rollout'      # (the reference's flat task logs under the same name)
        multi_stage_rewards = True
