"""Go2 main-rollout task configs (values of the reference's `envs/go2/batch_rollout/go2_batch_rollout_config.py:11-240` and
`go2_batch_rollout_flat_config.py:5-200`; tasks `go2_batch_rollout`, `go2_batch_rollout_flat`, `envs/__init__.py:142-143`)."""
from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout_percept_config import (
    RobotBatchRolloutPerceptCfg, RobotBatchRolloutPerceptCfgPPO)
from extended_legged_gym_amd.utils.gait_scheduler import AsyncGaitSchedulerCfg


class Go2BatchRolloutCfg(RobotBatchRolloutPerceptCfg):
    class gait_scheduler:
        period = 0.6
        duty = 0.5
        foot_phases = [0.0, 0.5, 0.5, 0.0]
        dt = 0.005
        swing_height = 0.15
        track_sigma = 0.25

    class async_gait_scheduler(AsyncGaitSchedulerCfg):
        dof_names = ['FL_hip_joint', 'FL_thigh_joint', 'FL_calf_joint', 'FR_hip_joint', 'FR_thigh_joint', 'FR_calf_joint',
                     'RL_hip_joint', 'RL_thigh_joint', 'RL_calf_joint', 'RR_hip_joint', 'RR_thigh_joint', 'RR_calf_joint']
        dof_align_sets = [['FL_thigh_joint', 'RR_thigh_joint'], ['FR_thigh_joint', 'RL_thigh_joint'],
                          ['FL_calf_joint', 'RR_calf_joint'], ['FR_calf_joint', 'RL_calf_joint']]
        dof_nominal_pos = [0.1, 0.8, -1.5, -0.1, 0.8, -1.5, 0.1, 1.0, -1.5, -0.1, 1.0, -1.5]
        foot_names = ['FL_foot', 'FR_foot', 'RL_foot', 'RR_foot']
        foot_z_align_sets = [['FL_foot', 'RR_foot'], ['FR_foot', 'RL_foot']]

    class env(RobotBatchRolloutPerceptCfg.env):
        num_envs = 32            # main envs
        rollout_envs = 1
        num_observations = 181   # 48 + 16 x 8 rays + 5 body SDF values
        num_actions = 12
        episode_length_s = 20

    class terrain(RobotBatchRolloutPerceptCfg.terrain):
        use_terrain_obj = True
        terrain_file = ""        # the reference names a file of its author's machine; point it at an OBJ mesh
        measure_heights = False
        curriculum = False
        random_origins = True
        origin_generation_max_attempts = 10000
        origins_x_range = [-20.0, 20.0]
        origins_y_range = [-20.0, 20.0]
        height_clearance_factor = 2.0

    class raycaster(RobotBatchRolloutPerceptCfg.raycaster):
        enable_raycast = True
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
        enable_sdf = True
        mesh_paths = []
        max_distance = 10.0
        enable_caching = True
        update_freq = 5
        query_bodies = ["base", "FL_calf", "FR_calf", "RL_calf", "RR_calf"]
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
        pos = [0.0, 0.0, 0.43]
        rot = [0.0, 0.0, 0.0, 1.0]
        default_joint_angles = {
            'FL_hip_joint': 0.1, 'FL_thigh_joint': 0.8, 'FL_calf_joint': -1.5,
            'FR_hip_joint': -0.1, 'FR_thigh_joint': 0.8, 'FR_calf_joint': -1.5,
            'RL_hip_joint': 0.1, 'RL_thigh_joint': 1.0, 'RL_calf_joint': -1.5,
            'RR_hip_joint': -0.1, 'RR_thigh_joint': 1.0, 'RR_calf_joint': -1.5,
        }

    class control(RobotBatchRolloutPerceptCfg.control):
        stiffness = {'joint': 55.0}
        damping = {'joint': 0.8}
        action_scale = 0.5
        decimation = 4
        use_actuator_network = False
        actuator_net_file = "{LEGGED_GYM_ROOT_DIR}/resources/actuator_nets/go2_actuator_net.pt"

    class asset(RobotBatchRolloutPerceptCfg.asset):
        file = "{LEGGED_GYM_ROOT_DIR}/resources/robots/go2/urdf/go2_description.urdf"
        name = "go2"
        foot_name = "foot"
        penalize_contacts_on = ["thigh", "calf"]
        terminate_after_contacts_on = ["base"]
        self_collisions = 1

    class rewards(RobotBatchRolloutPerceptCfg.rewards):
        max_contact_force = 350.
        base_height_target = 0.43
        only_positive_rewards = True
        multi_stage_rewards = True
        reward_stage_threshold = 6.0
        reward_min_stage = 0
        reward_max_stage = 1

        class scales(RobotBatchRolloutPerceptCfg.rewards.scales):
            termination = -0.0
            tracking_lin_vel = 1.0
            tracking_ang_vel = 0.5
            lin_vel_z = -2.0
            ang_vel_xy = -0.05
            orientation = -0.0
            torques = -0.00001
            dof_vel = -0.
            dof_acc = -2.5e-7
            base_height = -0.
            feet_slip = [-0.0, -0.4]
            feet_air_time = 1.0
            collision = -1.
            feet_stumble = -0.0
            action_rate = -0.003
            stand_still = -0.
            dof_pos_limits = -1.0

        class async_gait_scheduler:
            dof_align = 1.0
            dof_nominal_pos = [0.05, 0.2]
            reward_foot_z_align = [0.1, 0.6]

    class domain_rand(RobotBatchRolloutPerceptCfg.domain_rand):
        randomize_base_mass = True
        added_mass_range = [-1., 1.]

    class viewer(RobotBatchRolloutPerceptCfg.viewer):
        ref_env = 0
        pos = [2.0, 0.0, 2.0]
        lookat = [0.5, 0.0, 0.]


class Go2BatchRolloutCfgPPO(RobotBatchRolloutPerceptCfgPPO):
    class policy(RobotBatchRolloutPerceptCfgPPO.policy):
        actor_hidden_dims = [128, 64, 32]
        critic_hidden_dims = [128, 64, 32]
        activation = 'elu'

    class algorithm(RobotBatchRolloutPerceptCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(RobotBatchRolloutPerceptCfgPPO.runner):
        run_name = ''
        experiment_name = 'go2_batch_rollout'
        load_run = -1
        max_iterations = 3000
        multi_stage_rewards = True


class Go2BatchRolloutFlatCfg(Go2BatchRolloutCfg):
    class env(Go2BatchRolloutCfg.env):
        num_envs = 32
        rollout_envs = 0
        num_observations = 48
        num_actions = 12
        episode_length_s = 20

    class terrain(Go2BatchRolloutCfg.terrain):
        use_terrain_obj = False
        mesh_type = 'plane'
        random_origins = False

    class raycaster(Go2BatchRolloutCfg.raycaster):
        enable_raycast = False

    class sdf(Go2BatchRolloutCfg.sdf):
        enable_sdf = False

    class rewards(Go2BatchRolloutCfg.rewards):
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


class Go2BatchRolloutFlatCfgPPO(Go2BatchRolloutCfgPPO):
    class runner(Go2BatchRolloutCfgPPO.runner):
        experiment_name = 'go2_batch_rollout_flat'
