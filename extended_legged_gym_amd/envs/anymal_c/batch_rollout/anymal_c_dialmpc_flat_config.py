"""Task `anymal_c_dialmpc_flat` (reference `envs/anymal_c/batch_rollout/anymal_c_dialmpc_flat_config.py:5-221`, registered at
`envs/__init__.py:127`): the DIAL-MPC planner's ANYmal-C config -- plane, one rollout env per main env, PD actuators at action scale 1,
`tracking_sigma` 4, two-stage reward scales that include `async_gait_scheduler = [-0.2, -0.4]`.

As shipped the task cannot step in the reference: that term multiplies the 12 joint errors by the 18-entry (hexapod) weight vector the
section inherits (`utils/gait_scheduler.py:158-166`) and raises.  The same error is raised here when the env is built; give
`async_gait_scheduler.dof_nominal_pos_weight` twelve entries to run it (tests/test_async_gait.py does)."""
from .anymal_c_batch_rollout_config import AnymalCBatchRolloutCfg, AnymalCBatchRolloutCfgPPO


class AnymalCDialMPCFlatCfg(AnymalCBatchRolloutCfg):
    class env(AnymalCBatchRolloutCfg.env):
        num_envs = 32
        rollout_envs = 1
        num_observations = 48
        num_actions = 12
        episode_length_s = 20
        env_spacing = 2.0

    class terrain(AnymalCBatchRolloutCfg.terrain):
        use_terrain_obj = False
        mesh_type = 'plane'
        measure_heights = False
        curriculum = False
        random_origins = False
        origin_generation_max_attempts = 10000
        origins_x_range = [-20.0, 20.0]
        origins_y_range = [-20.0, 20.0]
        height_clearance_factor = 2.0

    class raycaster:
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

    class sdf:
        enable_sdf = False
        mesh_paths = []
        max_distance = 10.0
        enable_caching = True
        update_freq = 5
        query_bodies = ["base", "LF_SHANK", "RF_SHANK", "LH_SHANK", "RH_SHANK"]
        compute_gradients = True
        compute_nearest_points = True
        include_in_obs = True

    class init_state(AnymalCBatchRolloutCfg.init_state):
        pos = [0.0, 0.0, 0.5]
        rot = [0.0, 0.0, 0.0, 1.0]
        default_joint_angles = {
            'LF_HAA': 0.0, 'LF_HFE': 0.4, 'LF_KFE': -0.8,
            'RF_HAA': 0.0, 'RF_HFE': 0.4, 'RF_KFE': -0.8,
            'LH_HAA': 0.0, 'LH_HFE': -0.4, 'LH_KFE': 0.8,
            'RH_HAA': 0.0, 'RH_HFE': -0.4, 'RH_KFE': 0.8,
        }

    class control(AnymalCBatchRolloutCfg.control):
        stiffness = {'HAA': 80., 'HFE': 80., 'KFE': 80.}
        damping = {'HAA': 2., 'HFE': 2., 'KFE': 2.}
        action_scale = 1.0
        decimation = 4
        use_actuator_network = False

    class asset(AnymalCBatchRolloutCfg.asset):
        penalize_contacts_on = ["SHANK", "THIGH"]
        terminate_after_contacts_on = ["base"]
        self_collisions = 1

    class rewards(AnymalCBatchRolloutCfg.rewards):
        max_contact_force = 500.
        base_height_target = 0.5
        only_positive_rewards = False
        multi_stage_rewards = True
        reward_stage_threshold = 6.0
        reward_min_stage = 0
        reward_max_stage = 1
        tracking_sigma = 4.0

        class scales(AnymalCBatchRolloutCfg.rewards.scales):
            termination = -0.0
            tracking_lin_vel = 2.0
            tracking_ang_vel = 0.5
            lin_vel_z = -2.0
            ang_vel_xy = -0.05
            dof_vel = -0.
            feet_stumble = -0.0
            stand_still = -0.
            dof_pos_limits = -1.0
            orientation = -5.0
            torques = -0.00001
            action_rate = -0.001
            dof_acc = -0.5e-8
            feet_slip = [-0.0, -0.4]
            feet_air_time = 0.8
            collision = -1.0
            base_height = -8.0
            async_gait_scheduler = [-0.2, -0.4]

        class async_gait_scheduler:
            dof_align = 1.0
            dof_nominal_pos = [0.05, 0.2]
            reward_foot_z_align = [0.1, 0.6]

    class viewer(AnymalCBatchRolloutCfg.viewer):
        ref_env = 0
        pos = [-14.0, -14.0, 2.0]
        lookat = [-16.0, -16.0, 0.0]


class AnymalCDialMPCFlatCfgPPO(AnymalCBatchRolloutCfgPPO):
    class runner(AnymalCBatchRolloutCfgPPO.runner):
        run_name = ''
        experiment_name = 'anymal_c_batch_rollout'
        load_run = -1
        max_iterations = 3000
        multi_stage_rewards = True
