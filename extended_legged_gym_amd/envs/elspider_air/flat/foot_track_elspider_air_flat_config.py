"""Task `foot_track_elspider_air_flat` (values of the reference's `envs/elspider_air/flat/foot_track_elspider_air_flat_config.py:36-170`): the hexapod on a
plane following the Raibert planner of type 1 (random-walking pose shifts and footholds); 94 observations; the planner terms staged, the task pinned at
reward stage 2 (`reward_min_stage = reward_max_stage = 2`), where `async_gait_scheduler` and the velocity-tracking terms have scale 0."""
from extended_legged_gym_amd.envs.elspider_air.mixed_terrains.elspider_air_rough_config import ElSpiderAirRoughCfg, ElSpiderAirRoughCfgPPO


class FootTrackElSpiderAirFlatCfg(ElSpiderAirRoughCfg):
    class env(ElSpiderAirRoughCfg.env):
        num_observations = 94

    class terrain(ElSpiderAirRoughCfg.terrain):
        mesh_type = 'plane'
        measure_heights = False

    class asset(ElSpiderAirRoughCfg.asset):
        self_collisions = 0

    class rewards(ElSpiderAirRoughCfg.rewards):
        base_height_target = 0.30
        max_contact_force = 500.
        only_positive_rewards = True
        multi_stage_rewards = True
        reward_stage_threshold = 10.0
        reward_min_stage = 2
        reward_max_stage = 2

        class scales(ElSpiderAirRoughCfg.rewards.scales):
            termination = -0.0
            lin_vel_z = -2.0
            ang_vel_xy = -0.05
            orientation = 0.0
            torques = -0.00001
            dof_vel = -0.0
            action_rate = -0.001
            dof_acc = -5e-8
            base_height = 0.0
            feet_slip = [-0.2, -0.3]
            feet_air_time = [0.8, 0.8]
            feet_stumble = -0.0
            stand_still = -0.
            dof_pos_limits = -1.0
            collision = -1.0
            raibert_base_pos_track = [-1.0, -3.0]
            raibert_base_quat_track = [-1.0, -3.0, -6.0]
            raibert_foot_pos_track = [0.3, 0.5]
            raibert_foot_swing_contact = [-0.2, -0.3]
            raibert_foot_pos_track_z = [-0.4, -0.4]
            async_gait_scheduler = [-0.3, -0.0]
            tracking_lin_vel = [0.0, 0.0]
            tracking_ang_vel = [0.0, 0.0]

        class async_gait_scheduler:
            dof_align = 1.0
            dof_nominal_pos = [0.0, 0.2]
            reward_foot_z_align = [0.0, 0.6]

        class raibert_planner:
            planner_type = 1                 # 0: SimpleRaibertPlanner, 1: RaibertPlanner

    class commands(ElSpiderAirRoughCfg.commands):
        curriculum = False
        max_curriculum = 2.5
        num_commands = 4
        resampling_time = 4.
        heading_command = False
        pose_command = True

        class ranges(ElSpiderAirRoughCfg.commands.ranges):
            lin_vel_x = [-1.0, 1.0]
            lin_vel_y = [-0.4, 0.4]
            ang_vel_yaw = [-0.4, 0.4]
            heading = [-3.14, 3.14]

    class domain_rand(ElSpiderAirRoughCfg.domain_rand):
        friction_range = [0.5, 1.5]


class FootTrackElSpiderAirFlatCfgPPO(ElSpiderAirRoughCfgPPO):
    class policy(ElSpiderAirRoughCfgPPO.policy):
        actor_hidden_dims = [128, 64, 32]
        critic_hidden_dims = [128, 64, 32]
        activation = 'elu'

    class algorithm(ElSpiderAirRoughCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(ElSpiderAirRoughCfgPPO.runner):
        run_name = ''
        experiment_name = 'foot_track_elspider_air_flat'
        load_run = -1
        max_iterations = 3000
        multi_stage_rewards = True
