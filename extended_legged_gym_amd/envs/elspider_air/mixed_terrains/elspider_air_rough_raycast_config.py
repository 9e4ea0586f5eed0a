"""Task `elspider_air_rough_raycast` (values of the reference's `envs/elspider_air/mixed_terrains/elspider_air_rough_raycast_config.py:33-167`):
the hexapod on the confined two-layer terrain with 512 spherical rays in the observation (66 + 512) and a ray-cast depth camera, three
reward stages starting at the last one (`reward_min_stage = 2`)."""
from extended_legged_gym_amd.envs.base.legged_robot_config import LeggedRobotCfg
from extended_legged_gym_amd.envs.elspider_air.mixed_terrains.elspider_air_rough_train_config import ElSpiderAirRoughTrainCfg, ElSpiderAirRoughTrainCfgPPO


class ElSpiderAirRoughRaycastCfg(ElSpiderAirRoughTrainCfg):
    class env(ElSpiderAirRoughTrainCfg.env):
        num_observations = 66 + 512

    class terrain(ElSpiderAirRoughTrainCfg.terrain):
        use_terrain_obj = False
        terrain_file = None
        mesh_type = 'confined_trimesh'
        horizontal_scale = 0.1
        vertical_scale = 0.005
        border_size = 10
        curriculum = False
        static_friction = 1.0
        dynamic_friction = 1.0
        restitution = 0.
        measure_heights = False
        measured_points_x = [-0.8, -0.7, -0.6, -0.5, -0.4, -0.3, -0.2, -0.1, 0., 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
        measured_points_y = [-0.5, -0.4, -0.3, -0.2, -0.1, 0., 0.1, 0.2, 0.3, 0.4, 0.5]
        selected = False
        terrain_kwargs = None
        max_init_terrain_level = 0
        terrain_length = 6.
        terrain_width = 6.
        num_rows = 4
        num_cols = 6
        difficulty_scale = 1.0
        terrain_proportions = [0.1, 0.1, 0.3, 0.3, 0.2]
        confined_terrain_proportions = [0.0, 0.2, 0.3, 0.3]
        slope_treshold = 0.75

    class raycaster:
        enable_raycast = True
        terrain_file = None
        ray_pattern = "spherical2"
        spherical_num_azimuth = 12
        spherical_num_elevation = 8
        spherical2_num_points = 512
        spherical2_polar_axis = [0.0, 0.0, 1.0]
        ray_angle = 30
        num_rays = 96
        max_distance = 10.0
        attach_yaw_only = False
        offset_pos = [0.0, 0.0, 0.0]

    class depth(LeggedRobotCfg.depth):
        camera_type = "Warp"
        position = [0.45, 0, 0.03]
        angle = [30, 30]
        update_interval = 1
        original = (60, 30)
        resized = (56, 28)
        horizontal_fov = 100
        buffer_len = 2
        near_clip = 0
        far_clip = 10
        dis_noise = 0.0
        scale = 1
        invert = True

    class init_state(ElSpiderAirRoughTrainCfg.init_state):
        pos = [0.0, 0.0, 0.45]

    class commands(ElSpiderAirRoughTrainCfg.commands):
        curriculum = False
        max_curriculum = 1.
        num_commands = 4
        resampling_time = 10.
        heading_command = False

        class ranges:
            lin_vel_x = [-1.0, 1.0]
            lin_vel_y = [-0.5, 0.5]
            ang_vel_yaw = [-0.6, 0.6]
            heading = [-3.14, 3.14]

    class rewards(ElSpiderAirRoughTrainCfg.rewards):
        base_height_target = 0.35
        max_contact_force = 500.
        only_positive_rewards = True
        multi_stage_rewards = True
        reward_stage_threshold = 5.0
        reward_min_stage = 2
        reward_max_stage = 2

        class scales():
            tracking_lin_vel = 1.0
            tracking_ang_vel = 0.5
            lin_vel_z = -2.0
            ang_vel_xy = -0.05
            orientation = [-5.0, -5.0, 0.0]
            base_height = [-8.0, -8.0, 0.0]
            torques = -0.00001
            dof_vel = -0.
            dof_acc = [-5e-8, -5e-8, -5e-8]
            dof_pos_limits = -1.0
            action_rate = [-0.001, -0.001, -0.002]
            feet_slip = [-0.0, -0.4]
            feet_air_time = [0.8, 1.5]
            feet_stumble = [-1.0, -1.0, -2.0]
            feet_stumble_liftup = [1.0, 1.0, 2.0]
            feet_contact_forces = [0, 0, -0.05]
            termination = -1.0
            collision = -1.
            stand_still = -0.
            async_gait_scheduler = [-0.2, -0.2, -0.1]
            gait_2_step = [-5.0, -5.0, -2.0]

        class async_gait_scheduler:
            dof_align = 0.3
            dof_nominal_pos = 0.2
            reward_foot_z_align = 0.0


class ElSpiderAirRoughRaycastCfgPPO(ElSpiderAirRoughTrainCfgPPO):
    class runner(ElSpiderAirRoughTrainCfgPPO.runner):
        run_name = 'raycast512'
        experiment_name = 'rough_elspider_air'
        load_run = -1
        max_iterations = 5000
        multi_stage_rewards = True
