"""ElSpider Air flat-ground config, task `elspider_air_flat` (values of the reference's
`envs/elspider_air/flat/elspider_air_flat_config.py:34-118`): 66 observations, plane, the tripod-gait reward set with two stages."""
from extended_legged_gym_amd.envs.elspider_air.mixed_terrains.elspider_air_rough_config import ElSpiderAirRoughCfg, ElSpiderAirRoughCfgPPO


class ElSpiderAirFlatCfg(ElSpiderAirRoughCfg):
    class env(ElSpiderAirRoughCfg.env):
        num_observations = 66

    class terrain(ElSpiderAirRoughCfg.terrain):
        mesh_type = 'plane'
        measure_heights = False

    class asset(ElSpiderAirRoughCfg.asset):
        self_collisions = 0

    class rewards(ElSpiderAirRoughCfg.rewards):
        max_contact_force = 500.
        base_height_target = 0.28
        only_positive_rewards = True
        multi_stage_rewards = True
        reward_stage_threshold = 6.0
        reward_min_stage = 0
        reward_max_stage = 1

        class scales:        # (not derived from the rough config's scales in the reference either)
            termination = -0.0
            tracking_lin_vel = 1.0
            tracking_ang_vel = 0.5
            lin_vel_z = -2.0
            ang_vel_xy = -0.05
            orientation = -5.0
            torques = -0.00001
            dof_vel = -0.
            dof_acc = -5e-8
            base_height = -8.0
            feet_slip = [-0.0, -0.4]
            feet_air_time = 0.8
            collision = -1.
            feet_stumble = -0.0
            action_rate = -0.001
            stand_still = -0.
            dof_pos_limits = -1.0
            gait_2_step = -5.0

        class async_gait_scheduler:
            dof_align = 1.0
            dof_nominal_pos = [0.0, 0.2]
            reward_foot_z_align = [0.0, 0.6]

    class commands(ElSpiderAirRoughCfg.commands):
        curriculum = False
        max_curriculum = 1.
        num_commands = 4
        resampling_time = 4.
        heading_command = False

        class ranges(ElSpiderAirRoughCfg.commands.ranges):
            lin_vel_x = [-1.5, 1.5]
            lin_vel_y = [-0.6, 0.6]
            ang_vel_yaw = [-0.6, 0.6]
            heading = [-3.14, 3.14]

    class domain_rand(ElSpiderAirRoughCfg.domain_rand):
        friction_range = [0.5, 1.5]


class ElSpiderAirFlatCfgPPO(ElSpiderAirRoughCfgPPO):
    class policy(ElSpiderAirRoughCfgPPO.policy):
        actor_hidden_dims = [128, 64, 32]
        critic_hidden_dims = [128, 64, 32]
        activation = 'elu'

    class algorithm(ElSpiderAirRoughCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(ElSpiderAirRoughCfgPPO.runner):
        run_name = ''
        experiment_name = 'flat_elspider_air'
        load_run = -1
        max_iterations = 3000
        multi_stage_rewards = True
