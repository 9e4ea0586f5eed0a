"""Task `elspider_air_rough` (values of the reference's `envs/elspider_air/mixed_terrains/elspider_air_rough_train_config.py:33-196`; it is
this config, not `ElSpiderAirRoughCfg`, that the registry binds to the task name, `envs/__init__.py:154`): the 253-entry observation with
the height scan, on a plane, 4 x 4 m tiles, noise off, termination penalised."""
from extended_legged_gym_amd.envs.base.legged_robot_config import LeggedRobotCfg, LeggedRobotCfgPPO
from extended_legged_gym_amd.envs.elspider_air.mixed_terrains.elspider_air_rough_config import ElSpiderAirRoughCfg


class ElSpiderAirRoughTrainCfg(LeggedRobotCfg):
    class env(LeggedRobotCfg.env):
        num_envs = 4096
        num_actions = 18
        num_observations = 253

    class terrain(ElSpiderAirRoughCfg.terrain):
        mesh_type = 'plane'
        terrain_length = 4.
        terrain_width = 4.
        num_rows = 4
        num_cols = 4

    class init_state(ElSpiderAirRoughCfg.init_state):
        pass

    class control(ElSpiderAirRoughCfg.control):
        pass

    class asset(ElSpiderAirRoughCfg.asset):
        pass

    class domain_rand(LeggedRobotCfg.domain_rand):
        randomize_base_mass = True
        added_mass_range = [-5., 5.]

    class rewards(LeggedRobotCfg.rewards):
        max_contact_force = 500.
        base_height_target = 0.28
        only_positive_rewards = True
        multi_stage_rewards = True
        reward_stage_threshold = 6.0
        reward_min_stage = 0
        reward_max_stage = 1

        class scales:
            termination = -5.0
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
            dof_align = 0.5
            dof_nominal_pos = [0.1, 0.2]
            reward_foot_z_align = [0.2, 0.05]

        class raibert_planner:
            planner_type = 0
            base_pos_track = 1.0
            base_quat_track = 0.5
            foot_pos_track = 0.3

    class noise:
        add_noise = False
        noise_level = 1.0

        class noise_scales:
            dof_pos = 0.01
            dof_vel = 1.5
            lin_vel = 0.1
            ang_vel = 0.2
            gravity = 0.05
            height_measurements = 0.1


class ElSpiderAirRoughTrainCfgPPO(LeggedRobotCfgPPO):
    class policy(LeggedRobotCfgPPO.policy):
        actor_hidden_dims = [128, 64, 32]
        critic_hidden_dims = [128, 64, 32]
        activation = 'elu'

    class algorithm(LeggedRobotCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(LeggedRobotCfgPPO.runner):
        run_name = ''
        experiment_name = 'rough_elspider_air'
        load_run = -1
        max_iterations = 3000
        multi_stage_rewards = True
