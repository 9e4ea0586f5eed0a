"""Task `foot_track_elspider_air_hang` (values of the reference's `envs/elspider_air/flat/foot_track_elspider_air_hang_config.py:33-135`): the hexapod
hung by its base (`asset.fix_base_link`) tracking the planner of type 0 (`rewards.raibert_planner` of `ElSpiderAirRoughCfg`).  The config asks for 88
observations while `FootTrackElSpider.compute_observations` builds 94 (`elspider.py:561-581`), so with observation noise on -- the default it inherits --
the reference raises a shape error on the first step; the task is registered so that `make_env` answers with that error (`FootTrackElSpider.__init__`)."""
from extended_legged_gym_amd.envs.elspider_air.mixed_terrains.elspider_air_rough_config import ElSpiderAirRoughCfg, ElSpiderAirRoughCfgPPO


class FootTrackElSpiderAirHangCfg(ElSpiderAirRoughCfg):
    class env(ElSpiderAirRoughCfg.env):
        num_observations = 88

    class terrain(ElSpiderAirRoughCfg.terrain):
        mesh_type = 'plane'
        measure_heights = False

    class asset(ElSpiderAirRoughCfg.asset):
        self_collisions = 0
        fix_base_link = True

    class init_state(ElSpiderAirRoughCfg.init_state):
        pos = [0.0, 0.0, 0.28]

    class rewards(ElSpiderAirRoughCfg.rewards):
        base_height_target = 0.28
        max_contact_force = 500.
        only_positive_rewards = True
        multi_stage_rewards = True
        reward_stage_threshold = 5.0
        reward_min_stage = 0
        reward_max_stage = 0

        class scales(ElSpiderAirRoughCfg.rewards.scales):
            termination = -0.0
            lin_vel_z = -2.0
            ang_vel_xy = -0.05
            orientation = [-0.0, 0.0]
            torques = -0.00001
            dof_vel = -0.0
            action_rate = -0.001
            dof_acc = -5e-8
            base_height = [-0.0, 0.0]
            feet_slip = [-0.0, -0.3]
            feet_air_time = 0.8
            feet_stumble = -0.0
            stand_still = -0.
            dof_pos_limits = -1.0
            collision = -1.0
            raibert_base_pos_track = [-0.0, -6.0]
            raibert_base_quat_track = [-0.0, -10.0]
            raibert_foot_pos_track = [-1.0, -3.0]

        class async_gait_scheduler:
            dof_align = 1.0
            dof_nominal_pos = [0.0, 0.2]
            reward_foot_z_align = [0.0, 0.6]

    class commands(ElSpiderAirRoughCfg.commands):
        curriculum = False
        max_curriculum = 2.5
        num_commands = 4
        resampling_time = 1.
        heading_command = True
        pose_command = False

        class ranges(ElSpiderAirRoughCfg.commands.ranges):
            lin_vel_x = [-0.4, 0.4]
            lin_vel_y = [-0.3, 0.3]
            ang_vel_yaw = [-0.3, 0.3]
            heading = [-0.5, 0.5]

    class domain_rand(ElSpiderAirRoughCfg.domain_rand):
        friction_range = [0.5, 1.5]


class FootTrackElSpiderAirHangCfgPPO(ElSpiderAirRoughCfgPPO):
    class policy(ElSpiderAirRoughCfgPPO.policy):
        actor_hidden_dims = [128, 64, 32]
        critic_hidden_dims = [128, 64, 32]
        activation = 'elu'

    class algorithm(ElSpiderAirRoughCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(ElSpiderAirRoughCfgPPO.runner):
        run_name = ''
        experiment_name = 'foot_track_elspider_air_hang'
        load_run = -1
        max_iterations = 3000
        multi_stage_rewards = True
