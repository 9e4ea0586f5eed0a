"""Task `foot_track_elspider_air_hang` (values of the reference's `envs/elspider_air/flat/foot_track_elspider_air_hang_config.py:33-135`): the hexapod hung by
its base (`asset.fix_base_link`) tracking the planner of type 0 (`rewards.raibert_planner` of `ElSpiderAirRoughCfg`).  The config asks for 88 observations while
`FootTrackElSpider.compute_observations` builds 94 (`elspider.py:561-581`), so with observation noise on -- the default it inherits -- the reference raises a
shape error on the first step; the task is registered so that `make_env` answers with that error (`FootTrackElSpider.__init__`)."""
from extended_legged_gym_amd.envs.elspider_air.mixed_terrains.elspider_air_rough_config import ElSpiderAirRoughCfg as _Rough, ElSpiderAirRoughCfgPPO as _RoughPPO


class FootTrackElSpiderAirHangCfg(_Rough):
    class env(_Rough.env):
        num_observations = 88

    class terrain(_Rough.terrain):
        mesh_type, measure_heights = 'plane', False

    class asset(_Rough.asset):
        self_collisions, fix_base_link = 0, True

    class init_state(_Rough.init_state):
        pos = [0.0, 0.0, 0.28]

    class commands(_Rough.commands):
        num_commands, resampling_time = 4, 1.
        heading_command, pose_command = True, False
        curriculum, max_curriculum = False, 2.5

        class ranges(_Rough.commands.ranges):
            lin_vel_x, lin_vel_y, ang_vel_yaw, heading = [-0.4, 0.4], [-0.3, 0.3], [-0.3, 0.3], [-0.5, 0.5]

    class domain_rand(_Rough.domain_rand):
        friction_range = [0.5, 1.5]

    class rewards(_Rough.rewards):
        base_height_target, max_contact_force, only_positive_rewards = 0.28, 500., True
        multi_stage_rewards, reward_stage_threshold, reward_min_stage, reward_max_stage = True, 5.0, 0, 0

        class scales(_Rough.rewards.scales):
            raibert_base_pos_track, raibert_base_quat_track, raibert_foot_pos_track = [-0.0, -6.0], [-0.0, -10.0], [-1.0, -3.0]
            orientation, base_height, feet_slip = [-0.0, 0.0], [-0.0, 0.0], [-0.0, -0.3]
            feet_air_time, lin_vel_z, ang_vel_xy, collision, dof_pos_limits = 0.8, -2.0, -0.05, -1.0, -1.0
            torques, action_rate, dof_acc = -0.00001, -0.001, -5e-8
            termination = dof_vel = feet_stumble = stand_still = -0.0

        class async_gait_scheduler:
            dof_align, dof_nominal_pos, reward_foot_z_align = 1.0, [0.0, 0.2], [0.0, 0.6]


class FootTrackElSpiderAirHangCfgPPO(_RoughPPO):
    class policy(_RoughPPO.policy):
        actor_hidden_dims = critic_hidden_dims = [128, 64, 32]
        activation = 'elu'

    class algorithm(_RoughPPO.algorithm):
        entropy_coef = 0.01

    class runner(_RoughPPO.runner):
        experiment_name, run_name, load_run = 'foot_track_elspider_air_hang', '', -1
        max_iterations, multi_stage_rewards = 3000, True
