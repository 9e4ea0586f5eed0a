"""`pose_anymal_c_flat` task config (values of the reference's `envs/anymal_c/flat/pose_anymal_c_flat_config.py:33-117`)."""
from extended_legged_gym_amd.envs.anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg, AnymalCRoughCfgPPO


class PoseAnymalCFlatCfg(AnymalCRoughCfg):
    class env(AnymalCRoughCfg.env):
        num_observations = 52

    class terrain(AnymalCRoughCfg.terrain):
        mesh_type = 'plane'
        measure_heights = False

    class asset(AnymalCRoughCfg.asset):
        self_collisions = 0

    class normalization(AnymalCRoughCfg.normalization):
        class obs_scales(AnymalCRoughCfg.normalization.obs_scales):
            lin_vel = 2.0
            ang_vel = 0.25
            dof_pos = 1.0
            dof_vel = 0.05
            height_measurements = 5.0
        clip_observations = 100.
        clip_actions = 100.

    class rewards(AnymalCRoughCfg.rewards):
        max_contact_force = 350.

        class scales(AnymalCRoughCfg.rewards.scales):
            orientation = -5.0
            base_height = -30.0
            torques = -0.000025
            feet_air_time = 2.

    class commands(AnymalCRoughCfg.commands):
        curriculum = False
        max_curriculum = 1.
        num_commands = 8                 # lin_vel_x, lin_vel_y, ang_vel_yaw, heading, base yaw / pitch / roll shift, base height
        resampling_time = 4.
        heading_command = False
        pose_command = True

        class ranges:
            lin_vel_x = [-1.3, 1.3]
            lin_vel_y = [-1.0, 1.0]
            ang_vel_yaw = [-1, 1]
            heading = [-3.14, 3.14]
            base_yaw_shift = [-0., 0.]
            base_pitch_shift = [-0.5, 0.5]
            base_roll_shift = [-0.3, 0.3]
            base_height = [0.3, 0.7]

    class domain_rand(AnymalCRoughCfg.domain_rand):
        friction_range = [0., 1.5]


class PoseAnymalCFlatCfgPPO(AnymalCRoughCfgPPO):
    class policy(AnymalCRoughCfgPPO.policy):
        actor_hidden_dims = [128, 64, 32]
        critic_hidden_dims = [128, 64, 32]
        activation = 'elu'

    class algorithm(AnymalCRoughCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(AnymalCRoughCfgPPO.runner):
        run_name = ''
        experiment_name = 'pose_anymal_c_flat'
        load_run = -1
        max_iterations = 1000
