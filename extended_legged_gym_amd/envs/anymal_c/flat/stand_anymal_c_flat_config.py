"""`stand_anymal_c_flat` task config (values of the reference's
`envs/anymal_c/flat/stand_anymal_c_flat_config.py:34-105`): ANYmal C upright on its hind feet, base x axis up."""
from extended_legged_gym_amd.envs.anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg, AnymalCRoughCfgPPO


class StandAnymalCFlatCfg(AnymalCRoughCfg):
    class env(AnymalCRoughCfg.env):
        num_observations = 48

    class terrain(AnymalCRoughCfg.terrain):
        mesh_type = 'plane'
        measure_heights = False

    class asset(AnymalCRoughCfg.asset):
        self_collisions = 0
        penalize_contacts_on = ["SHANK", "THIGH"]
        terminate_after_contacts_on = ["base"]

    class init_state(AnymalCRoughCfg.init_state):
        pos = [0.0, 0.0, 0.9]
        rot = [0.0, -0.707, 0.0, 0.707]      # pitched up by 90 degrees
        lin_vel = [0.0, 0.0, 0.0]
        ang_vel = [0.0, 0.0, 0.0]
        default_joint_angles = {
            "LF_HAA": 0.0, "LH_HAA": 0.0, "RF_HAA": -0.0, "RH_HAA": -0.0,
            "LF_HFE": 0.8, "LH_HFE": 1.0, "RF_HFE": 0.8, "RH_HFE": 1.0,
            "LF_KFE": -2.0, "LH_KFE": 0.8, "RF_KFE": -2.0, "RH_KFE": 0.8,
        }

    class rewards(AnymalCRoughCfg.rewards):
        max_contact_force = 700.
        base_height_target = 0.9

        class scales(AnymalCRoughCfg.rewards.scales):
            orientation = -4.0
            torques = -0.000025
            feet_air_time = 1.
            base_height = -4.
            collision = -2.
            penalty_in_the_air = -4.

    class commands(AnymalCRoughCfg.commands):
        heading_command = False
        resampling_time = 4.

        class ranges(AnymalCRoughCfg.commands.ranges):
            ang_vel_yaw = [-1.5, 1.5]

    class domain_rand(AnymalCRoughCfg.domain_rand):
        friction_range = [0., 1.5]


class StandAnymalCFlatCfgPPO(AnymalCRoughCfgPPO):
    class policy(AnymalCRoughCfgPPO.policy):
        actor_hidden_dims = [128, 64, 32]
        critic_hidden_dims = [128, 64, 32]
        activation = 'elu'

    class algorithm(AnymalCRoughCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(AnymalCRoughCfgPPO.runner):
        run_name = ''
        experiment_name = 'stand_flat_anymal_c'
        load_run = -1
        max_iterations = 1500
