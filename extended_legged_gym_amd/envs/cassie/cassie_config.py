"""Config tree of task `cassie` (values of the reference's `envs/cassie/cassie_config.py:33-112`, held to its registry by tests/test_task_configs.py).
Written from the joint list: the biped's two legs differ in the sign of the hip abduction only."""
from extended_legged_gym_amd.envs.base.legged_robot_config import LeggedRobotCfg, LeggedRobotCfgPPO

_JOINTS = ("hip_abduction", "hip_rotation", "hip_flexion", "thigh_joint", "ankle_joint", "toe_joint")
_STANCE = (0.1, 0., 1., -1.8, 1.57, -1.57)                      # left leg; the right leg mirrors the abduction
_SCAN = [round(0.1 * i, 1) for i in range(-5, 6)]               # 11 x 11 points, 1 m x 1 m


class CassieRoughCfg(LeggedRobotCfg):
    class env(LeggedRobotCfg.env):
        num_envs, num_observations, num_actions = 4096, 169, 12          # 48 + 121 heights

    class terrain(LeggedRobotCfg.terrain):
        measured_points_x, measured_points_y = list(_SCAN), list(_SCAN)

    class init_state(LeggedRobotCfg.init_state):
        pos = [0.0, 0.0, 1.]
        default_joint_angles = {f"{j}_{side}": (-q if (side == "right" and j == "hip_abduction") else q)
                                for side in ("left", "right") for j, q in zip(_JOINTS, _STANCE)}

    class control(LeggedRobotCfg.control):
        stiffness = dict(zip(_JOINTS, (100.0, 100.0, 200., 200., 200., 40.)))     # [N m / rad]
        damping = dict(zip(_JOINTS, (3.0, 3.0, 6., 6., 6., 1.)))                  # [N m s / rad]
        action_scale, decimation = 0.5, 4

    class asset(LeggedRobotCfg.asset):
        name, foot_name = "cassie", "toe"
        file = '{LEGGED_GYM_ROOT_DIR}/resources/robots/' + f'{name}/urdf/{name}.urdf'
        terminate_after_contacts_on = ['pelvis']
        flip_visual_attachments = False
        self_collisions = 1                                              # (the bitmask convention: 1 = off)

    class rewards(LeggedRobotCfg.rewards):
        soft_dof_pos_limit, soft_dof_vel_limit, soft_torque_limit = 0.95, 0.9, 0.9
        max_contact_force, only_positive_rewards = 300., False

        class scales(LeggedRobotCfg.rewards.scales):
            termination, tracking_ang_vel, no_fly, feet_air_time = -200., 1.0, 0.25, 5.
            torques, dof_acc, lin_vel_z, dof_pos_limits = -5.e-6, -2.e-7, -0.5, -1.
            dof_vel = ang_vel_xy = feet_contact_forces = -0.0


class CassieRoughCfgPPO(LeggedRobotCfgPPO):
    class algorithm(LeggedRobotCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(LeggedRobotCfgPPO.runner):
        run_name, experiment_name = '', 'rough_cassie'
