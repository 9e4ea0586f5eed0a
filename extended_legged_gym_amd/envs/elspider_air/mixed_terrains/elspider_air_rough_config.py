"""ElSpider Air rough-terrain config (values of the reference's `envs/elspider_air/mixed_terrains/elspider_air_rough_config.py:33-171`).
The robot asset is the reduced model `resources/robots/el_mini.json` (tools/refgen/make_robot_models.py)."""
from extended_legged_gym_amd.envs.base.legged_robot_config import LeggedRobotCfg, LeggedRobotCfgPPO

_LEGS = ("RF", "RM", "RB", "LF", "LM", "LB")
_SCAN_X = [round(-0.8 + 0.1 * i, 1) for i in range(17)]      # 1.6 m x 1.0 m scan rectangle
_SCAN_Y = [round(-0.5 + 0.1 * i, 1) for i in range(11)]


class ElSpiderAirRoughCfg(LeggedRobotCfg):
    class env(LeggedRobotCfg.env):
        num_envs = 4096
        num_actions = 18
        num_observations = 253

    class terrain:          # (a section of its own in the reference as well: nothing is inherited from LeggedRobotCfg.terrain)
        use_terrain_obj = False
        terrain_file = ""
        mesh_type = 'trimesh'
        horizontal_scale = 0.1
        vertical_scale = 0.005
        border_size = 100
        curriculum = True
        static_friction = 1.0
        dynamic_friction = 1.0
        restitution = 0.
        measure_heights = True
        measured_points_x = _SCAN_X
        measured_points_y = _SCAN_Y
        selected = False
        terrain_kwargs = None
        max_init_terrain_level = 0
        terrain_length = 8.
        terrain_width = 8.
        num_rows = 10
        num_cols = 10
        terrain_proportions = [0.1, 0.1, 0.3, 0.3, 0.2]
        slope_treshold = 0.75

    class init_state(LeggedRobotCfg.init_state):
        pos = [0.0, 0.0, 0.4]
        default_joint_angles = {**{f"{leg}_HAA": 0.0 for leg in _LEGS}, **{f"{leg}_HFE": 0.6 for leg in _LEGS},
                                **{f"{leg}_KFE": 0.6 for leg in _LEGS}}

    class control(LeggedRobotCfg.control):
        stiffness = {'HAA': 80., 'HFE': 80., 'KFE': 80.}
        damping = {'HAA': 2., 'HFE': 2., 'KFE': 2.}
        action_scale = 0.5
        decimation = 4
        use_actuator_network = True
        actuator_net_file = "{LEGGED_GYM_ROOT_DIR}/resources/actuator_nets/anydrive_v3_lstm.pt"

    class asset(LeggedRobotCfg.asset):
        file = "{LEGGED_GYM_ROOT_DIR}/resources/robots/el_mini/urdf/el_mini.urdf"
        name = "elspider_air"
        foot_name = "FOOT"
        penalize_contacts_on = ["THIGH", "HIP"]
        terminate_after_contacts_on = ["trunk"]
        self_collisions = 0
        flip_visual_attachments = False

    class domain_rand(LeggedRobotCfg.domain_rand):
        randomize_base_mass = True
        added_mass_range = [-5., 5.]

    class rewards(LeggedRobotCfg.rewards):
        base_height_target = 0.25
        max_contact_force = 500.
        only_positive_rewards = True
        multi_stage_rewards = True
        reward_stage_threshold = 5.0
        reward_min_stage = 0
        reward_max_stage = 1

        class scales(LeggedRobotCfg.rewards.scales):
            termination = -0.0
            tracking_lin_vel = 1.0
            tracking_ang_vel = 0.5
            lin_vel_z = -2.0
            ang_vel_xy = -0.05
            torques = -0.00001
            dof_vel = -0.
            dof_acc = -2.5e-8
            feet_slip = [-0.0, -0.4]
            feet_air_time = 1.0
            collision = -1.
            feet_stumble = -0.0
            stand_still = -0.
            dof_pos_limits = -1.0
            orientation = -0.3
            action_rate = -0.001
            gait_2_step = -5.0
            base_height = [-2.0, -4.0]

        class async_gait_scheduler:
            dof_align = 0.5
            dof_nominal_pos = [0.1, 0.2]
            reward_foot_z_align = [0.2, 0.05]

        class raibert_planner:
            planner_type = 0
            base_pos_track = 1.0
            base_quat_track = 0.5
            foot_pos_track = 0.3


class ElSpiderAirRoughCfgPPO(LeggedRobotCfgPPO):
    class runner(LeggedRobotCfgPPO.runner):
        run_name = ''
        experiment_name = 'rough_elspider_air'
        load_run = -1
        max_iterations = 3000
        multi_stage_rewards = True
