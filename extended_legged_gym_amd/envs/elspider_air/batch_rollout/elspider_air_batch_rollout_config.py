"""ElSpider Air main-rollout task configs: the values of the reference's `envs/elspider_air/batch_rollout/`
`elspider_air_batch_rollout_config.py:38-319` (confined two-layer mesh, collision-sphere URDF), `..._flat_config.py:36-169` (plane),
`elspider_air_dialmpc_config.py:36-213` (user-supplied OBJ terrain + SDF observations) and `elspider_air_dialmpc_flat_config.py:36-174`.
The four trees are held to the reference's by `tests/test_task_configs.py`.  Every leg has the same default pose and gains, so the
per-joint tables are written as comprehensions over the leg names."""
from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout_config import RobotBatchRolloutCfg, RobotBatchRolloutCfgPPO
from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout_percept_config import (
    RobotBatchRolloutPerceptCfg, RobotBatchRolloutPerceptCfgPPO)
from extended_legged_gym_amd.utils.gait_scheduler import AsyncGaitSchedulerCfg

LEGS = ("RF", "RM", "RB", "LF", "LM", "LB")               # the order the reference's dictionaries list them in
SORTED_LEGS = ("LB", "LF", "LM", "RB", "RF", "RM")        # the simulator's (alphabetical) joint / foot order


def default_pose(hfe, kfe):
    return {**{f"{l}_HAA": 0.0 for l in LEGS}, **{f"{l}_HFE": hfe for l in LEGS}, **{f"{l}_KFE": kfe for l in LEGS}}


SHANKS = ["trunk"] + [f"{l}_SHANK" for l in LEGS]
GAINS_P = {'HAA': 80., 'HFE': 80., 'KFE': 80.}
GAINS_D = {'HAA': 2., 'HFE': 2., 'KFE': 2.}
URDF = "{LEGGED_GYM_ROOT_DIR}/resources/robots/el_mini/urdf/el_mini.urdf"
CMD_RANGES = dict(lin_vel_x=[-1.5, 1.5], lin_vel_y=[-0.6, 0.6], ang_vel_yaw=[-0.6, 0.6], heading=[-3.14, 3.14])


class ElSpiderAirBatchRolloutCfg(RobotBatchRolloutPerceptCfg):
    class gait_scheduler:            # time-driven gait shaping of the class (no shipped config scales it)
        period = 1.4
        duty = 0.5
        foot_phases = [0.0, 0.5, 0.0, 0.5, 0.0, 0.5]
        dt = 0.005
        swing_height = 0.07
        track_sigma = 0.25

    class async_gait_scheduler(AsyncGaitSchedulerCfg):      # the tripods: (RF, RB, LM) and (LF, LB, RM)
        dof_names = [f"{l}_{j}" for l in SORTED_LEGS for j in ("HAA", "HFE", "KFE")]
        dof_align_sets = [[f"{l}_{j}" for l in tripod] for j in ("HFE", "KFE") for tripod in (("RF", "RB", "LM"), ("LF", "LB", "RM"))]
        dof_nominal_pos = [0.0, 1.0, 1.0] * 6
        foot_names = [f"{l}_FOOT" for l in SORTED_LEGS]
        foot_z_align_sets = [[f"{l}_FOOT" for l in tripod] for tripod in (("RF", "RB", "LM"), ("LF", "LB", "RM"))]

    class env(RobotBatchRolloutPerceptCfg.env):
        num_envs = 32            # main envs
        rollout_envs = 0
        num_observations = 66
        num_actions = 18
        episode_length_s = 20

    class terrain(RobotBatchRolloutPerceptCfg.terrain):
        mesh_type = 'confined_trimesh'
        measure_heights = False
        curriculum = True
        max_init_terrain_level = 2
        terrain_length = 6.
        terrain_width = 6.
        num_rows = 2
        num_cols = 1
        difficulty_scale = 0.6
        terrain_proportions = [0.2, 0.2, 0.3, 0.2, 0.1]
        confined_terrain_proportions = [0.0, 1.0, 0.0, 0.0]
        use_terrain_obj = False
        terrain_file = "resources/terrains/confined/confined_terrain.obj"
        random_origins = False
        origin_generation_max_attempts = 10000
        origins_x_range = [0.5, 1.5]
        origins_y_range = [-5, 1]
        height_clearance_factor = 1.5

    class raycaster:
        enable_raycast = False
        ray_pattern = "spherical2"
        num_rays = 10
        ray_angle = 30.0
        terrain_file = None
        max_distance = 10.0
        attach_yaw_only = False
        offset_pos = [0.0, 0.0, 0.0]
        spherical_num_azimuth = 16
        spherical_num_elevation = 8
        spherical2_num_points = 128
        spherical2_polar_axis = [0.0, 0.0, 1.0]

    class sdf:
        enable_sdf = False
        mesh_paths = []
        max_distance = 10.0
        enable_caching = True
        update_freq = 5
        query_bodies = list(SHANKS)
        compute_gradients = True
        compute_nearest_points = True
        include_in_obs = True

    class commands(RobotBatchRolloutPerceptCfg.commands):
        curriculum = False
        max_curriculum = 1.
        num_commands = 4
        resampling_time = 10.
        heading_command = False

        class ranges(RobotBatchRolloutPerceptCfg.commands.ranges):
            lin_vel_x, lin_vel_y, ang_vel_yaw, heading = (CMD_RANGES[k] for k in ("lin_vel_x", "lin_vel_y", "ang_vel_yaw", "heading"))

    class init_state(RobotBatchRolloutPerceptCfg.init_state):
        pos = [0.0, 0.0, 0.32]
        rot = [0.0, 0.0, 0.0, 1.0]
        default_joint_angles = default_pose(0.6, 0.6)

    class control(RobotBatchRolloutPerceptCfg.control):
        stiffness = dict(GAINS_P)
        damping = dict(GAINS_D)
        action_scale = 0.2            # ("Enable Network-0.3 | Disable Network-0.2")
        decimation = 4
        use_actuator_network = False
        actuator_net_file = "{LEGGED_GYM_ROOT_DIR}/resources/actuator_nets/anydrive_v3_lstm.pt"

    class asset(RobotBatchRolloutPerceptCfg.asset):
        file = "{LEGGED_GYM_ROOT_DIR}/resources/robots/el_mini/urdf/el_mini_collsp.urdf"
        name = "elspider"
        foot_name = "FOOT"
        penalize_contacts_on = ["base", "HIP", "THIGH", "SHANK"]
        terminate_after_contacts_on = []
        self_collisions = 0
        flip_visual_attachments = False

    class rewards(RobotBatchRolloutPerceptCfg.rewards):
        max_contact_force = 500.
        base_height_target = 0.34
        only_positive_rewards = False
        multi_stage_rewards = True
        tracking_sigma = 0.25

        class scales:
            termination = -0.0
            tracking_lin_vel = 3.0
            tracking_ang_vel = 0.5
            lin_vel_z = -2.0
            ang_vel_xy = -0.05
            orientation = -5.0
            torques = -0.00001
            dof_vel = -0.0
            dof_acc = -0.5e-8
            base_height = -8.0
            feet_slip = [-0.0, -0.4]
            feet_air_time = 0.8
            collision = -0.05
            feet_stumble = -0.4
            feet_stumble_liftup = 1.0
            action_rate = -0.001
            stand_still = -0.0
            dof_pos_limits = -1.0
            gait_2_step = -1.0

        class async_gait_scheduler:
            dof_align = 1.0
            dof_nominal_pos = [0.05, 0.2]
            reward_foot_z_align = [0.1, 0.6]

    class domain_rand(RobotBatchRolloutPerceptCfg.domain_rand):
        randomize_base_mass = True
        added_mass_range = [-5., 5.]
        rollout_envs_sync_pos_drift = 0.0

    class viewer(RobotBatchRolloutPerceptCfg.viewer):
        ref_env = 0
        pos = [2, 2, 2.0]
        lookat = [0.0, -2.0, 0.0]
        render_rollouts = False


class ElSpiderAirBatchRolloutCfgPPO(RobotBatchRolloutPerceptCfgPPO):
    class policy(RobotBatchRolloutPerceptCfgPPO.policy):
        actor_hidden_dims = [128, 64, 32]
        critic_hidden_dims = [128, 64, 32]
        activation = 'elu'

    class algorithm(RobotBatchRolloutPerceptCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(RobotBatchRolloutPerceptCfgPPO.runner):
        run_name = ''
        experiment_name = 'elspider_air_batch_rollout'
        load_run = -1
        max_iterations = 3000
        multi_stage_rewards = True


# ---- plane, el_mini.urdf, trunk contact ends the episode
class ElSpiderAirBatchRolloutFlatCfg(ElSpiderAirBatchRolloutCfg):
    class env(ElSpiderAirBatchRolloutCfg.env):
        num_envs = 32
        rollout_envs = 2
        env_spacing = 2.0
        num_observations = 66
        num_actions = 18
        episode_length_s = 20

    class raycaster(ElSpiderAirBatchRolloutCfg.raycaster):
        enable_raycast = False

    class sdf(ElSpiderAirBatchRolloutCfg.sdf):
        enable_sdf = False

    class terrain(ElSpiderAirBatchRolloutCfg.terrain):
        use_terrain_obj = False
        measure_heights = False
        curriculum = False
        mesh_type = 'plane'

    class commands(ElSpiderAirBatchRolloutCfg.commands):
        curriculum = False
        max_curriculum = 1.
        num_commands = 4
        resampling_time = 4.
        heading_command = False

        class ranges(ElSpiderAirBatchRolloutCfg.commands.ranges):
            lin_vel_x, lin_vel_y, ang_vel_yaw, heading = (CMD_RANGES[k] for k in ("lin_vel_x", "lin_vel_y", "ang_vel_yaw", "heading"))

    class init_state(ElSpiderAirBatchRolloutCfg.init_state):
        pos = [0.0, 0.0, 0.32]
        rot = [0.0, 0.0, 0.0, 1.0]
        default_joint_angles = default_pose(0.6, 0.6)

    class control(ElSpiderAirBatchRolloutCfg.control):
        stiffness = dict(GAINS_P)
        damping = dict(GAINS_D)
        action_scale = 0.2
        decimation = 4
        use_actuator_network = False

    class asset(ElSpiderAirBatchRolloutCfg.asset):
        file = URDF
        name = "elspider"
        foot_name = "FOOT"
        penalize_contacts_on = ["THIGH", "HIP"]
        terminate_after_contacts_on = ["trunk"]
        self_collisions = 1
        flip_visual_attachments = False

    class rewards(ElSpiderAirBatchRolloutCfg.rewards):
        max_contact_force = 500.
        base_height_target = 0.28
        only_positive_rewards = True
        multi_stage_rewards = True

        class scales():
            termination = -0.0
            tracking_lin_vel = 1.0
            tracking_ang_vel = 0.5
            lin_vel_z = -2.0
            ang_vel_xy = -0.05
            orientation = -5.0
            torques = -0.00001
            dof_vel = -0.0
            dof_acc = -0.5e-8
            base_height = -8.0
            feet_slip = -0.0
            feet_air_time = 0.8
            collision = -1.0
            action_rate = -0.001
            stand_still = -0.0
            dof_pos_limits = -1.0

    class domain_rand(ElSpiderAirBatchRolloutCfg.domain_rand):
        randomize_base_mass = True
        added_mass_range = [-5., 5.]
        rollout_envs_sync_pos_drift = 0.0

    class viewer(ElSpiderAirBatchRolloutCfg.viewer):
        ref_env = 0
        pos = [2.0, 0.0, 2.0]
        lookat = [0.5, 0.0, 0.]


class ElSpiderAirBatchRolloutFlatCfgPPO(ElSpiderAirBatchRolloutCfgPPO):
    class policy(ElSpiderAirBatchRolloutCfgPPO.policy):
        actor_hidden_dims = [128, 64, 32]
        critic_hidden_dims = [128, 64, 32]
        activation = 'elu'

    class algorithm(ElSpiderAirBatchRolloutCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(ElSpiderAirBatchRolloutCfgPPO.runner):
        run_name = ''
        experiment_name = 'elspider_air_batch_rollout_flat'
        load_run = -1
        max_iterations = 3000


# ---- DIAL-MPC tasks: position targets at action_scale 1, the AsyncGaitScheduler term scaled
class _DialMPCRewards:
    max_contact_force = 500.
    base_height_target = 0.24
    only_positive_rewards = False
    multi_stage_rewards = False
    tracking_sigma = 2.0
    SCALES = dict(termination=-0.0, tracking_lin_vel=4.0, tracking_ang_vel=2.0, lin_vel_z=-2.0, ang_vel_xy=-0.05, orientation=-5.0, torques=-0.00001,
                  dof_vel=-0., dof_acc=-0.5e-8, base_height=-8.0, feet_slip=-0.0, feet_air_time=0.8, collision=-1.0, action_rate=-0.001,
                  stand_still=-0., dof_pos_limits=-1.0)


class ElSpiderAirDialMPCCfg(ElSpiderAirBatchRolloutCfg):
    class env(ElSpiderAirBatchRolloutCfg.env):
        num_envs = 32
        rollout_envs = 0
        env_spacing = 2.0
        num_observations = 72          # 66 + the SDF values of the seven query bodies... of which the row carries 6 (reference value)
        num_actions = 18
        episode_length_s = 20

    class terrain(ElSpiderAirBatchRolloutCfg.terrain):
        use_terrain_obj = True
        terrain_file = "resources/terrains/confined/confined_terrain.obj"      # (not shipped by the reference: the user's mesh)
        measure_heights = False
        curriculum = False
        mesh_type = 'trimesh'
        terrain_length = 5.
        terrain_width = 5.
        random_origins = True
        origin_generation_max_attempts = 10000
        origins_x_range = [0.5, 0.5]
        origins_y_range = [-2.0, -2.0]
        height_clearance_factor = 2.0

    class raycaster:
        enable_raycast = False
        ray_pattern = "spherical"
        num_rays = 10
        ray_angle = 30.0
        terrain_file = "resources/terrains/confined/confined_terrain.obj"
        max_distance = 10.0
        attach_yaw_only = False
        offset_pos = [0.0, 0.0, 0.0]
        spherical_num_azimuth = 16
        spherical_num_elevation = 8

    class sdf:
        enable_sdf = True
        mesh_paths = ["resources/terrains/confined/confined_terrain.obj"]
        max_distance = 10.0
        enable_caching = True
        update_freq = 1
        query_bodies = list(SHANKS)
        compute_gradients = True
        compute_nearest_points = True
        include_in_obs = True

    class commands(ElSpiderAirBatchRolloutCfg.commands):
        curriculum = False
        max_curriculum = 1.
        num_commands = 4
        resampling_time = 4.
        heading_command = False

        class ranges(ElSpiderAirBatchRolloutCfg.commands.ranges):
            lin_vel_x, lin_vel_y, ang_vel_yaw, heading = (CMD_RANGES[k] for k in ("lin_vel_x", "lin_vel_y", "ang_vel_yaw", "heading"))

    class init_state(ElSpiderAirBatchRolloutCfg.init_state):
        pos = [0.0, 0.0, 0.42]
        rot = [0.0, 0.0, 0.0, 1.0]
        default_joint_angles = default_pose(1.0, 1.0)

    class control(ElSpiderAirBatchRolloutCfg.control):
        control_type = 'P'
        stiffness = dict(GAINS_P)
        damping = dict(GAINS_D)
        action_scale = 1.0
        decimation = 4
        use_actuator_network = False

    class asset(ElSpiderAirBatchRolloutCfg.asset):
        file = URDF
        name = "elspider"
        foot_name = "FOOT"
        penalize_contacts_on = ["THIGH", "HIP"]
        terminate_after_contacts_on = ["trunk"]
        self_collisions = 1
        flip_visual_attachments = False

    class rewards(ElSpiderAirBatchRolloutCfg.rewards):
        max_contact_force = _DialMPCRewards.max_contact_force
        base_height_target = _DialMPCRewards.base_height_target
        only_positive_rewards = _DialMPCRewards.only_positive_rewards
        multi_stage_rewards = _DialMPCRewards.multi_stage_rewards
        tracking_sigma = _DialMPCRewards.tracking_sigma

        class scales(ElSpiderAirBatchRolloutCfg.rewards.scales):
            locals().update(_DialMPCRewards.SCALES)
            async_gait_scheduler = -0.4

        class async_gait_scheduler:
            dof_align = 0.5
            dof_nominal_pos = [0.2, 0.2]
            reward_foot_z_align = [0.1, 0.05]

    class domain_rand(ElSpiderAirBatchRolloutCfg.domain_rand):
        randomize_base_mass = True
        added_mass_range = [-5., 5.]

    class viewer(ElSpiderAirBatchRolloutCfg.viewer):
        ref_env = 0
        pos = [2, 2, 2.0]
        lookat = [0.0, 0.0, 0.0]


class ElSpiderAirDialMPCCfgPPO(ElSpiderAirBatchRolloutCfgPPO):
    class policy(ElSpiderAirBatchRolloutCfgPPO.policy):
        actor_hidden_dims = [128, 64, 32]
        critic_hidden_dims = [128, 64, 32]
        activation = 'elu'

    class algorithm(ElSpiderAirBatchRolloutCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(ElSpiderAirBatchRolloutCfgPPO.runner):
        run_name = ''
        experiment_name = 'elspider_air_batch_rollout_flat'
        load_run = -1
        max_iterations = 3000


class ElSpiderAirDialMPCFlatCfg(RobotBatchRolloutCfg):
    """Derived from `RobotBatchRolloutCfg` in the reference, i.e. WITHOUT the `gait_scheduler` / `async_gait_scheduler` sections the env class
    reads in its constructor (`elspider_air_batch_rollout.py:66-93`): the reference cannot build this task (AttributeError), and neither
    can this package -- `ElSpiderAirBatchRollout` raises the same error with the reason."""
    class env(RobotBatchRolloutCfg.env):
        num_envs = 32
        rollout_envs = 0
        env_spacing = 2.0
        num_observations = 66
        num_actions = 18
        episode_length_s = 20

    class terrain(RobotBatchRolloutCfg.terrain):
        use_terrain_obj = False
        measure_heights = False
        curriculum = False
        mesh_type = 'plane'

    class commands(RobotBatchRolloutCfg.commands):
        curriculum = False
        max_curriculum = 1.
        num_commands = 4
        resampling_time = 4.
        heading_command = False

        class ranges(RobotBatchRolloutCfg.commands.ranges):
            lin_vel_x, lin_vel_y, ang_vel_yaw, heading = (CMD_RANGES[k] for k in ("lin_vel_x", "lin_vel_y", "ang_vel_yaw", "heading"))

    class init_state(RobotBatchRolloutCfg.init_state):
        pos = [0.0, 0.0, 0.28]
        rot = [0.0, 0.0, 0.0, 1.0]
        default_joint_angles = default_pose(1.0, 1.0)

    class control(RobotBatchRolloutCfg.control):
        control_type = 'P'
        stiffness = dict(GAINS_P)
        damping = dict(GAINS_D)
        action_scale = 1.0
        decimation = 4
        use_actuator_network = False

    class asset(RobotBatchRolloutCfg.asset):
        file = URDF
        name = "elspider"
        foot_name = "FOOT"
        penalize_contacts_on = ["THIGH", "HIP"]
        terminate_after_contacts_on = ["trunk"]
        self_collisions = 1
        flip_visual_attachments = False

    class rewards(RobotBatchRolloutCfg.rewards):
        max_contact_force = _DialMPCRewards.max_contact_force
        base_height_target = _DialMPCRewards.base_height_target
        only_positive_rewards = _DialMPCRewards.only_positive_rewards
        multi_stage_rewards = _DialMPCRewards.multi_stage_rewards
        tracking_sigma = _DialMPCRewards.tracking_sigma

        class scales(RobotBatchRolloutCfg.rewards.scales):
            locals().update(_DialMPCRewards.SCALES)
            async_gait_scheduler = -0.2

        class async_gait_scheduler:
            dof_align = 0.6
            dof_nominal_pos = [0.2, 0.2]
            reward_foot_z_align = [0.2, 0.05]

    class domain_rand(RobotBatchRolloutCfg.domain_rand):
        randomize_base_mass = True
        added_mass_range = [-5., 5.]

    class viewer(RobotBatchRolloutCfg.viewer):
        ref_env = 0
        pos = [-14.0, -14.0, 2.0]
        lookat = [-16.0, -16.0, 0.0]


class ElSpiderAirDialMPCFlatCfgPPO(RobotBatchRolloutCfgPPO):
    class policy(RobotBatchRolloutCfgPPO.policy):
        actor_hidden_dims = [128, 64, 32]
        critic_hidden_dims = [128, 64, 32]
        activation = 'elu'

    class algorithm(RobotBatchRolloutCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(RobotBatchRolloutCfgPPO.runner):
        run_name = ''
        experiment_name = 'elspider_air_batch_rollout_flat'
        load_run = -1
        max_iterations = 3000
