"""ctypes mirror of `include/lgstep.h` (the C ABI of the env-step library).

Only declarations live here: struct layouts, enum values and function prototypes.  `tests/test_abi.py` parses the
header and checks every enum value and struct size against this file so the two cannot drift apart.
"""
import ctypes as C

LG_ABI_VERSION = 5
LG_MAX_LEGS, LG_JOINTS_PER_LEG, LG_MAX_JOINTS_PER_LEG = 6, 3, 6
LG_MAX_DOF = 18
LG_MAX_CP, LG_MAX_BODIES, LG_MAX_REWARD_TERMS, LG_MAX_INDEX_LIST = 8, 25, 32, 25
LG_MAX_SC_PAIRS = 96
SUPPORTED_LEG_COUNTS = (4, 6, 2)       # kernel instances of the library (csrc/lg_instance.h): 4 x 3, 6 x 3, 2 x 6 joints
JOINTS_PER_LEG_OF = {4: 3, 6: 3, 2: 6}
LG_LSTM_NPARAM = 969

LG_OK, LG_ERR_INVALID, LG_ERR_HIP, LG_ERR_UNSUPPORTED, LG_ERR_NO_DEVICE = 0, -1, -2, -3, -4
LG_F32, LG_I64, LG_U8, LG_I16, LG_I32, LG_F64 = 0, 1, 2, 3, 4, 5
LG_CTRL_P, LG_CTRL_V, LG_CTRL_T, LG_CTRL_ACTUATOR_NET = 0, 1, 2, 3
LG_MESH_PLANE, LG_MESH_HEIGHTFIELD, LG_MESH_TRIMESH = 0, 1, 2
LG_RNG_PHILOX, LG_RNG_INJECT = 0, 1
LG_SOLVER_PGS, LG_SOLVER_TGS = 0, 1
LG_FRICTION_CONE, LG_FRICTION_PYRAMID = 0, 1



def rand_slots(num_dof=12):
    """Slots of the per-env uniform-draw table (enum lg_rand_slot + the LG_RS_*_OF(dof) macros of lgstep.h)."""
    root_xy = 8 + num_dof
    return dict(LG_RS_CMD_CB=0, LG_RS_PUSH=4, LG_RS_LEVEL=6, LG_RS_DOF=8, LG_RS_ROOT_XY=root_xy, LG_RS_ROOT_VEL=root_xy + 2,
                LG_RS_CMD_RESET=root_xy + 8, LG_RS_NOISE=(root_xy + 8 + 3 + 3) & ~3)


def num_proprio(num_dof=12):
    """Observation entries in front of the height scan (LG_NUM_PROPRIO_OF): 48 for 12 DOF, 66 for 18."""
    return 12 + 3 * num_dof


RAND_SLOTS = rand_slots(12)            # the four-legged robots' table (20 / 22 / 28 / 32)
LG_RS_NOISE = RAND_SLOTS["LG_RS_NOISE"]

# reward term name (as in cfg.rewards.scales) -> lg_reward_term
REWARD_TERMS = [
    "action_rate", "ang_vel_xy", "base_foot_height", "base_height", "collision", "dof_acc", "dof_pos_limits", "dof_vel",
    "dof_vel_limits", "feet_air_time", "feet_contact_forces", "feet_slip", "feet_stumble", "feet_stumble_liftup",
    "four_footup", "gait_2_step", "gait_scheduler", "jump_air", "lin_vel_z", "orientation", "stand_still", "termination",
    "torque_limits", "torques", "tracking_ang_vel", "tracking_lin_vel",
    # class-specific variants: selected through an env class's `reward_term_variants`, never named in a config
    "orientation_load_adapt",
    # StandAnymal / StandGo2 only (anymal.py:301-308); the other overrides of those classes come with lg_config.reward_class
    "penalty_in_the_air", "async_gait_scheduler",
    "no_fly"]                                              # Cassie (cassie.py:42-45)
REWARD_CLASSES = {"base": 0, "stand": 1}                   # enum lg_reward_class
REWARD_TERM_ID = {n: i for i, n in enumerate(REWARD_TERMS)}

TENSOR_NAMES = [
    "root_states", "dof_state", "rigid_body_state", "contact_forces", "torques", "actions", "last_actions", "last_dof_vel",
    "last_root_vel", "commands", "base_lin_vel", "base_ang_vel", "projected_gravity", "base_lin_acc", "base_ang_acc",
    "feet_air_time", "feet_contact_time", "last_contacts", "measured_heights", "obs_buf", "rew_buf", "reset_buf",
    "time_out_buf", "episode_length_buf", "episode_sums", "terrain_levels", "terrain_types", "env_origins",
    "friction_coeffs", "base_mass_added", "sea_hidden_state", "sea_cell_state", "gait_idx", "gait_foot_z",
    "extras_episode", "rand_inject", "step_counters", "height_samples", "terrain_origins", "episode_stats", "command_ranges"]
TENSOR_ID = {n: i for i, n in enumerate(TENSOR_NAMES)}
LG_T_COUNT = len(TENSOR_NAMES)

f32, i32 = C.c_float, C.c_int32


class lg_robot_model(C.Structure):
    _fields_ = [
        ("num_legs", i32), ("num_joints_per_leg", i32), ("num_bodies", i32), ("has_foot_body", i32),
        ("base_mass", f32), ("base_com", f32 * 3), ("base_inertia", f32 * 6),
        ("joint_pos", (f32 * 3) * LG_MAX_JOINTS_PER_LEG * LG_MAX_LEGS), ("joint_rot", (f32 * 9) * LG_MAX_JOINTS_PER_LEG * LG_MAX_LEGS),
        ("joint_axis", (f32 * 3) * LG_MAX_JOINTS_PER_LEG * LG_MAX_LEGS),
        ("link_mass", (f32 * LG_MAX_JOINTS_PER_LEG) * LG_MAX_LEGS), ("link_com", (f32 * 3) * LG_MAX_JOINTS_PER_LEG * LG_MAX_LEGS),
        ("link_inertia", (f32 * 6) * LG_MAX_JOINTS_PER_LEG * LG_MAX_LEGS),
        ("foot_pos", (f32 * 3) * LG_MAX_LEGS), ("foot_rot", (f32 * 9) * LG_MAX_LEGS),
        ("dof_lower", f32 * LG_MAX_DOF), ("dof_upper", f32 * LG_MAX_DOF), ("dof_vel_limit", f32 * LG_MAX_DOF), ("torque_limit", f32 * LG_MAX_DOF),
        ("cp_count", i32 * LG_MAX_LEGS), ("cp_link", (i32 * LG_MAX_CP) * LG_MAX_LEGS), ("cp_body", (i32 * LG_MAX_CP) * LG_MAX_LEGS),
        ("cp_pos", (f32 * 3) * LG_MAX_CP * LG_MAX_LEGS), ("cp_radius", (f32 * LG_MAX_CP) * LG_MAX_LEGS),
        ("cp_slide", (f32 * 3) * LG_MAX_CP * LG_MAX_LEGS),
        ("num_sc_pairs", i32), ("sc_pairs", (i32 * 4) * LG_MAX_SC_PAIRS),
        ("feet_indices", i32 * LG_MAX_LEGS),
        ("num_penalised", i32), ("penalised_contact_indices", i32 * LG_MAX_INDEX_LIST),
        ("num_termination", i32), ("termination_contact_indices", i32 * LG_MAX_INDEX_LIST),
    ]


class lg_terrain(C.Structure):
    _fields_ = [
        ("mesh_type", i32), ("rows", i32), ("cols", i32),
        ("horizontal_scale", f32), ("vertical_scale", f32), ("border_size", f32), ("static_friction", f32),
        ("height_samples", C.POINTER(C.c_int16)),
        ("num_levels", i32), ("num_types", i32),
        ("terrain_origins", C.POINTER(f32)),
        ("env_length", f32),
        ("collision_mesh", C.c_void_p),
        ("grid_vertices", C.POINTER(f32)),
    ]


class lg_config(C.Structure):
    _fields_ = [
        ("abi_version", i32), ("num_envs", i32), ("num_obs", i32), ("num_height_points", i32), ("num_extra_obs", i32),
        ("sim_dt", f32), ("decimation", i32), ("gravity", f32 * 3),
        ("control_type", i32), ("action_scale", f32),
        ("p_gains", f32 * LG_MAX_DOF), ("d_gains", f32 * LG_MAX_DOF), ("default_dof_pos", f32 * LG_MAX_DOF),
        ("clip_actions", f32), ("clip_observations", f32),
        ("actuator_net", f32 * LG_LSTM_NPARAM), ("actuator_in_scale", f32 * 2), ("actuator_out_scale", f32),
        ("obs_scale_lin_vel", f32), ("obs_scale_ang_vel", f32), ("obs_scale_dof_pos", f32), ("obs_scale_dof_vel", f32),
        ("obs_scale_height", f32),
        ("measure_heights", i32), ("add_noise", i32),
        ("noise_scale_vec", C.POINTER(f32)), ("height_points", C.POINTER(f32)),
        ("heading_command", i32), ("resampling_steps", i32),
        ("cmd_lin_vel_x", f32 * 2), ("cmd_lin_vel_y", f32 * 2), ("cmd_ang_vel_yaw", f32 * 2), ("cmd_heading", f32 * 2),
        ("command_curriculum", i32), ("max_curriculum", f32),
        ("push_robots", i32), ("push_interval", i32), ("max_push_vel_xy", f32),
        ("num_reward_terms", i32), ("reward_term_ids", i32 * LG_MAX_REWARD_TERMS),
        ("reward_scales", f32 * LG_MAX_REWARD_TERMS), ("only_positive_rewards", i32), ("reward_class", i32),
        ("tracking_sigma", f32), ("base_height_target", f32), ("max_contact_force", f32), ("soft_dof_vel_limit", f32),
        ("soft_torque_limit", f32),
        ("dof_pos_limits", (f32 * 2) * LG_MAX_DOF),
        ("max_episode_length", f32), ("max_episode_length_s", f32),
        ("curriculum", i32), ("custom_origins", i32), ("max_terrain_level", i32),
        ("reset_z_from_terrain", i32),
        ("terminate_on_flip", i32),
        ("base_init_state", f32 * 13),
        ("gait_enabled", i32), ("gait_period", f32), ("gait_swing_height", f32), ("gait_foot_phases", f32 * LG_MAX_LEGS),
        ("solver_iterations", i32), ("contact_offset", f32), ("max_depenetration_velocity", f32), ("erp", f32),
        ("cfm", f32), ("solver_type", i32), ("friction_model", i32), ("self_collisions", i32),
        ("seed", C.c_uint64), ("rng_mode", i32),
        ("async_num_dof_sets", i32), ("async_dof_sets", (i32 * 3) * 4), ("async_dof_nominal", f32 * LG_MAX_DOF), ("async_dof_weight", f32 * LG_MAX_DOF),
        ("async_weights", f32 * 3), ("async_foot_z_align", f32), ("keep_small_commands", i32), ("feet_air_time_ungated", i32), ("inject_sim_state", i32),
    ]


class lg_tile_spec(C.Structure):
    _fields_ = [("kind", i32), ("max_height", i32), ("clip_lo", i32), ("clip_hi", i32), ("step_width", i32), ("step_height", i32),
                ("num_steps", i32), ("noise_lo", i32), ("noise_step", i32), ("noise_levels", i32), ("noise_coarse", i32),
                ("rect_min", i32), ("rect_max", i32), ("rect_count", i32), ("platform", i32), ("seed", C.c_uint32)]


LG_TILE_FLAT, LG_TILE_PYRAMID_SLOPE, LG_TILE_PYRAMID_STAIRS, LG_TILE_DISCRETE_OBSTACLES = 0, 1, 2, 3
LG_TILE_STEPPING_STONES, LG_TILE_GAP, LG_TILE_PIT = 4, 5, 6


class lg_pose_params(C.Structure):
    _fields_ = [("ranges", (f32 * 2) * 4), ("resampling_steps", i32), ("scale_orientation", f32), ("scale_base_height", f32),
                ("scale_termination", f32), ("only_positive_rewards", i32), ("max_episode_length_s", f32), ("clip_observations", f32),
                ("num_heights", i32), ("num_proprio", i32)]


class lg_foottrack_params(C.Structure):
    _fields_ = [("dt", f32), ("gait_period", f32), ("swing_ema", f32), ("reward_sigma", f32), ("swing_height", f32), ("phase_offsets", f32 * 6),
                ("base_bounds", (f32 * 6) * 2), ("foot_mean", f32 * 18), ("foot_sigma", f32 * 18), ("base_interval", f32), ("base_max_vel", f32),
                ("foot_interval", f32), ("foot_max_vel", f32), ("scales", f32 * 5), ("scale_termination", f32), ("only_positive_rewards", i32),
                ("max_episode_length_s", f32), ("clip_observations", f32), ("num_bodies", i32), ("feet_indices", i32 * 6), ("add_noise", i32)]


class lg_foottrack_state(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("base_pos", "base_quat", "base_pos_shift", "base_quat_shift", "base_x_world", "base_y_world", "foot_pos", "gait_idx",
                                          "gait_phases", "last_contacts", "bw_cur", "bw_tgt", "bw_timer", "fw_cur", "fw_tgt", "fw_timer")]


class lg_depth_params(C.Structure):
    _fields_ = [("width", i32), ("height", i32), ("resized_width", i32), ("resized_height", i32), ("buffer_len", i32),
                ("near_clip", f32), ("far_clip", f32), ("position", f32 * 3), ("quat_offset", f32 * 4)]


def declare_product(lib):
    """Prototypes of liblgstep.so (include/lgstep.h)."""
    vp = C.c_void_p
    lib.lg_abi_sizes.argtypes = [C.POINTER(i32)]
    lib.lg_abi_sizes.restype = None
    lib.lg_arena_bytes.argtypes = [C.POINTER(lg_config), C.POINTER(lg_robot_model), C.POINTER(lg_terrain)]
    lib.lg_arena_bytes.restype = C.c_size_t
    lib.lg_create.argtypes = [C.POINTER(lg_config), C.POINTER(lg_robot_model), C.POINTER(lg_terrain), C.c_int, vp]
    lib.lg_create.restype = vp
    lib.lg_get_tensor.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(i32), C.POINTER(i32)]
    lib.lg_get_tensor.restype = C.c_int
    lib.lg_step.argtypes = [vp, vp, vp]
    lib.lg_step.restype = C.c_int
    lib.lg_step_physics.argtypes = [vp, vp, vp]
    lib.lg_step_physics.restype = C.c_int
    lib.lg_step_subset.argtypes = [vp, vp, vp, i32, i32, vp]
    lib.lg_step_subset.restype = C.c_int
    lib.lg_set_reward_terms.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(f32), vp]
    lib.lg_set_reward_terms.restype = C.c_int
    lib.lg_set_async_gait.argtypes = [vp, C.POINTER(f32), f32, vp]
    lib.lg_set_async_gait.restype = C.c_int
    lib.lg_step_subset_physics.argtypes = [vp, vp, vp, i32, vp]
    lib.lg_step_subset_physics.restype = C.c_int
    lib.lg_post_physics_subset.argtypes = [vp, vp, i32, i32, vp]
    lib.lg_post_physics_subset.restype = C.c_int
    lib.lg_sync_main_to_rollout.argtypes = [vp, i32, f32, vp]
    lib.lg_sync_main_to_rollout.restype = C.c_int
    lib.lg_step_transition.argtypes = [vp, vp, vp, vp, f32, vp, vp, vp]
    lib.lg_step_transition.restype = C.c_int
    lib.lg_rollout_batch.argtypes = [vp, vp, i32, vp, i32, i32, f32, vp, vp]
    lib.lg_rollout_batch.restype = C.c_int
    lib.lg_compute_torques.argtypes = [vp, vp, vp]
    lib.lg_compute_torques.restype = C.c_int
    lib.lg_simulate.argtypes = [vp, vp]
    lib.lg_simulate.restype = C.c_int
    lib.lg_post_physics_step.argtypes = [vp, vp]
    lib.lg_post_physics_step.restype = C.c_int
    lib.lg_reset_idx.argtypes = [vp, vp, i32, i32, vp]
    lib.lg_reset_idx.restype = C.c_int
    lib.lg_set_state_indexed.argtypes = [vp, vp, vp, vp, i32, vp]
    lib.lg_set_state_indexed.restype = C.c_int
    lib.lg_gather_step_rows.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp]
    lib.lg_gather_step_rows.restype = C.c_int
    lib.lg_step_subset_rows.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]
    lib.lg_step_subset_rows.restype = C.c_int
    lib.lg_step_rollout.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp]
    lib.lg_step_rollout.restype = C.c_int
    lib.lg_set_extra_termination.argtypes = [vp, vp]
    lib.lg_set_extra_termination.restype = C.c_int
    lib.lg_set_extra_obs.argtypes = [vp, vp]
    lib.lg_set_extra_obs.restype = C.c_int
    lib.lg_terrain_generate.argtypes = [C.POINTER(lg_tile_spec), i32, i32, i32, i32, i32, f32, f32, f32, f32, vp, vp, vp]
    lib.lg_terrain_generate.restype = C.c_int
    lib.lg_heightfield_to_trimesh.argtypes = [vp, i32, i32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, vp, vp, vp]
    lib.lg_heightfield_to_trimesh.restype = C.c_int
    lib.lg_pose_layer_step.argtypes = [C.POINTER(lg_pose_params), i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.lg_pose_layer_step.restype = C.c_int
    lib.lg_foottrack_stray.argtypes = [i32, vp, vp, vp, vp, vp]
    lib.lg_foottrack_stray.restype = C.c_int
    lib.lg_foottrack_layer_step.argtypes = [C.POINTER(lg_foottrack_params), C.POINTER(lg_foottrack_state), i32, i32] + [vp] * 8 + [i32] + [vp] * 10
    lib.lg_foottrack_layer_step.restype = C.c_int
    lib.lg_mesh_create.argtypes = [C.POINTER(f32), C.c_int64, C.POINTER(i32), C.c_int64, C.c_int]
    lib.lg_mesh_create.restype = vp
    lib.lg_mesh_destroy.argtypes = [vp]
    lib.lg_mesh_destroy.restype = None
    lib.lg_mesh_info.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.lg_mesh_info.restype = C.c_int
    lib.lg_mesh_ray_lattice.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.lg_mesh_ray_lattice.restype = C.c_int
    lib.lg_mesh_contact_lattice.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.lg_mesh_contact_lattice.restype = C.c_int
    lib.lg_mesh_last_error.argtypes = [vp]
    lib.lg_mesh_last_error.restype = C.c_char_p
    lib.lg_raycast_mesh.argtypes = [vp, vp, vp, C.c_int64, f32, vp, vp, vp]
    lib.lg_raycast_mesh.restype = C.c_int
    lib.lg_mesh_query_sdf.argtypes = [vp, vp, C.c_int64, f32, vp, vp, vp]
    lib.lg_mesh_query_sdf.restype = C.c_int
    lib.lg_raycaster_update.argtypes = [vp, vp, vp, vp, i32, i32, f32, i32, vp, vp, vp, vp]
    lib.lg_raycaster_update.restype = C.c_int
    lib.lg_raycaster_update_subset.argtypes = [vp, vp, vp, vp, i32, f32, i32, vp, i32, vp, vp, vp, i32, vp]
    lib.lg_raycaster_update_subset.restype = C.c_int
    lib.lg_sdf_bodies_update.argtypes = [vp, vp, i32, vp, vp, i32, vp, i32, f32, vp, i32, vp, vp, vp]
    lib.lg_sdf_bodies_update.restype = C.c_int
    lib.lg_depth_camera_update.argtypes = [vp, C.POINTER(lg_depth_params), vp, vp, vp, i32, vp, vp, vp, vp, vp]
    lib.lg_depth_camera_update.restype = C.c_int
    lib.lg_profile_begin.argtypes = [vp, i32, i32]
    lib.lg_profile_begin.restype = C.c_int
    lib.lg_profile_end.argtypes = [vp, C.POINTER(f32), C.POINTER(i32)]
    lib.lg_profile_end.restype = C.c_int
    lib.lg_last_error.argtypes = [vp]
    lib.lg_last_error.restype = C.c_char_p
    lib.lg_destroy.argtypes = [vp]
    lib.lg_destroy.restype = None
    return lib


class lg_rollout(C.Structure):
    """include/lgpolicy.h: device pointers to the (T, n, .) rows of one collected rollout."""
    _fields_ = [(k, C.c_void_p) for k in ("observations", "actions", "rewards", "dones", "values", "actions_log_prob", "mu", "sigma",
                                          "last_values", "returns", "advantages")]


def declare_policy(lib):
    """Prototypes of the rollout-collection entry points (include/lgpolicy.h), same library."""
    vp = C.c_void_p
    lib.lg_mlp_create.argtypes = [i32, C.POINTER(i32), C.POINTER(C.POINTER(f32)), C.POINTER(C.POINTER(f32)), i32, C.c_int]
    lib.lg_mlp_create.restype = vp
    lib.lg_mlp_destroy.argtypes = [vp]
    lib.lg_mlp_destroy.restype = None
    lib.lg_mlp_last_error.argtypes = [vp]
    lib.lg_mlp_last_error.restype = C.c_char_p
    lib.lg_mlp_forward.argtypes = [vp, vp, C.c_int64, vp, vp]
    lib.lg_mlp_forward.restype = C.c_int
    lib.lg_policy_act.argtypes = [vp, vp, vp, vp, C.c_int64, vp, C.c_uint64, C.c_uint64, i32, vp, vp, vp, vp, vp]
    lib.lg_policy_act.restype = C.c_int
    lib.lg_compute_returns.argtypes = [vp, vp, vp, vp, i32, C.c_int64, f32, f32, i32, vp, vp, vp]
    lib.lg_compute_returns.restype = C.c_int
    lib.lg_collect_rollout.argtypes = [vp, vp, vp, vp, C.c_uint64, C.c_uint64, i32, f32, f32, i32, C.POINTER(lg_rollout), vp]
    lib.lg_collect_rollout.restype = C.c_int
    lib.lg_plan_from_nodes.argtypes = [vp, vp, C.c_int64, i32, i32, i32, vp, vp]
    lib.lg_plan_from_nodes.restype = C.c_int
    lib.lg_mppi_update.argtypes = [vp, vp, i32, i32, i32, i32, i32, f32, vp, vp, vp]
    lib.lg_mppi_update.restype = C.c_int
    u64 = C.c_uint64
    lib.lg_mppi_sample_plans.argtypes = [vp, vp, f32, vp, i32, i32, i32, i32, i32, u64, u64, vp, vp, vp]
    lib.lg_mppi_sample_plans.restype = C.c_int
    lib.lg_planner_diffuse.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, f32, u64, u64, vp, i32, f32, vp, vp, vp, vp, vp]
    lib.lg_planner_diffuse.restype = C.c_int
    return lib


POLICY_SYMBOLS = ["lg_mlp_create", "lg_mlp_destroy", "lg_mlp_last_error", "lg_mlp_forward", "lg_policy_act", "lg_compute_returns",
                  "lg_collect_rollout", "lg_plan_from_nodes", "lg_mppi_update", "lg_mppi_sample_plans", "lg_planner_diffuse"]
ACTIVATIONS = {"elu": 0, "relu": 1, "tanh": 2, "lrelu": 3, "selu": 4}

PRODUCT_SYMBOLS = ["lg_abi_sizes", "lg_arena_bytes", "lg_create", "lg_get_tensor", "lg_step", "lg_step_physics", "lg_step_subset", "lg_step_transition", "lg_sync_main_to_rollout", "lg_rollout_batch", "lg_compute_torques",
                   "lg_simulate", "lg_post_physics_step", "lg_reset_idx", "lg_profile_begin", "lg_profile_end", "lg_last_error",
                   "lg_destroy", "lg_set_extra_obs", "lg_mesh_create", "lg_mesh_destroy", "lg_mesh_info", "lg_mesh_ray_lattice", "lg_mesh_contact_lattice", "lg_mesh_last_error",
                   "lg_raycast_mesh", "lg_mesh_query_sdf", "lg_raycaster_update", "lg_depth_camera_update",
                   "lg_terrain_generate", "lg_heightfield_to_trimesh", "lg_pose_layer_step", "lg_set_reward_terms", "lg_set_async_gait", "lg_step_subset_physics", "lg_post_physics_subset", "lg_raycaster_update_subset", "lg_sdf_bodies_update", "lg_set_state_indexed", "lg_gather_step_rows", "lg_step_subset_rows", "lg_step_rollout", "lg_set_extra_termination", "lg_foottrack_stray", "lg_foottrack_layer_step"]
