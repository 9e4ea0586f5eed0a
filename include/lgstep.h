/*
 * lgstep.h — C ABI of the MI355X-native batched legged-robot environment step.
 *
 * This is the drop-in boundary that replaces, for the LeggedRobot.step() hot path,
 * the Isaac Gym tensor API the reference calls from Python
 * (reference: legged_gym/legged_gym/envs/base/legged_robot.py):
 *
 *   lg_create / lg_get_tensor     <->  gym.create_sim + gym.acquire_*_tensor + gymtorch.wrap_tensor   (:258, :564-584)
 *   lg_step                       <->  LeggedRobot.step(): 4 x {_compute_torques, set_dof_actuation_force_tensor,
 *                                      simulate, refresh_dof_state_tensor} + post_physics_step          (:87-111, :113-153)
 *   lg_compute_torques            <->  LeggedRobot._compute_torques (:425-448) / Anymal._compute_torques (anymal.py:93-105)
 *   lg_simulate                   <->  gym.set_dof_actuation_force_tensor + gym.simulate + refresh_*    (:99-103, :118-120)
 *   lg_post_physics_step          <->  LeggedRobot.post_physics_step (:113-153) [+ Anymal.post_physics_step, anymal.py:107-110]
 *   lg_reset_idx                  <->  LeggedRobot.reset_idx (:162-213) incl. set_*_tensor_indexed (:463-465, :487-489)
 *
 * Rules of the ABI: plain C, plain pointers and sizes, no torch types.  Every tensor lives in ONE device arena;
 * the arena is either supplied by the host (e.g. a torch uint8 tensor, so torch owns the memory and views it
 * zero-copy) or hipMalloc'ed by the library.  All entry points are asynchronous on the hipStream_t passed in,
 * return 0 on success or a negative lg_status, and never throw.  One context per GPU, not re-entrant per context.
 */
#ifndef LGSTEP_H
#define LGSTEP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LG_ABI_VERSION 5

/* Robots of this library: a floating base carrying `lg_robot_model.num_legs` serial chains ("legs") of `num_joints_per_leg` revolute joints
 * each.  The library holds one instance of its kernels per supported topology -- 4 x 3 (ANYmal-B/C, A1, Go2), 6 x 3 (ElSpider Air, el_mini.urdf)
 * and 2 x 6 (Cassie, cassie.urdf: an open chain, the knee-spring joints are commented out in the reference's file) -- chosen by lg_create from the
 * model; the structs below are sized for the largest, and every (N, dof) / (N, legs) / (N, bodies) tensor has the model's own extents (12 / 4 / 17
 * for a quadruped with FOOT bodies, 18 / 6 / 25 for the hexapod, 12 / 2 / 13 for the biped). */
#define LG_MAX_LEGS 6
#define LG_JOINTS_PER_LEG 3    /* the three-joint instances */
#define LG_MAX_JOINTS_PER_LEG 6
#define LG_MAX_DOF 18          /* max over the instances of num_legs * num_joints_per_leg */
#define LG_MAX_CP 8            /* collision points per leg lane */
#define LG_MAX_SC_PAIRS 96     /* candidate sphere pairs of the self-collision pass */
#define LG_MAX_BODIES 25       /* base + 6 x (HIP, THIGH, SHANK, FOOT) */
#define LG_MAX_REWARD_TERMS 32
#define LG_MAX_INDEX_LIST 25     /* = LG_MAX_BODIES: a task may penalise contacts on every body (elspider_air_batch_rollout: base + 18 links) */
#define LG_LSTM_HIDDEN 8
#define LG_LSTM_NPARAM 969     /* 2-layer LSTM(2->8->8) + Linear(8->1), anydrive_v3_lstm */

/* status codes */
enum lg_status {
  LG_OK = 0,
  LG_ERR_INVALID = -1,     /* bad argument / config */
  LG_ERR_HIP = -2,         /* a HIP runtime call failed */
  LG_ERR_UNSUPPORTED = -3, /* config asks for something this build does not implement */
  LG_ERR_NO_DEVICE = -4
};

/* element types of arena tensors */
enum lg_dtype { LG_F32 = 0, LG_I64 = 1, LG_U8 = 2, LG_I16 = 3, LG_I32 = 4, LG_F64 = 5 };

/* control_type (legged_robot.py:438-447; anymal.py:93-105) */
enum lg_control { LG_CTRL_P = 0, LG_CTRL_V = 1, LG_CTRL_T = 2, LG_CTRL_ACTUATOR_NET = 3 };

/* terrain mesh type (legged_robot.py:259-274).  LG_MESH_HEIGHTFIELD collides against the int16 grid itself;
 * LG_MESH_TRIMESH collides against an arbitrary triangle mesh (gym.add_triangle_mesh, legged_robot.py:652-672, with
 * TerrainObj / TerrainConfined vertices) through closest-point queries on lg_terrain.collision_mesh, while the height
 * scan keeps reading height_samples (all zero for TerrainObj, terrain_obj.py:116). */
enum lg_mesh_type { LG_MESH_PLANE = 0, LG_MESH_HEIGHTFIELD = 1, LG_MESH_TRIMESH = 2 };

/* sim.physx.solver_type (legged_robot_config.py:262: "0: pgs, 1: tgs").  LG_SOLVER_PGS: num_position_iterations projected
 * Gauss-Seidel sweeps over the contact / limit rows at the frozen pose of the start of sim.dt, then one pose update.
 * LG_SOLVER_TGS (what the reference configures): temporal Gauss-Seidel -- num_position_iterations sub-intervals of
 * dt / iterations, each one pass over the rows with the separation re-evaluated from the pose advanced so far, then the pose
 * advances by the sub-interval. */
enum lg_solver { LG_SOLVER_PGS = 0, LG_SOLVER_TGS = 1 };

/* friction rows of a contact: LG_FRICTION_CONE = exact 2x2 tangential block projected on the disc |f_t| <= mu f_n;
 * LG_FRICTION_PYRAMID = PhysX's two scalar rows along the tangents, each clamped to +-mu f_n on its own */
enum lg_friction { LG_FRICTION_CONE = 0, LG_FRICTION_PYRAMID = 1 };

/* rng_mode: counter-based Philox4x32-10 in-kernel, or uniforms injected by the host (parity / golden tests) */
enum lg_rng { LG_RNG_PHILOX = 0, LG_RNG_INJECT = 1 };

/* slots of the per-env uniform-draw table (one row per env, LG_RS_NOISE_OF(dof) + num_obs columns); the slots behind the per-DOF draws
 * depend on the robot's DOF count: 12 DOF -> 20 / 22 / 28 / 32, 18 DOF -> 26 / 28 / 34 / 40 */
enum lg_rand_slot {
  LG_RS_CMD_CB = 0,     /* 3 draws: _resample_commands from _post_physics_step_callback (:391-393) */
  LG_RS_PUSH = 4,       /* 2 draws: _push_robots (:495) */
  LG_RS_LEVEL = 6,      /* 1 draw : randint_like in _update_terrain_curriculum (:516) */
  LG_RS_DOF = 8         /* dof draws: _reset_dofs (:459) */
};
#define LG_RS_ROOT_XY_OF(dof)   (LG_RS_DOF + (dof))                     /* 2 draws: _reset_root_states (:479) */
#define LG_RS_ROOT_VEL_OF(dof)  (LG_RS_ROOT_XY_OF(dof) + 2)             /* 6 draws: _reset_root_states (:485) */
#define LG_RS_CMD_RESET_OF(dof) (LG_RS_ROOT_VEL_OF(dof) + 6)            /* 3 draws: _resample_commands from reset_idx (:185) */
#define LG_RS_NOISE_OF(dof)     ((LG_RS_CMD_RESET_OF(dof) + 3 + 3) & ~3) /* num_obs draws: compute_observations (:252); a multiple of 4 */
#define LG_NUM_PROPRIO_OF(dof)  (12 + 3 * (dof))                        /* observation entries in front of the height scan (:237-244): 48 / 66 */

/* reward terms: legged_robot_rew_mixin.py:41-234 (+ Anymal._reward_gait_scheduler, anymal.py:112-114) */
enum lg_reward_term {
  LG_REW_ACTION_RATE = 0, LG_REW_ANG_VEL_XY, LG_REW_BASE_FOOT_HEIGHT, LG_REW_BASE_HEIGHT, LG_REW_COLLISION,
  LG_REW_DOF_ACC, LG_REW_DOF_POS_LIMITS, LG_REW_DOF_VEL, LG_REW_DOF_VEL_LIMITS, LG_REW_FEET_AIR_TIME,
  LG_REW_FEET_CONTACT_FORCES, LG_REW_FEET_SLIP, LG_REW_FEET_STUMBLE, LG_REW_FEET_STUMBLE_LIFTUP, LG_REW_FOUR_FOOTUP,
  LG_REW_GAIT_2_STEP, LG_REW_GAIT_SCHEDULER, LG_REW_JUMP_AIR, LG_REW_LIN_VEL_Z, LG_REW_ORIENTATION,
  LG_REW_STAND_STILL, LG_REW_TERMINATION, LG_REW_TORQUE_LIMITS, LG_REW_TORQUES, LG_REW_TRACKING_ANG_VEL,
  LG_REW_TRACKING_LIN_VEL,
  /* class-specific variants of a term (same name in cfg.rewards.scales, chosen by the env class) */
  LG_REW_ORIENTATION_LOAD_ADAPT,     /* LoadAdaptAnymal / LoadAdaptGo2._reward_orientation (anymal.py:140-143, go2.py:141-144) */
  LG_REW_PENALTY_IN_THE_AIR,         /* StandAnymal / StandGo2._reward_penalty_in_the_air (anymal.py:301-308): neither of feet 1, 3 in (filtered) contact */
  LG_REW_ASYNC_GAIT_SCHEDULER,       /* AnymalCBatchRollout / Go2BatchRollout._reward_async_gait_scheduler (anymal_c_batch_rollout.py:207-220) over
                                      * AsyncGaitScheduler's three terms (utils/gait_scheduler.py:151-175), see lg_config.async_* */
  LG_REW_NO_FLY,                     /* Cassie._reward_no_fly (envs/cassie/cassie.py:42-45): exactly one foot with contact_forces z > 0.1 */
  LG_REW_COUNT
};

/* lg_config.reward_class: the env class whose overrides of the shared term names apply.  LG_RC_STAND = StandAnymal / StandGo2
 * (anymal.py:253-308, go2.py): the robot stands on feet_indices[1] and [3] with its x axis up, so
 *   ang_vel_xy        sums base_ang_vel[1:]^2,            orientation      sums projected_gravity[1:]^2,
 *   tracking_lin_vel  tracks commands[:2] against -base_lin_vel[1:],   tracking_ang_vel tracks commands[2] against base_ang_vel[0],
 *   feet_air_time     runs on feet 1 and 3 only (columns 1, 3 of feet_air_time / last_contacts; feet_contact_time untouched). */
enum lg_reward_class { LG_RC_BASE = 0, LG_RC_STAND = 1 };

/* arena tensors (names follow the reference's attribute names, legged_robot.py:559-647, base_task.py:71-79) */
enum lg_tensor_id {
  LG_T_ROOT_STATES = 0,     /* (N,13) f32  pos3, quat xyzw4, lin vel3, ang vel3 — world frame          */
  LG_T_DOF_STATE,           /* (N,dof,2) f32 pos, vel                                                  */
  LG_T_RIGID_BODY_STATE,    /* (N,B,13) f32                                                            */
  LG_T_CONTACT_FORCES,      /* (N,B,3) f32 net contact force per body, world frame, last substep       */
  LG_T_TORQUES,             /* (N,dof) f32                                                             */
  LG_T_ACTIONS,             /* (N,dof) f32 clipped actions                                             */
  LG_T_LAST_ACTIONS,        /* (N,dof)                                                                 */
  LG_T_LAST_DOF_VEL,        /* (N,dof)                                                                 */
  LG_T_LAST_ROOT_VEL,       /* (N,6)                                                                   */
  LG_T_COMMANDS,            /* (N,4)                                                                   */
  LG_T_BASE_LIN_VEL,        /* (N,3)                                                                   */
  LG_T_BASE_ANG_VEL,        /* (N,3)                                                                   */
  LG_T_PROJECTED_GRAVITY,   /* (N,3)                                                                   */
  LG_T_BASE_LIN_ACC,        /* (N,3)                                                                   */
  LG_T_BASE_ANG_ACC,        /* (N,3)                                                                   */
  LG_T_FEET_AIR_TIME,       /* (N,legs)                                                                */
  LG_T_FEET_CONTACT_TIME,   /* (N,legs)                                                                */
  LG_T_LAST_CONTACTS,       /* (N,legs) u8                                                             */
  LG_T_MEASURED_HEIGHTS,    /* (N,P) f32                                                               */
  LG_T_OBS_BUF,             /* (N,num_obs) f32                                                         */
  LG_T_REW_BUF,             /* (N) f32                                                                 */
  LG_T_RESET_BUF,           /* (N) u8                                                                  */
  LG_T_TIME_OUT_BUF,        /* (N) u8                                                                  */
  LG_T_EPISODE_LENGTH_BUF,  /* (N) i64                                                                 */
  LG_T_EPISODE_SUMS,        /* (LG_MAX_REWARD_TERMS,N) f32, row k < K = episode_sums[reward_names[k]]   */
  LG_T_TERRAIN_LEVELS,      /* (N) i64                                                                 */
  LG_T_TERRAIN_TYPES,       /* (N) i64                                                                 */
  LG_T_ENV_ORIGINS,         /* (N,3) f32                                                               */
  LG_T_FRICTION_COEFFS,     /* (N) f32 per-env shape friction (legged_robot.py:332-343)                */
  LG_T_BASE_MASS_ADDED,     /* (N) f32 per-env payload (legged_robot.py:381-383)                       */
  LG_T_SEA_HIDDEN_STATE,    /* (2,N*dof,8) f32 (anymal.py:88)                                          */
  LG_T_SEA_CELL_STATE,      /* (2,N*dof,8) f32 (anymal.py:89)                                          */
  LG_T_GAIT_IDX,            /* (N) f32 (gait_scheduler.py:60)                                          */
  LG_T_GAIT_FOOT_Z,         /* (N,legs) f32 foot heights handed to GaitScheduler.step by the previous step (gait_scheduler.py:71) */
  LG_T_EXTRAS_EPISODE,      /* (LG_MAX_REWARD_TERMS+1) f32: [k < K] mean episode sum / max_episode_length_s per reward term
                               over the envs reset in the most recent step that reset any (:200-203); [K] = mean terrain level */
  LG_T_RAND_INJECT,         /* (N, LG_RS_NOISE_OF(dof)+num_obs) f32, only read when rng_mode == LG_RNG_INJECT   */
  LG_T_STEP_COUNTERS,       /* (4) i64: [0] common_step_counter, [1] #envs reset by the last step       */
  LG_T_HEIGHT_SAMPLES,      /* (rows, cols) i16, read-only terrain grid (lg_create derives the scan's table from it) */
  LG_T_TERRAIN_ORIGINS,     /* (levels, types, 3) f32                                                  */
  LG_T_EPISODE_STATS,       /* (4) f64 running totals since lg_create: sum of finished-episode returns, sum of their
                               lengths, #finished episodes, #env-steps — what a rank contributes to the cross-GPU
                               all-gather of episode statistics                                         */
  LG_T_COMMAND_RANGES,      /* (4,2) f32 [lin_vel_x, lin_vel_y, ang_vel_yaw, heading] x [min, max]: the ranges commands
                               are drawn from (:405-423); starts as lg_config.cmd_*, widened in place by the command
                               curriculum (:520-533), writable by the host                               */
  LG_T_COUNT
};

typedef struct lg_robot_model {
  int32_t num_legs;                   /* 4, 6 or 2 */
  int32_t num_joints_per_leg;         /* 3 (num_legs 4, 6) or 6 (num_legs 2) */
  int32_t num_bodies;                 /* 1 + num_legs*(num_joints_per_leg + has_foot_body) */
  int32_t has_foot_body;              /* FOOT links kept by dont_collapse="true" */
  float base_mass;
  float base_com[3];                  /* base frame */
  float base_inertia[6];              /* about COM, base axes: xx xy xz yy yz zz */
  /* DOF / leg order = Isaac Gym asset order (alphabetical depth-first): leg l, joint j -> dof num_joints_per_leg*l+j */
  float joint_pos[LG_MAX_LEGS][LG_MAX_JOINTS_PER_LEG][3];   /* joint frame origin in parent body frame */
  float joint_rot[LG_MAX_LEGS][LG_MAX_JOINTS_PER_LEG][9];   /* row-major rotation parent body -> joint frame at q = 0 */
  float joint_axis[LG_MAX_LEGS][LG_MAX_JOINTS_PER_LEG][3];  /* unit axis, joint frame */
  float link_mass[LG_MAX_LEGS][LG_MAX_JOINTS_PER_LEG];
  float link_com[LG_MAX_LEGS][LG_MAX_JOINTS_PER_LEG][3];    /* link frame */
  float link_inertia[LG_MAX_LEGS][LG_MAX_JOINTS_PER_LEG][6];/* about COM, link axes */
  float foot_pos[LG_MAX_LEGS][3];     /* FOOT body frame in the last link's frame */
  float foot_rot[LG_MAX_LEGS][9];
  float dof_lower[LG_MAX_DOF], dof_upper[LG_MAX_DOF]; /* hard limits; lower >= upper means unlimited */
  float dof_vel_limit[LG_MAX_DOF];
  float torque_limit[LG_MAX_DOF];     /* URDF effort */
  /* collision spheres, grouped by the leg lane that owns them; link -1 = base, 0..J-1 = leg link, J = foot body (J = num_joints_per_leg) */
  int32_t cp_count[LG_MAX_LEGS];
  int32_t cp_link[LG_MAX_LEGS][LG_MAX_CP];
  int32_t cp_body[LG_MAX_LEGS][LG_MAX_CP];              /* rigid-body index the contact force is reported on */
  float cp_pos[LG_MAX_LEGS][LG_MAX_CP][3];              /* in the owning link's frame (foot: last link frame) */
  float cp_radius[LG_MAX_LEGS][LG_MAX_CP];
  /* Capsule segments (asset.replace_cylinder_with_capsule, legged_robot_config.py:171): a capsule is a chain of spheres along its axis; cp_slide is the
   * vector (link frame) from a sphere to the NEXT one of its chain, zero for the last and for lone spheres.  The faces of a piecewise-planar terrain meet
   * a capsule at its spheres first; what would pass between two spheres is a convex edge, and the creases of a height grid lie on its lines: the slot of
   * a sphere with a segment also holds the contact of [cp_pos, cp_pos + cp_slide] with the pieces of the first x line and first y line its ground track
   * crosses (exact segment-segment closest points), whenever that is deeper than the sphere's own contact.  Plane terrains have no lines.  Triangle-mesh terrains: on a GRID mesh
   * (lg_terrain.grid_vertices: the slope-corrected triangulation of the height grid, what the registered rough tasks collide with) the same rule with the mesh's own
   * edges -- the edge (L, j)-(L, j + 1) of the first lattice line of each axis, wherever the correction put its vertices; upright or collapsed edges and candidates
   * whose axis lies below the edge by more than the radius are left out; LG_MESH_CAPS=0 switches it off -- ; on other meshes the spheres stand alone. */
  float cp_slide[LG_MAX_LEGS][LG_MAX_CP][3];
  /* Self-collision (asset.self_collisions = 0, legged_robot_config.py:170,176: PhysX collides the actor's own shapes, parent-child links excepted):
   * the sphere pairs (leg a, slot a, leg b, slot b) the pass tests every substep -- the pairs the host found reachable within the joint limits.
   * Read only when lg_config.self_collisions is set. */
  int32_t num_sc_pairs;
  int32_t sc_pairs[LG_MAX_SC_PAIRS][4];
  int32_t feet_indices[LG_MAX_LEGS];
  int32_t num_penalised, penalised_contact_indices[LG_MAX_INDEX_LIST];
  int32_t num_termination, termination_contact_indices[LG_MAX_INDEX_LIST];
} lg_robot_model;

typedef struct lg_terrain {
  int32_t mesh_type;                  /* lg_mesh_type */
  int32_t rows, cols;                 /* height_samples shape (tot_rows, tot_cols); 0 for plane */
  float horizontal_scale, vertical_scale, border_size;
  float static_friction;
  const int16_t* height_samples;      /* HOST pointer, rows*cols, copied at lg_create */
  int32_t num_levels, num_types;      /* terrain_origins shape (num_levels, num_types, 3) */
  const float* terrain_origins;       /* HOST pointer, copied at lg_create */
  float env_length;                   /* terrain.env_length, curriculum distance threshold (:510) */
  const struct lg_mesh* collision_mesh; /* LG_MESH_TRIMESH: handle from lg_mesh_create (same device), must outlive the ctx */
  const float* grid_vertices;         /* HOST pointer or NULL, rows*cols*3, copied at lg_create.  Set when the collision mesh is the
                                       * regular triangulation of a height grid (convert_heightfield_to_trimesh, terrain.py:77-80: cell
                                       * (i, j) -> triangles (v0, v3, v1), (v0, v2, v3); vertices moved by at most one cell in x / y by the
                                       * slope correction), in world coordinates: contact queries then index the cells around a sphere
                                       * directly instead of walking the BVH -- same triangles, same closest points, same tie rule */
} lg_terrain;

typedef struct lg_config {
  int32_t abi_version;
  int32_t num_envs, num_obs, num_height_points;
  int32_t num_extra_obs;              /* observation columns appended after the height scan from a caller-owned device
                                         buffer (lg_set_extra_obs): LeggedRobotRayCast's raycast_distances */
  float sim_dt; int32_t decimation; float gravity[3];
  /* control */
  int32_t control_type; float action_scale;
  float p_gains[LG_MAX_DOF], d_gains[LG_MAX_DOF], default_dof_pos[LG_MAX_DOF];
  float clip_actions, clip_observations;
  float actuator_net[LG_LSTM_NPARAM]; /* w_ih0(32x2) w_hh0(32x8) b_ih0(32) b_hh0(32) w_ih1(32x8) w_hh1(32x8) b_ih1 b_hh1 lin_w(8) lin_b(1) */
  float actuator_in_scale[2], actuator_out_scale;
  /* observations (legged_robot.py:234-252) */
  float obs_scale_lin_vel, obs_scale_ang_vel, obs_scale_dof_pos, obs_scale_dof_vel, obs_scale_height;
  int32_t measure_heights, add_noise;
  const float* noise_scale_vec;       /* HOST, num_obs */
  const float* height_points;         /* HOST, num_height_points x 2 (x, y) base-frame scan grid (:884-898) */
  /* commands (:405-423) */
  int32_t heading_command, resampling_steps;
  float cmd_lin_vel_x[2], cmd_lin_vel_y[2], cmd_ang_vel_yaw[2], cmd_heading[2];   /* initial LG_T_COMMAND_RANGES */
  int32_t command_curriculum;         /* commands.curriculum (:178-179, :520-533) */
  float max_curriculum;               /* commands.max_curriculum */
  /* domain randomisation (:491-496) */
  int32_t push_robots, push_interval; float max_push_vel_xy;
  /* rewards (:215-232, :649-674) — terms in evaluation order, scales already multiplied by dt */
  int32_t num_reward_terms;           /* K, incl. termination if present (always last row) */
  int32_t reward_term_ids[LG_MAX_REWARD_TERMS];
  float reward_scales[LG_MAX_REWARD_TERMS];
  int32_t only_positive_rewards;
  int32_t reward_class;               /* enum lg_reward_class */
  float tracking_sigma, base_height_target, max_contact_force, soft_dof_vel_limit, soft_torque_limit;
  float dof_pos_limits[LG_MAX_DOF][2];/* soft limits (:366-370) */
  /* episode / curriculum */
  float max_episode_length, max_episode_length_s;
  int32_t curriculum, custom_origins, max_terrain_level;
  int32_t reset_z_from_terrain;       /* RobotBatchRollout._reset_root_states (robot_batch_rollout.py:1379-1391): with custom
                                       * origins, root z = height sample under the drawn (x, y) + init z */
  int32_t terminate_on_flip;          /* AnymalCBatchRollout.check_termination (anymal_c_batch_rollout.py:192-198): also reset
                                       * when projected_gravity.z > 0 (robot upside down) */
  float base_init_state[13];
  /* gait scheduler (anymal.py:59-63, gait_scheduler.py:63-81) */
  int32_t gait_enabled; float gait_period, gait_swing_height, gait_foot_phases[LG_MAX_LEGS];
  /* contact solver (legged_robot_config.py:250-267) */
  int32_t solver_iterations;          /* physx.num_position_iterations */
  float contact_offset, max_depenetration_velocity, erp, cfm;
  int32_t solver_type;                /* enum lg_solver: physx.solver_type */
  int32_t friction_model;             /* enum lg_friction */
  int32_t self_collisions;            /* 1 = asset.self_collisions == 0 (legged_robot_config.py:176, create_actor at legged_robot.py:792): PhysX collides
                                       * the robot's own shapes.  Selects the kernel instance with the self-collision pass: every substep the pairs
                                       * lg_robot_model.sc_pairs are tested, and the (up to two) deepest ones closer than contact_offset become
                                       * frictionless unilateral rows between the two bodies, relaxed behind the terrain contacts of each pass. */
  /* rng */
  uint64_t seed; int32_t rng_mode;
  /* AsyncGaitScheduler (utils/gait_scheduler.py:97-175; sections cfg.async_gait_scheduler / cfg.rewards.async_gait_scheduler):
   *   dof_align       = sum over sets of the unbiased std of dof_pos[set]            (sets of DOF indices, -1 padded)
   *   dof_nominal_pos = sum_d weight[d] * (dof_pos[d] - nominal[d])^2
   *   foot_z_align    = sum over foot sets of the unbiased std of foot z -- of the feet positions the scheduler object captured when
   *                     the env was built (the reference's `foot_positions` attribute is re-bound every step, the scheduler keeps the
   *                     first tensor): a constant of the spawn pose, async_foot_z_align
   *   term = dof_align * w[0] + dof_nominal_pos * w[1] + foot_z_align * w[2]         (w: the stage's weights, lg_set_async_gait) */
  int32_t async_num_dof_sets, async_dof_sets[4][3];
  float async_dof_nominal[LG_MAX_DOF], async_dof_weight[LG_MAX_DOF];
  float async_weights[3], async_foot_z_align;
  int32_t keep_small_commands;        /* 1 = FootTrackElSpider._resample_commands (elspider.py:620-637): drawn xy commands with norm <= 0.2 are NOT zeroed */
  int32_t feet_air_time_ungated;      /* 1 = FootTrackElSpider._reward_feet_air_time (elspider.py:639-650): no "zero command, zero reward" factor */
  int32_t inject_sim_state;           /* parity tests only (like LG_RNG_INJECT): lg_step's post-physics half takes the post-simulation state
                                       * from what the caller put into LG_T_ROOT_STATES / DOF_STATE / TORQUES / CONTACT_FORCES / RIGID_BODY_STATE
                                       * before the call -- the way the recording harness injected it under the reference's step() -- and
                                       * leaves those tensors as they are; what the substeps computed from it is discarded */
} lg_config;

typedef struct lg_ctx lg_ctx;

/* ---- terrain construction on the device (reference utils/terrain.py:39-173 + isaacgym.terrain_utils, numpy on the host).
 * lg_terrain_generate fills the (tot_rows, tot_cols) int16 height grid tile by tile -- the curriculum / random layouts of Terrain
 * (`terrain.py:82-173`) hand it one lg_tile_spec per (row, col) -- and the (num_rows, num_cols, 3) tile origins
 * [(row + 0.5) length, (col + 0.5) width, max height of the central 2 m x 2 m window] (`:156-173`).  The deterministic generators
 * (pyramid slope, pyramid stairs, gap, pit, flat) produce the host generators' integers exactly; the random ones (uniform noise, discrete
 * obstacles, stepping stones) draw from Philox4x32-10 keyed by the tile seed instead of numpy's global stream: same distributions, other samples.
 * lg_heightfield_to_trimesh is convert_heightfield_to_trimesh (`terrain.py:77-80`) -- regular triangulation, cell (i, j) ->
 * (v0, v3, v1), (v0, v2, v3), vertices next to a step steeper than slope_threshold moved by one cell -- bit for bit.
 * Device pointers; asynchronous on `stream`. */
enum lg_tile_kind { LG_TILE_FLAT = 0, LG_TILE_PYRAMID_SLOPE = 1, LG_TILE_PYRAMID_STAIRS = 2, LG_TILE_DISCRETE_OBSTACLES = 3,
                    LG_TILE_STEPPING_STONES = 4,   /* rect_min = stone size, rect_max = stone distance (pixels), max_height, clip_lo = depth (units), platform */
                    LG_TILE_GAP = 5,               /* terrain.py:125-136 gap_terrain: rect_min = gap width, platform (pixels) */
                    LG_TILE_PIT = 6 };             /* terrain.py:139-148 pit_terrain: max_height = depth (units), platform = HALF the pit's width (pixels) */
typedef struct lg_tile_spec {
  int32_t kind;               /* enum lg_tile_kind */
  int32_t max_height;         /* PYRAMID_SLOPE: int(slope * hs / vs * width / 2) (may be negative); OBSTACLES: int(max_height / vs) */
  int32_t clip_lo, clip_hi;   /* PYRAMID_SLOPE: the platform clip np.clip(h, min(edge, 0), max(edge, 0)) */
  int32_t step_width, step_height, num_steps;   /* PYRAMID_STAIRS: pixels, height units, number of nested squares */
  int32_t noise_lo, noise_step, noise_levels;   /* uniform noise added on top (0 levels = none): level k = noise_lo + k * noise_step */
  int32_t noise_coarse;       /* pixels of the fine grid per coarse noise sample */
  int32_t rect_min, rect_max, rect_count, platform;   /* OBSTACLES: sizes in pixels (step 4 as the host generator), flat platform in the middle */
  uint32_t seed;
} lg_tile_spec;
int lg_terrain_generate(const lg_tile_spec* tiles_host, int32_t num_rows, int32_t num_cols, int32_t tile_len_px, int32_t tile_wid_px,
                        int32_t border_px, float horizontal_scale, float vertical_scale, float env_length, float env_width,
                        int16_t* heights, float* origins, void* stream);
int lg_heightfield_to_trimesh(const int16_t* heights, int32_t rows, int32_t cols, double step_x, double step_y, double stop_x, double stop_y,
                              double horizontal_scale, double vertical_scale, double slope_threshold_units /* < 0: none */,
                              float* vertices, uint32_t* triangles, void* stream);

/* sizes of the ABI structs as compiled into the library, for binding-side sanity checks */
void lg_abi_sizes(int32_t out[4]); /* {LG_ABI_VERSION, sizeof(lg_config), sizeof(lg_robot_model), sizeof(lg_terrain)} */

/* Bytes of device arena a context with this configuration needs (256-B aligned tensors). */
size_t lg_arena_bytes(const lg_config* cfg, const lg_robot_model* model, const lg_terrain* terrain);

/* Create a context on `device_id`.  `arena` is a device pointer to lg_arena_bytes() bytes owned by the host
 * (zero-initialised by the library), or NULL to let the library hipMalloc it.  Returns NULL on failure;
 * lg_last_error(NULL) then holds the reason. */
lg_ctx* lg_create(const lg_config* cfg, const lg_robot_model* model, const lg_terrain* terrain,
                  int device_id, void* arena);

/* Device pointer / shape / dtype of an arena tensor.  Pointers stay valid until lg_destroy. */
int lg_get_tensor(lg_ctx* ctx, int tensor_id, void** dptr, int64_t shape[4], int32_t* ndim, int32_t* dtype);

/* One policy step: clip actions, `decimation` x (actuator torques + articulated dynamics + contact), then the
 * fused post-physics step.  `actions` is a device pointer to (N,dof) f32.  Replaces legged_robot.py:93-110. */
int lg_step(lg_ctx* ctx, const float* actions, void* stream);

/* The physics half of lg_step only (LR:93-103): used when a sensor update (ray caster, LR raycast :219-230) must run
 * between the last substep and post_physics_step. */
int lg_step_physics(lg_ctx* ctx, const float* actions, void* stream);

/* lg_step for a caller that keeps a rollout storage (rsl_rl's runner loop, on_policy_runner.py:401-445): the post-physics
 * kernel also writes the new observation rows to `next_observations` (num_envs, num_obs; RolloutStorage.observations[t + 1],
 * may be NULL), rewards[k] = rew + gamma * values[k] * time_out (PPO.process_env_step, ppo.py:179-183) and dones[k]
 * (ppo.py:165) -- no copy / transition kernels after the step.  Device pointers. */
int lg_step_transition(lg_ctx* ctx, const float* actions, float* next_observations, const float* values, float gamma,
                       float* rewards, float* dones, void* stream);

/* Main-rollout stepping (envs/batch_rollout/robot_batch_rollout.py): advance only the n listed envs; row k of `actions`
 * (n,dof) belongs to env_ids[k] (device pointers).  rollout_mode = 0: the listed envs take a full LeggedRobot step
 * (RobotBatchRollout.step :535-600 for the main envs); rollout_mode = 1: post_physics_step_rollout semantics (:763-817) —
 * no command resampling / pushes / termination / reset, rewards computed but not added to the episode sums, the other
 * envs are not touched (the reference simulates and then restores them, :687, :1585-1640). */
int lg_step_subset(lg_ctx* ctx, const float* actions, const int32_t* env_ids, int32_t n, int32_t rollout_mode, void* stream);

/* RobotBatchRollout.step_rollout (robot_batch_rollout.py:602-716): one rollout step of the n listed rollout envs with dense row outputs -- the name SURVEY s8(b)
 * gives the entry point; = lg_step_subset_rows(..., rollout_mode = 1, ...). */
int lg_step_rollout(lg_ctx* ctx, const float* actions, const int32_t* env_ids, int32_t n, float* obs_out, float* rew_out, uint8_t* reset_out,
                    uint8_t* time_out_out, void* stream);

/* Multi-stage rewards (legged_robot_rew_mixin.py:15-38: update_reward_scales -> _prepare_reward_function): replace the
 * active reward terms (evaluation order, scales already multiplied by dt, num_terms <= LG_MAX_REWARD_TERMS; HOST
 * pointers) and zero every episode sum, as re-creating `episode_sums` does (:672-674).  Stream-ordered. */
int lg_set_reward_terms(lg_ctx* ctx, int32_t num_terms, const int32_t* term_ids, const float* scales, void* stream);

/* AsyncGaitScheduler term: the three weights of the current reward stage (`get_weight(key, reward_scales_stage)`,
 * anymal_c_batch_rollout.py:212-216) and the foot_z_align constant of the spawn pose (see lg_config.async_*).  Stream-ordered. */
int lg_set_async_gait(lg_ctx* ctx, const float weights[3], float foot_z_align, void* stream);

/* The two halves of lg_step_subset, so that sensor kernels (ray caster, body SDF) can run on the post-physics, pre-reset state
 * in between (RobotBatchRolloutPercept._post_physics_step_callback, robot_batch_rollout_percept.py:301-331). */
int lg_step_subset_physics(lg_ctx* ctx, const float* actions, const int32_t* env_ids, int32_t n, void* stream);
int lg_post_physics_subset(lg_ctx* ctx, const int32_t* env_ids, int32_t n, int32_t rollout_mode, void* stream);

/* _sync_main_to_rollout (:1447-1535): with env i*(1+R) the i-th main env and the next R envs its rollouts, copy root /
 * DOF state, actions, history, base velocities, projected gravity and the feet contact state from every main env to its
 * rollouts; pos_drift > 0 adds U(-drift/2, drift/2) to the copied base positions (:1493-1497). */
int lg_sync_main_to_rollout(lg_ctx* ctx, int32_t rollouts_per_main, float pos_drift, void* stream);

/* rollout_batch (envs/batch_rollout/robot_traj_grad_sampling.py:249-280, the horizon loop every sampling planner drives):
 * _sync_main_to_rollout, then `horizon` times step_rollout with all_us[:, i, :] (row k of the (n, horizon, dof) plan belongs
 * to env_ids[k]; read in place, no per-step copy) while rewards[k, i] receives the reward of that step, then
 * _sync_main_to_rollout again.  Device pointers; 2 + 2 * horizon launches enqueued by one call.  Envs without perception
 * sensors between physics and post-physics only (RobotBatchRollout; the percept env keeps its per-step calls). */
int lg_rollout_batch(lg_ctx* ctx, const float* all_us, int32_t horizon, const int32_t* env_ids, int32_t n, int32_t rollouts_per_main,
                     float pos_drift, float* rewards, void* stream);

/* Pieces of lg_step, exposed because the reference exposes them as overridable methods / gym calls. */
int lg_compute_torques(lg_ctx* ctx, const float* actions, void* stream);  /* -> LG_T_TORQUES (and LSTM state) */
int lg_simulate(lg_ctx* ctx, void* stream);                               /* one dt with LG_T_TORQUES applied */
int lg_post_physics_step(lg_ctx* ctx, void* stream);

/* Reset the listed envs (device pointer to n int32 ids).  `update_curriculum` = the reference's init_done. */
int lg_reset_idx(lg_ctx* ctx, const int32_t* env_ids, int32_t n, int32_t update_curriculum, void* stream);

/* gym.set_actor_root_state_tensor_indexed / set_dof_state_tensor_indexed (legged_robot.py:463-465, 487-489, 496): teleport the n
 * listed envs.  `root_states` (N,13) and `dof_state` (N,12,2) are FULL tensors (device pointers, either may be NULL), of which the
 * rows of env_ids (device pointer to n int32) are copied into the simulation state, exactly the calling convention of the gym
 * functions.  The library's own LG_T_ROOT_STATES / LG_T_DOF_STATE are the simulation state (edits through their zero-copy views
 * need no call at all: passing them here is accepted and copies nothing); the rigid-body states of the listed envs are refreshed
 * from the new pose either way and their contact forces zeroed. */
int lg_set_state_indexed(lg_ctx* ctx, const float* root_states, const float* dof_state, const int32_t* env_ids, int32_t n, void* stream);

/* Dense copies of the listed envs' rows of obs_buf (n, num_obs), rew_buf (n), reset_buf (n) u8 and time_out_buf (n) u8 -- what `step()` /
 * `step_rollout()` of the main-rollout env return (`obs_buf[main_env_indices]` ..., robot_batch_rollout.py:598-600, 714-716) -- in one launch.
 * Any output may be null.  Device pointers. */
int lg_gather_step_rows(lg_ctx* ctx, const int32_t* env_ids, int32_t n, float* obs_out, float* rew_out, uint8_t* reset_out,
                        uint8_t* time_out_out, void* stream);
/* lg_step_subset followed by lg_gather_step_rows of the same envs -- for a rollout step of a context that takes the one-launch path the rows leave
 * the tail of that launch (no second kernel).  All four outputs required. */
int lg_step_subset_rows(lg_ctx* ctx, const float* actions, const int32_t* env_ids, int32_t n, int32_t rollout_mode, float* obs_out, float* rew_out,
                        uint8_t* reset_out, uint8_t* time_out_out, void* stream);

/* Bind a (N) u8 device buffer of per-env flags that check_termination ORs into the contact terminations of the next post-physics steps (an env class's
 * own reset rule on top of LeggedRobot.check_termination: FootTrackElSpider, elspider.py:598-603, resets when the base strays from its planner); NULL
 * unbinds.  The caller fills it between lg_step_physics and lg_post_physics_step. */
int lg_set_extra_termination(lg_ctx* ctx, const uint8_t* dptr);

/* Bind the (N, num_extra_obs) f32 device buffer whose rows are appended to the observation (legged_robot_raycast.py:252-254). */
int lg_set_extra_obs(lg_ctx* ctx, const float* dptr);

/* ---- triangle-mesh queries (replace NVIDIA Warp: utils/ray_caster.py:39-167, utils/mesh_sdf.py:32-116, utils/depth_camera.py) ---- */
typedef struct lg_mesh lg_mesh;

/* Build a BVH over (vertices (n,3) f32, triangles (m,3) i32) — HOST pointers — and upload it to `device_id`
 * (wp.Mesh(points, indices), ray_caster.py:23-42). */
lg_mesh* lg_mesh_create(const float* vertices, int64_t n_vertices, const int32_t* triangles, int64_t n_triangles, int device_id);
void lg_mesh_destroy(lg_mesh* mesh);
int lg_mesh_info(lg_mesh* mesh, int64_t out[2]);            /* {#triangles, #bvh nodes} */
/* {nx, ny} cells of the ray lattice, {0, 0} when the mesh has none.  A mesh whose vertices sit on a rectilinear lattice in x and y (any
 * heightfield-derived mesh) gets, next to the BVH, a per-cell triangle table that rays walk instead of the tree: same hits, same t
 * (environment LG_RAY_GRID=0 at creation time: never).  Warp has one structure for every mesh (ray_caster.py:39-42); this is an accelerator
 * behind the same calls, not a different query. */
int lg_mesh_ray_lattice(lg_mesh* mesh, int32_t out[2]);
/* {nx, ny} cells of the contact-query table, {0, 0} when the mesh has none.  A lattice mesh whose lines are evenly spaced (what the heightfield and
 * confined-terrain converters write) also gets, per cell, its faces sorted by height in two groups; the physics kernel's closest-point contact queries
 * (legged_robot.py:87-153 on `mesh_type = 'trimesh'` with `use_terrain_obj`) index those cells instead of walking the tree -- same contacts
 * (environment LG_LATTICE_CP=0 at creation time: never). */
int lg_mesh_contact_lattice(lg_mesh* mesh, int32_t out[2]);
const char* lg_mesh_last_error(lg_mesh* mesh);

/* raycast_mesh (ray_caster.py:95-167): device pointers, n rays; hits (n,3) = o + t d or o + d max_dist, found (n) u8. */
int lg_raycast_mesh(lg_mesh* mesh, const float* origins, const float* dirs, int64_t n_rays, float max_dist,
                    float* hits, uint8_t* found, void* stream);

/* MeshSDF.query (mesh_sdf.py:230-336, kernel :38-116): signed distance (n) and unit gradient (n,3); max_dist / 0 when out of range. */
int lg_mesh_query_sdf(lg_mesh* mesh, const float* points, int64_t n_points, float max_dist, float* sdf, float* grad, void* stream);

/* RayCaster._update_ray_casting + LeggedRobotRayCast._get_raycast_distances (ray_caster.py:558-594,
 * legged_robot_raycast.py:262-297) for all envs: rays = pattern rotated by the base (yaw-only or full) + base position. */
int lg_raycaster_update(lg_mesh* mesh, const float* root_states, const float* ray_origins, const float* ray_dirs,
                        int32_t num_envs, int32_t num_rays, float max_dist, int32_t attach_yaw_only,
                        float* ray_hits, uint8_t* hits_found, float* raycast_distances, void* stream);

/* The same for the n listed envs only (env_ids: device pointer, NULL = envs 0..n-1; RayCaster.update(env_ids=...),
 * ray_caster.py:518-556): outputs stay indexed by env id; row e of the distance observation starts at
 * raycast_distances + e * distance_stride (>= num_rays), so it can sit inside a wider extra-observation row. */
int lg_raycaster_update_subset(lg_mesh* mesh, const float* root_states, const float* ray_origins, const float* ray_dirs,
                               int32_t num_rays, float max_dist, int32_t attach_yaw_only, const int32_t* env_ids, int32_t n,
                               float* ray_hits, uint8_t* hits_found, float* raycast_distances, int32_t distance_stride, void* stream);

/* RobotBatchRolloutPercept._update_sdf_values (robot_batch_rollout_percept.py:384-440): for the n listed envs (NULL = all)
 * and the num_query_bodies bodies body_indices[] (device), query point = body position + body rotation * sphere_offsets[b]
 * (device (nq,3) or NULL) read from rigid_body_state (N, num_bodies, 13); writes sdf_values[e * sdf_stride + b] and, when not
 * NULL, sdf_gradients (N,nq,3) and nearest_points (N,nq,3) = p - sdf * grad.  One launch for all bodies. */
int lg_sdf_bodies_update(lg_mesh* mesh, const float* rigid_body_state, int32_t num_bodies, const int32_t* body_indices,
                         const float* sphere_offsets, int32_t num_query_bodies, const int32_t* env_ids, int32_t n, float max_dist,
                         float* sdf_values, int32_t sdf_stride, float* sdf_gradients, float* nearest_points, void* stream);

typedef struct lg_depth_params {
  int32_t width, height;                 /* cfg.depth.original */
  int32_t resized_width, resized_height; /* cfg.depth.resized */
  int32_t buffer_len;
  float near_clip, far_clip;
  float position[3];                     /* camera mount in the base frame */
  float quat_offset[4];                  /* mount rotation exactly as the reference hands it to quat_mul (depth_camera.py:546-562) */
} lg_depth_params;

/* DepthCameraWarp.update + update_depth_buffer (depth_camera.py:402-566): camera pose from the base pose, one ray per
 * pixel, depth = -distance (or -far_clip), + per-env noise, clip, bicubic resize, normalise to [-0.5, 0.5], FIFO of
 * buffer_len frames (all frames = newest when episode_length_buf <= 1).  env_noise may be NULL. */
int lg_depth_camera_update(lg_mesh* mesh, const lg_depth_params* params, const float* root_states, const float* ray_dirs,
                           const int64_t* episode_length_buf, int32_t num_envs, const float* env_noise,
                           float* camera_pos, float* camera_rot, float* depth_buffer, void* stream);

/* ---- PoseAnymal / PoseGo2 (reference envs/anymal_c/anymal.py:146-250, envs/go2/go2.py:146-246): what those classes add to a step, behind
 * lg_step on the same stream.  In the reference's order: envs whose episode length hits a multiple of the resampling period draw the four pose
 * channels (yaw / pitch / roll shift, base height) from `ranges` with the uniforms u[:, 0:4]; `_reward_orientation` against the commanded pitch /
 * roll and `_reward_base_height` against the commanded height are added to the native reward (which ran without these two terms and without the
 * positivity clip) before the clip, the termination term after it (legged_robot.py:226-232); the two terms' episode sums (2, n) accumulate, the
 * sums of the envs reset in this step go to `extras` as means over those envs / max_episode_length_s and are cleared; reset envs draw again
 * (u[:, 4:8]); observation rows = native[0:12] | pose channels | native[12:], + (2 noise_u - 1) * noise_scale_vec when noise_u is given, clipped.
 * pose_cmd: rows of `cmd_stride` floats (columns 4..7 of the class's (n, 8) commands tensor); base_z: root height before the reset, one value per
 * `base_z_stride` floats (column 2 of the base row of rigid_body_state); measured_heights (n, num_heights) or NULL; acc: 3 doubles, zero before the
 * first call, owned by the caller (left zero by every call).  Device pointers; asynchronous on `stream`. */
typedef struct lg_pose_params {
  float ranges[4][2];                    /* commands.ranges: base_yaw_shift, base_pitch_shift, base_roll_shift, base_height */
  int32_t resampling_steps;              /* int(commands.resampling_time / dt) */
  float scale_orientation, scale_base_height, scale_termination;   /* rewards.scales x dt */
  int32_t only_positive_rewards;
  float max_episode_length_s, clip_observations;
  int32_t num_heights;                   /* height-scan points in the native row (0 without measure_heights) */
  int32_t num_proprio;                   /* proprioceptive entries of the native row: 12 + 9 x legs (48 | 66; 0 = 48) */
} lg_pose_params;
int lg_pose_layer_step(const lg_pose_params* params, int32_t n, float* pose_cmd, int32_t cmd_stride, float* sums, float* extras,
                       const float* nat_obs, const float* nat_rew, const uint8_t* reset, const uint8_t* time_out, const int64_t* eplen_before,
                       const float* base_z, int32_t base_z_stride, const float* projected_gravity, const float* measured_heights,
                       const float* u, const float* noise_u, const float* noise_scale_vec, float* obs_out, float* rew_out, double* acc,
                       void* stream);

/* What FootTrackElSpider adds to an env step (reference envs/elspider_air/elspider.py:547-676; the foothold planner of type 1, utils/raibert_planner.py:304-497,
 * with its two random walks, utils/math_utils.py:217-288), on the device: csrc/lg_foottrack.hip.  The planner's state lives in caller-owned device arrays
 * (lg_foottrack_state), one row per env, updated in place.
 *   lg_foottrack_stray       between lg_step_physics and lg_post_physics_step: |root position - planner base position| > 0.5 m -> the per-env flags that
 *                            lg_set_extra_termination has bound (check_termination, :583-588);
 *   lg_foottrack_layer_step  behind lg_post_physics_step: the five `_reward_raibert_*` terms (:660-674) on the pre-reset pose added to the native reward
 *                            before the positivity clip, their episode sums (sums: (5, n)) and `extras` means (acc: six doubles of scratch, zero before the
 *                            first call), the planner re-anchored at reset envs (:590-592), the 94-entry observation row (:561-581; noise_u: (n, 94) uniforms,
 *                            noise_scale_vec: 94 scales), then the planner's step with the commands (:594-597).  u_base (n, 6) uniforms and n_foot (n, 18)
 *                            standard normals: the targets a walk redraws in this step.  planner_stepped = 0 until the planner has taken its first step
 *                            (all six feet count as supporting before). */
typedef struct lg_foottrack_params {
  float dt, gait_period, swing_ema, reward_sigma, swing_height;
  float phase_offsets[6];
  float base_bounds[2][6];               /* uniform walk [min | max] of (x shift, y shift, height, yaw, pitch, roll) */
  float foot_mean[18], foot_sigma[18];   /* normal walk of the six nominal footholds in the base frame */
  float base_interval, base_max_vel, foot_interval, foot_max_vel;
  float scales[5];                       /* x dt: raibert_base_pos_track, raibert_base_quat_track, raibert_foot_pos_track, raibert_foot_pos_track_z, raibert_foot_swing_contact */
  float scale_termination;
  int32_t only_positive_rewards;
  float max_episode_length_s, clip_observations;
  int32_t num_bodies, feet_indices[6];
  int32_t add_noise;
} lg_foottrack_params;
typedef struct lg_foottrack_state {
  float *base_pos, *base_quat, *base_pos_shift, *base_quat_shift;   /* (n, 3) (n, 4) (n, 3) (n, 4) */
  float *base_x_world, *base_y_world;                                /* (n, 3): the planner's heading axes as its last step left them */
  float *foot_pos, *gait_idx, *gait_phases;                          /* (n, 6, 3) (n) (n, 6) */
  uint8_t* last_contacts;                                            /* (n, 6) */
  float *bw_cur, *bw_tgt, *bw_timer;                                 /* base-pose walk: (n, 6) (n, 6) (n) */
  float *fw_cur, *fw_tgt, *fw_timer;                                 /* foothold walk: (n, 18) (n, 18) (n) */
} lg_foottrack_state;
int lg_foottrack_stray(int32_t n, const float* root_states, const float* planner_base_pos, float* pos_diff, uint8_t* stray, void* stream);
int lg_foottrack_layer_step(const lg_foottrack_params* params, const lg_foottrack_state* state, int32_t n, int32_t planner_stepped, const float* nat_obs,
                            const float* nat_rew, const uint8_t* reset, const uint8_t* time_out, const float* rigid_body_state, const float* contact_forces,
                            const float* root_states, const float* commands, int32_t cmd_stride, const float* u_base, const float* n_foot, const float* noise_u,
                            const float* noise_scale_vec, float* obs_out, float* rew_out, float* sums, float* extras, double* acc, void* stream);

/* Per-kernel timing with HIP events recorded on the caller's stream around the kernels of lg_step.
 * lg_profile_begin arms up to `max_samples` instrumented steps (every `stride`-th lg_step call is sampled);
 * lg_profile_end synchronises the events and returns the mean duration in ms of {physics, post-physics, finalize}
 * and the number of sampled steps.  Used by bench.py for the roofline figure. */
int lg_profile_begin(lg_ctx* ctx, int32_t max_samples, int32_t stride);
int lg_profile_end(lg_ctx* ctx, float mean_ms[3], int32_t* nsamples);

const char* lg_last_error(lg_ctx* ctx);
void lg_destroy(lg_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* LGSTEP_H */
