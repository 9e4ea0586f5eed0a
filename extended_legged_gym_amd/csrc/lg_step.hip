// lg_step.hip — gfx950 (MI355X) kernels and C-ABI implementation of the legged-robot environment step.
// ABI: include/lgstep.h.  Reference behaviour: legged_gym/envs/base/legged_robot.py ("LR"),
// legged_robot_rew_mixin.py ("RM"), anymal_c/anymal.py, utils/gait_scheduler.py, utils/math_utils.py.
//
// Kernels
//   physics_kernel   one DPP quad (4 lanes) per env, one leg per lane, 16 envs per wave64; clip actions, then
//                    `decimation` x (actuator torques [PD | LSTM] + articulated dynamics + contact) entirely in
//                    registers/LDS; state is read once and written once per policy step.
//   post_kernel      4 envs per 256-thread workgroup.  Wide stages one wave per env: rows staged in LDS with one round of
//                    loads, terrain height scan, observation rows with noise.  Narrow stages (the post-physics logic in
//                    lane-parallel stages that keep the reference's order of side effects) for all four envs on one
//                    wave, sixteen lanes per env; the last workgroup to arrive
//                    (sharded device-scope arrival counters) publishes the episode statistics of the step, summed
//                    with integer atomics (deterministic).
//   finalize_kernel  one workgroup: the same statistics step behind lg_reset_idx.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <algorithm>
#include <vector>

#include "lg_device.h"
#include "lg_bvh.h"
#define LG_INSTANCE_TU
#include "lg_instance.h"          // LG_LEGS -> this instance's namespace and the names of its entry points (lg4_* / lg6_*)

namespace LG_NS {
#include "lg_physics.h"
#if NJ != 3
#include "lg_chain.h"             // the instances with longer legs (2 x 6): generic-NJ physics, one wave per workgroup
#endif

#define LG_MESH_OOB_MARGIN 1.0f   // metres beyond the collision mesh's bounding box before an env counts as lost
#define EPB EPW         // envs per workgroup (= per main wave) in physics_kernel: 16 quads, or 8 groups of eight lanes
#define LPE (NDOF <= 16 ? 16 : 32)   // lanes per env in the narrow stages of post_kernel: one lane per DOF (12 of 16, 18 of 32)
#define EPBP (64 / LPE) // envs per 256-thread workgroup in post_kernel (a wave per env in the wide stages, LPE lanes per env in the narrow ones); quadrupeds: 4 workgroups per CU at N = 4096
#define MAX_P 192       // height-scan points per env held in LDS
#define PART_STRIDE (LG_MAX_REWARD_TERMS + 3)
#define NBODY_MAX (1 + (NJ + 1) * NLEG)         // rigid bodies of this instance's robots: base + legs x (NJ links + FOOT)
#define NPROP LG_NUM_PROPRIO_OF(NDOF)           // observation entries in front of the height scan (48 / 66)
#define LG_RS_ROOT_XY LG_RS_ROOT_XY_OF(NDOF)
#define LG_RS_ROOT_VEL LG_RS_ROOT_VEL_OF(NDOF)
#define LG_RS_CMD_RESET LG_RS_CMD_RESET_OF(NDOF)
#define LG_RS_NOISE LG_RS_NOISE_OF(NDOF)

// The scalars the post-physics tail of the fused step reads, packed into one contiguous block (built on the host, copied into
// LDS at kernel start): lg_config is 8 KB (the actuator weights sit in the middle of it) and the tail touches ~20 different
// 64-byte lines of it, each a scalar-cache miss on first touch, serialised by the tail's dependent chain.
enum { HC_DT = 0, HC_K, HC_KFAT, HC_KTERM, HC_TERM_SCALE, HC_TERM_MASK, HC_RESAMPLING_STEPS, HC_HEADING, HC_PUSH, HC_PUSH_INTERVAL, HC_MAX_PUSH,
       HC_FLIP, HC_MAX_EPLEN, HC_ONLY_POS, HC_STAND, HC_CURRICULUM, HC_GAIT_ON, HC_GAIT_PERIOD, HC_GAIT_SWING, HC_GAIT_PHASE /* NLEG */, HC_SIGMA = HC_GAIT_PHASE + NLEG,
       HC_BH_TARGET, HC_MAX_CF, HC_MEASURE_H, HC_P, HC_FEET /* NLEG */, HC_NPEN = HC_FEET + NLEG, HC_PEN /* NBODY_MAX */, HC_NTERM = HC_PEN + NBODY_MAX,
       HC_TERMB /* NBODY_MAX */, HC_SOFT_VEL = HC_TERMB + NBODY_MAX, HC_SOFT_TQ, HC_NUM_OBS, HC_ADD_NOISE, HC_INJECT, HC_OS_LIN, HC_OS_ANG, HC_OS_POS,
       HC_OS_VEL, HC_OS_H, HC_CLIP_OBS, HC_SEED_LO, HC_SEED_HI, HC_NUM_EXTRA, HC_IDS /* 32 */, HC_SCALES = HC_IDS + LG_MAX_REWARD_TERMS /* 32 */,
       HC_DEFAULT_POS = HC_SCALES + LG_MAX_REWARD_TERMS /* NDOF */, HC_COUNT = HC_DEFAULT_POS + NDOF };

// Device-side view of every pointer member: the global address space.  A pointer read out of a struct is generic to the compiler and
// every access through it a flat_load / flat_store, which counts on BOTH wait counters and returns out of order, so that any wait for
// LDS data with one of them in flight becomes a full drain; with the members typed as global pointers (device compilation only: the
// layout is the same) the accesses are global_load / global_store (LG_G, lg_device.h).
struct DevCtx {
  lg_config cfg;
  lg_robot_model model;
  int N, B, K, P, per_leg;
  TerrainView ter;
  float terrain_mu, env_length; int num_levels, num_types;
  // tensors (device pointers into the arena)
  float LG_G *root, *dof, *rigid, *cforce, *torques, *actions, *last_actions, *last_dof_vel, *last_root_vel, *commands;
  float LG_G *base_lin_vel, *base_ang_vel, *proj_grav, *base_lin_acc, *base_ang_acc, *feet_air, *feet_ctime;
  uint8_t LG_G* last_contacts;
  float LG_G *heights, *obs, *rew;
  uint8_t LG_G *reset_buf, *time_out;
  int64_t LG_G* ep_len;
  float LG_G* ep_sums;
  int64_t LG_G *levels, *types;
  float LG_G *origins, *friction, *mass_added, *sea_h, *sea_c, *gait_idx, *gait_foot_z, *extras, *rand_inject;
  int64_t LG_G* counters;
  float LG_G* cmd_ranges;   // LG_T_COMMAND_RANGES (4,2)
  double LG_G* ep_stats;
  const float LG_G* terrain_origins;
  const float LG_G *noise_vec, *height_points;
  const float LG_G* extra_obs;   // (N, cfg.num_extra_obs) caller-owned rows appended to the observation
  const uint8_t LG_G* extra_term;   // (N) caller-owned flags ORed into the contact terminations (lg_set_extra_termination), or null
  const float4 LG_G* obs_tab;    // fused tail: per observation entry (source code, scale, offset, noise scale), packed on the host (pack_obs_table)
  float LG_G* partials;     // [nblocks][PART_STRIDE] : per-workgroup sums of episode_sums over reset envs, #reset, sum of levels, sum of finished lengths
  float LG_G* lvl_part;     // [nblocks] : per-workgroup sum of terrain levels
  unsigned LG_G* part_flag; // [nblocks] : 1 when some env of the workgroup was reset in this step (its partials row is valid)
  long long LG_G* acc;      // [PART_STRIDE] fixed-point (x 2^24) sums of the reset envs' rows of one post-kernel launch
  unsigned LG_G* tickets;   // 9 counters, one per 128-B line: per-shard arrivals of the post kernel's workgroups + the shards' own
  float mesh_lo[3], mesh_hi[3];   // bounding box of the collision mesh (LG_MESH_TRIMESH): a base that leaves it by more than LG_MESH_OOB_MARGIN ends the episode
  float LG_G* mesh_cache;   // [N][NLEG][LG_MAX_CP][4]: last closest-point query of every collision sphere (mesh terrains)
  float lstm_w[912];   // actuator network weights, gate-interleaved (pack_lstm_weights): read with scalar loads
  int nblocks_post;
  // reward-term bookkeeping of the post kernel, derived from cfg.reward_term_ids on the host (reward_meta): a walk over the
  // term list in the kernel is one dependent scalar load per term on the narrow-stage chain
  unsigned rew_term_mask; int rew_kfat, rew_kterm; float rew_term_scale;
  float hot[HC_COUNT];          // see the HC_* enum (ints stored as bit patterns)
  float lmod[LM_FIELDS * GRP];  // per-leg model table, packed on the host (pack_leg_model)
  unsigned slide_mask;          // bit sl: some leg's collision sphere in slot sl stands for a capsule part (lg_robot_model.cp_slide)
  unsigned slot_perm;           // nibble p: the slot at position p of the contact-detection deal (main wave 0-2, wave 1: 3, wave 2: 4-5, wave 3: 6-7), see lg_create
  float mesh_reach;             // PhysParams::cache_reach (LG_MESH_REACH overrides LG_MESH_CACHE_REACH: A/B)
  unsigned mesh_perm;           // the same for triangle-mesh terrains, where every wave takes a PAIR of positions (MESH_PAIR0): which spheres share a wave
  int n_sc; unsigned sc_pairs[LG_MAX_SC_PAIRS];   // self-collision candidates, packed leg a | slot a << 8 | leg b << 16 | slot b << 24 (0 pairs unless lg_config.self_collisions)
  uint4 sc_tab[LG_MAX_SC_PAIRS];                  // ... as the pair filter reads them (sc_prefilter): {slot-record indices (slot * 64 + leg) a | b << 16, radius a, radius b, (ra + rb + contact_offset)^2 (1 + 1e-4)}
  int n_stepped;                // envs advanced by the last step launch (N, or the subset size)
  float lvl_total_before;       // subset steps with a terrain curriculum: sum of ALL terrain levels before the launch (level_total_kernel)
  unsigned long long LG_G* stamps;   // 16 counters, written only by the -DLG_STAMPS diagnostic build
};

struct TensorInfo { size_t off; int64_t shape[4]; int ndim; int dtype; };

struct lg_ctx {
  int32_t legs = NLEG; // first member: how the library's entry points (lg_dispatch.cpp) find the instance a context belongs to
  DevCtx h;            // host copy
  DevCtx* d = nullptr; // device copy
  void* arena = nullptr; bool own_arena = false; size_t arena_bytes = 0;
  void* aux = nullptr; // noise_vec, height_points, partials
  void* obs_tab = nullptr;
  void* mesh_cache = nullptr;
  void* grid_verts = nullptr;   // device copy of lg_terrain.grid_vertices
  void* hmin = nullptr;         // TerrainView::Hmin
  int grid_mesh = 1;           // LG_GRID_MESH=0: walk the BVH for grid meshes too (diagnostic / A-B)
  TensorInfo t[LG_T_COUNT];
  int device = 0;
  unsigned long sync_calls = 0;
  int split = 1;       // fused step: run the LSTM actuators on three extra waves (LG_SPLIT=0 disables, diagnostic)
  int fuse = 1;        // lg_step ends inside the physics kernel (LG_FUSE=0: separate post kernel, diagnostic / A-B)
  int persist = 1;     // lg_rollout_batch is one launch per horizon (LG_PERSIST=0: one launch per step, the checker of that path)
  int gfuse = 0;       // LG_GFUSE=1: the six-legged instance's lg_step in one launch (can_gfuse)
  int spec = 1;        // A/B build 13 only: TGS + pyramid steps run a compile-time instance of that solver.  Measured (one session, three
                       // rounds each): 0.0792 ms per step against 0.0782 for the generic instance -- 20 instructions fewer per relaxation, a
                       // different schedule of the same dependent chain, 1.2 % slower; not instantiated in the product library
  std::string err;
  // optional per-kernel timing (lg_profile_begin / lg_profile_end)
  std::vector<hipEvent_t> ev; int prof_max = 0, prof_stride = 1, prof_n = 0; long prof_calls = 0;
};

static_assert(offsetof(lg_ctx, legs) == 0, "lg_dispatch.cpp finds a context's kernel instance in its first four bytes");
static thread_local std::string g_err;

static void hot_config(DevCtx& h) {
  const lg_config& g = h.cfg; const lg_robot_model& m = h.model;
  pack_leg_model(h.lmod, &m, &g);
  auto I = [&](int i, int v) { memcpy(&h.hot[i], &v, 4); };
  auto U = [&](int i, unsigned v) { memcpy(&h.hot[i], &v, 4); };
  auto F = [&](int i, float v) { h.hot[i] = v; };
  F(HC_DT, g.sim_dt * g.decimation); I(HC_K, g.num_reward_terms); I(HC_KFAT, h.rew_kfat); I(HC_KTERM, h.rew_kterm); F(HC_TERM_SCALE, h.rew_term_scale);
  U(HC_TERM_MASK, h.rew_term_mask); I(HC_RESAMPLING_STEPS, g.resampling_steps); I(HC_HEADING, g.heading_command); I(HC_PUSH, g.push_robots);
  I(HC_PUSH_INTERVAL, g.push_interval); F(HC_MAX_PUSH, g.max_push_vel_xy); I(HC_FLIP, g.terminate_on_flip); F(HC_MAX_EPLEN, g.max_episode_length);
  I(HC_ONLY_POS, g.only_positive_rewards); I(HC_STAND, g.reward_class == LG_RC_STAND); I(HC_CURRICULUM, g.curriculum); I(HC_GAIT_ON, g.gait_enabled); F(HC_GAIT_PERIOD, g.gait_period);
  F(HC_GAIT_SWING, g.gait_swing_height);
  for (int f = 0; f < NLEG; ++f) { F(HC_GAIT_PHASE + f, g.gait_foot_phases[f]); I(HC_FEET + f, m.feet_indices[f]); }
  F(HC_SIGMA, g.tracking_sigma); F(HC_BH_TARGET, g.base_height_target); F(HC_MAX_CF, g.max_contact_force); I(HC_MEASURE_H, g.measure_heights);
  I(HC_P, g.measure_heights ? h.P : 0);
  I(HC_NPEN, m.num_penalised); I(HC_NTERM, m.num_termination);
  for (int i = 0; i < NBODY_MAX; ++i) { I(HC_PEN + i, i < m.num_penalised ? m.penalised_contact_indices[i] : 0); I(HC_TERMB + i, i < m.num_termination ? m.termination_contact_indices[i] : 0); }
  F(HC_SOFT_VEL, g.soft_dof_vel_limit); F(HC_SOFT_TQ, g.soft_torque_limit); I(HC_NUM_OBS, g.num_obs); I(HC_ADD_NOISE, g.add_noise);
  I(HC_INJECT, g.rng_mode == LG_RNG_INJECT); F(HC_OS_LIN, g.obs_scale_lin_vel); F(HC_OS_ANG, g.obs_scale_ang_vel); F(HC_OS_POS, g.obs_scale_dof_pos);
  F(HC_OS_VEL, g.obs_scale_dof_vel); F(HC_OS_H, g.obs_scale_height); F(HC_CLIP_OBS, g.clip_observations);
  U(HC_SEED_LO, (unsigned)g.seed); U(HC_SEED_HI, (unsigned)(g.seed >> 32)); I(HC_NUM_EXTRA, g.num_extra_obs);
  for (int k = 0; k < LG_MAX_REWARD_TERMS; ++k) { I(HC_IDS + k, g.reward_term_ids[k]); F(HC_SCALES + k, g.reward_scales[k]); }
  for (int d = 0; d < NDOF; ++d) F(HC_DEFAULT_POS + d, g.default_dof_pos[d]);
}

static void reward_meta(DevCtx& h) {
  const lg_config& g = h.cfg;
  h.rew_term_mask = 0; h.rew_kfat = g.num_reward_terms; h.rew_kterm = -1; h.rew_term_scale = 0.f;
  for (int k = 0; k < g.num_reward_terms; ++k) {
    const int id = g.reward_term_ids[k];
    h.rew_term_mask |= 1u << id;
    if (id == LG_REW_FEET_AIR_TIME) h.rew_kfat = k;       // position of feet_air_time in the evaluation order (K = absent)
    if (id == LG_REW_TERMINATION) { h.rew_kterm = k; h.rew_term_scale = g.reward_scales[k]; }   // added after the clip, LR:226-232
  }
}

// ============================================================================================ device: RNG
LG_DEV float uniform_draw(const DevCtx* __restrict__ C, int e, int slot, int64_t step, uint32_t stream) {
  if (C->cfg.rng_mode == LG_RNG_INJECT) return C->rand_inject[(size_t)e * (LG_RS_NOISE + C->cfg.num_obs) + slot];
  uint32_t o[4];
  philox4((uint32_t)e, (uint32_t)step, (uint32_t)(slot >> 2), stream, (uint32_t)C->cfg.seed, (uint32_t)(C->cfg.seed >> 32), o);
  uint32_t x = (slot & 3) == 0 ? o[0] : ((slot & 3) == 1 ? o[1] : ((slot & 3) == 2 ? o[2] : o[3]));
  return u01(x);
}
LG_DEV float rand_float(float lo, float hi, float u) { return (hi - lo) * u + lo; }
// the four uniforms of slot group `grp` (slots 4*grp .. 4*grp+3) with ONE Philox call
LG_DEV void uniform_draw4(const DevCtx* __restrict__ C, int e, int grp, int64_t step, uint32_t stream, float u[4]) {
  if (C->cfg.rng_mode == LG_RNG_INJECT) {
    const float* p = C->rand_inject + (size_t)e * (LG_RS_NOISE + C->cfg.num_obs) + 4 * grp;
    u[0] = p[0]; u[1] = p[1]; u[2] = p[2]; u[3] = p[3];
    return;
  }
  uint32_t o[4];
  philox4((uint32_t)e, (uint32_t)step, (uint32_t)grp, stream, (uint32_t)C->cfg.seed, (uint32_t)(C->cfg.seed >> 32), o);
  u[0] = u01(o[0]); u[1] = u01(o[1]); u[2] = u01(o[2]); u[3] = u01(o[3]);
}

// ============================================================================================ device: actuators
struct LegActuator { float h[2][3][8], c[2][3][8]; };   // LSTM state of this lane's three joints

// The ANYdrive LSTM actuator (anymal.py:93-105): 2-layer LSTM(2->8->8) + Linear(8->1), torch gate order i, f, g, o.
//
// The 969 weights are the same for every lane: they stay in global memory in a gate-interleaved order (DevCtx::lstm_w,
// pack_lstm_weights) and are fetched with scalar loads (s_load_dwordx4..x16 through the scalar cache) straight into
// SGPRs; for every unit k and input kk the four gate weights (i, f, g, o) are adjacent, so one SGPR pair feeds a packed
// FMA (v_pk_fma_f32: gates i,f and g,o) — half the instructions of a scalar dot product, no LDS traffic, and the scalar
// unit runs ahead of the vector ALU.  Offsets in floats:
enum { LW_B0 = 0, LW_X0 = 32, LW_H0 = 96, LW_B1 = 352, LW_I1 = 384, LW_H1 = 640, LW_OUT = 896, LW_COUNT = 905 };
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// torch layout (lg_config.actuator_net) -> gate-interleaved layout, once, on the host at lg_create
static void pack_lstm_weights(float* W, const float* net) {
  for (int i = 0; i < LW_COUNT; ++i) W[i] = 0.f;
  for (int i = 0; i < LG_LSTM_NPARAM; ++i) {
    int d = -1;
    if (i < 64) { int r = i >> 1, j = i & 1; d = LW_X0 + ((r & 7) * 2 + j) * 4 + (r >> 3); }
    else if (i < 320) { int r = (i - 64) >> 3, kk = (i - 64) & 7; d = LW_H0 + ((r & 7) * 8 + kk) * 4 + (r >> 3); }
    else if (i < 384) d = -1;                              // biases: below
    else if (i < 640) { int r = (i - 384) >> 3, kk = (i - 384) & 7; d = LW_I1 + ((r & 7) * 8 + kk) * 4 + (r >> 3); }
    else if (i < 896) { int r = (i - 640) >> 3, kk = (i - 640) & 7; d = LW_H1 + ((r & 7) * 8 + kk) * 4 + (r >> 3); }
    else if (i < 960) d = -1;
    else d = LW_OUT + (i - 960);
    if (d >= 0) W[d] = net[i];
  }
  for (int i = 0; i < 64; ++i) {                           // (b_ih + b_hh) of both layers, gate-interleaved
    const int lay = i >> 5, r = i & 31, base = lay ? 896 : 320;
    W[(lay ? LW_B1 : LW_B0) + (r & 7) * 4 + (r >> 3)] = net[base + r] + net[base + 32 + r];
  }
}

#if NJ == 3      // ---- three-joint instances: LSTM actuator waves, the tuned physics kernel (lg_chain.h + physics_kernel_chain below serve NJ = 6)
LG_DEV float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
LG_DEV float fast_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }
LG_DEV v2f splat2(float x) { v2f r = {x, x}; return r; }
// sigmoid(a) * tanh(b) with ONE reciprocal: sign(b) (1 - E) / ((1 + e^-a)(1 + E)), E = e^(-2|b|) in (0, 1].  The gate
// products i * tanh(g) and o * tanh(c') of an LSTM unit cost 3 transcendentals each this way instead of 4 (quarter-rate
// instructions: 16 cycles per wave64), 8 instead of 10 per unit.  No overflow: e^-a = inf gives a zero reciprocal.
LG_DEV float sigmoid_times_tanh(float a, float b) {
  const float E = __expf(-2.f * fabsf(b));
  const float r = __builtin_amdgcn_rcpf((1.f + __expf(-a)) * (1.f + E));
  return copysignf((1.f - E) * r, b);
}

// one joint: inputs x0 (scaled position error), x1 (scaled velocity); state h0, c0, h1, c1 of 8 each, updated in place
LG_DEV float lstm_actuator1(const float* __restrict__ W, float x0, float x1, float* h0, float* c0, float* h1, float* c1,
                            float out_scale) {
  const v4f* B0 = (const v4f*)(W + LW_B0); const v4f* X0 = (const v4f*)(W + LW_X0); const v4f* H0 = (const v4f*)(W + LW_H0);
  const v4f* B1 = (const v4f*)(W + LW_B1); const v4f* I1 = (const v4f*)(W + LW_I1); const v4f* H1 = (const v4f*)(W + LW_H1);
  float hn0[8], hn1[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    v4f w = B0[k];
    v2f a_if = w.xy, a_go = w.zw;
    w = X0[2 * k];     a_if = __builtin_elementwise_fma(w.xy, splat2(x0), a_if); a_go = __builtin_elementwise_fma(w.zw, splat2(x0), a_go);
    w = X0[2 * k + 1]; a_if = __builtin_elementwise_fma(w.xy, splat2(x1), a_if); a_go = __builtin_elementwise_fma(w.zw, splat2(x1), a_go);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      w = H0[8 * k + kk];
      a_if = __builtin_elementwise_fma(w.xy, splat2(h0[kk]), a_if); a_go = __builtin_elementwise_fma(w.zw, splat2(h0[kk]), a_go);
    }
    float cn = fast_sigmoid(a_if.y) * c0[k] + sigmoid_times_tanh(a_if.x, a_go.x);
    c0[k] = cn; hn0[k] = sigmoid_times_tanh(a_go.y, cn);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    v4f w = B1[k];
    v2f a_if = w.xy, a_go = w.zw;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      w = I1[8 * k + kk];
      a_if = __builtin_elementwise_fma(w.xy, splat2(hn0[kk]), a_if); a_go = __builtin_elementwise_fma(w.zw, splat2(hn0[kk]), a_go);
      w = H1[8 * k + kk];
      a_if = __builtin_elementwise_fma(w.xy, splat2(h1[kk]), a_if); a_go = __builtin_elementwise_fma(w.zw, splat2(h1[kk]), a_go);
    }
    float cn = fast_sigmoid(a_if.y) * c1[k] + sigmoid_times_tanh(a_if.x, a_go.x);
    c1[k] = cn; hn1[k] = sigmoid_times_tanh(a_go.y, cn);
  }
  float o = W[LW_OUT + 8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { o += W[LW_OUT + k] * hn1[k]; h0[k] = hn0[k]; h1[k] = hn1[k]; }
  return out_scale * o;       // no torque clip on this path (anymal.py:101-102)
}

// The same network in two halves for the actuator waves of the multi-wave kernel.  The recurrent terms of BOTH layers
// (bias + W_hh h) depend only on the hidden state left by the previous substep, so they are evaluated while the main
// wave runs its Gauss-Seidel sweeps (the actuator waves have nothing else to do then); what stays on the critical path
// in front of the rendezvous is the input part: 2 of 10 columns of layer 0 and 8 of 16 of layer 1, plus the gates.
struct LstmPre { v4f a0[8], a1[8]; };
#ifndef LG_LSTM_LDS
#define LG_LSTM_LDS (LG_AB == 7)     // A/B build 7: the actuator waves read the LSTM weights from LDS instead of through scalar loads
#endif
LG_DEV void lstm_recurrent_part(const float* __restrict__ W, const float* h0, const float* h1, LstmPre& pre) {
  const v4f* B0 = (const v4f*)(W + LW_B0); const v4f* H0 = (const v4f*)(W + LW_H0);
  const v4f* B1 = (const v4f*)(W + LW_B1); const v4f* H1 = (const v4f*)(W + LW_H1);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    v4f w = B0[k];
    v2f a_if = w.xy, a_go = w.zw;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      w = H0[8 * k + kk];
      a_if = __builtin_elementwise_fma(w.xy, splat2(h0[kk]), a_if); a_go = __builtin_elementwise_fma(w.zw, splat2(h0[kk]), a_go);
    }
    pre.a0[k] = (v4f){a_if.x, a_if.y, a_go.x, a_go.y};
    w = B1[k];
    a_if = w.xy; a_go = w.zw;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      w = H1[8 * k + kk];
      a_if = __builtin_elementwise_fma(w.xy, splat2(h1[kk]), a_if); a_go = __builtin_elementwise_fma(w.zw, splat2(h1[kk]), a_go);
    }
    pre.a1[k] = (v4f){a_if.x, a_if.y, a_go.x, a_go.y};
  }
}
LG_DEV float lstm_input_part(const float* __restrict__ W, float x0, float x1, const LstmPre& pre, float* h0, float* c0, float* h1,
                             float* c1, float out_scale) {
  const v4f* X0 = (const v4f*)(W + LW_X0); const v4f* I1 = (const v4f*)(W + LW_I1);
  float hn0[8], hn1[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    v2f a_if = pre.a0[k].xy, a_go = pre.a0[k].zw;
    v4f w = X0[2 * k]; a_if = __builtin_elementwise_fma(w.xy, splat2(x0), a_if); a_go = __builtin_elementwise_fma(w.zw, splat2(x0), a_go);
    w = X0[2 * k + 1]; a_if = __builtin_elementwise_fma(w.xy, splat2(x1), a_if); a_go = __builtin_elementwise_fma(w.zw, splat2(x1), a_go);
    float cn = fast_sigmoid(a_if.y) * c0[k] + sigmoid_times_tanh(a_if.x, a_go.x);
    c0[k] = cn; hn0[k] = sigmoid_times_tanh(a_go.y, cn);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    v2f a_if = pre.a1[k].xy, a_go = pre.a1[k].zw;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      v4f w = I1[8 * k + kk];
      a_if = __builtin_elementwise_fma(w.xy, splat2(hn0[kk]), a_if); a_go = __builtin_elementwise_fma(w.zw, splat2(hn0[kk]), a_go);
    }
    float cn = fast_sigmoid(a_if.y) * c1[k] + sigmoid_times_tanh(a_if.x, a_go.x);
    c1[k] = cn; hn1[k] = sigmoid_times_tanh(a_go.y, cn);
  }
  float o = W[LW_OUT + 8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { o += W[LW_OUT + k] * hn1[k]; h0[k] = hn0[k]; h1[k] = hn1[k]; }
  return out_scale * o;
}

// the three joints of this lane's leg (single-wave builds: lg_compute_torques, LG_SPLIT=0)
LG_DEV void lstm_actuator3(const float* __restrict__ W, const float x0[3], const float x1[3], LegActuator& A, float out_scale,
                           float tau[3]) {
#pragma unroll
  for (int j = 0; j < 3; ++j) tau[j] = lstm_actuator1(W, x0[j], x1[j], A.h[0][j], A.c[0][j], A.h[1][j], A.c[1][j], out_scale);
}

LG_DEV void load_lstm(const DevCtx* __restrict__ C, int e, int l, LegActuator& A) {
  const size_t N12 = (size_t)C->N * NDOF, row = (size_t)e * NDOF + 3 * l;
#pragma unroll
  for (int lay = 0; lay < 2; ++lay)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float4* ph = (const float4*)(C->sea_h + (lay * N12 + row + j) * 8);
      const float4* pc = (const float4*)(C->sea_c + (lay * N12 + row + j) * 8);
      float4 h0 = ph[0], h1 = ph[1], c0 = pc[0], c1 = pc[1];
      A.h[lay][j][0] = h0.x; A.h[lay][j][1] = h0.y; A.h[lay][j][2] = h0.z; A.h[lay][j][3] = h0.w;
      A.h[lay][j][4] = h1.x; A.h[lay][j][5] = h1.y; A.h[lay][j][6] = h1.z; A.h[lay][j][7] = h1.w;
      A.c[lay][j][0] = c0.x; A.c[lay][j][1] = c0.y; A.c[lay][j][2] = c0.z; A.c[lay][j][3] = c0.w;
      A.c[lay][j][4] = c1.x; A.c[lay][j][5] = c1.y; A.c[lay][j][6] = c1.z; A.c[lay][j][7] = c1.w;
    }
}
LG_DEV void store_lstm(const DevCtx* __restrict__ C, int e, int l, const LegActuator& A) {
  const size_t N12 = (size_t)C->N * NDOF, row = (size_t)e * NDOF + 3 * l;
#pragma unroll
  for (int lay = 0; lay < 2; ++lay)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float4* ph = (float4*)(C->sea_h + (lay * N12 + row + j) * 8);
      float4* pc = (float4*)(C->sea_c + (lay * N12 + row + j) * 8);
      ph[0] = make_float4(A.h[lay][j][0], A.h[lay][j][1], A.h[lay][j][2], A.h[lay][j][3]);
      ph[1] = make_float4(A.h[lay][j][4], A.h[lay][j][5], A.h[lay][j][6], A.h[lay][j][7]);
      pc[0] = make_float4(A.c[lay][j][0], A.c[lay][j][1], A.c[lay][j][2], A.c[lay][j][3]);
      pc[1] = make_float4(A.c[lay][j][4], A.c[lay][j][5], A.c[lay][j][6], A.c[lay][j][7]);
    }
}

// torques of this lane's three joints (LR:425-448 / anymal.py:93-105).  ALLOW_NET = false: the caller's launch evaluates the actuator
// network on its helper waves; only the PD / velocity / torque modes are compiled in
template <bool ALLOW_NET = true>
LG_DEV void leg_torques(const DevCtx* __restrict__ C, const LegModel& lm_, const float* __restrict__ Wlds, const float act[3],
                        const float q[3], const float qd[3], const float last_qd[3], LegActuator& A, float tau[3]) {
  const lg_config& g = C->cfg;
  if (ALLOW_NET && g.control_type == LG_CTRL_ACTUATOR_NET) {
    float x0[3], x1[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      x0[j] = (act[j] * g.action_scale + lm_.f(LM_DEFAULT_POS + j) - q[j]) * g.actuator_in_scale[0];
      x1[j] = qd[j] * g.actuator_in_scale[1];
    }
    lstm_actuator3(Wlds, x0, x1, A, g.actuator_out_scale, tau);
    return;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float a = act[j] * g.action_scale, kp = lm_.f(LM_PGAIN + j), kd = lm_.f(LM_DGAIN + j);
    float t;
    if (g.control_type == LG_CTRL_P) t = kp * (a + lm_.f(LM_DEFAULT_POS + j) - q[j]) - kd * qd[j];
    else if (g.control_type == LG_CTRL_V) t = kp * (a - qd[j]) - kd * (qd[j] - last_qd[j]) / g.sim_dt;
    else t = a;
    const float lim = lm_.f(LM_TORQUE_LIMIT + j);
    tau[j] = fminf(fmaxf(t, -lim), lim);
  }
}

// Body states of one leg (+ the base from leg 0) from the generalised state: forward kinematics once more, then the
// (N, B, 13) rows [pos3, quat xyzw4, lin vel3, ang vel3] the reference reads from refresh_rigid_body_state_tensor.
// `part` selects what this caller stores: -1 everything; 0 / 1 / 2 = link 0 (+ base) / link 1 / link 2 (+ foot body).
// per_leg_ >= 0: the caller has read C->per_leg, C->B, C->rigid already
LG_DEV void write_rigid_body_state(const DevCtx* __restrict__ C, const LegModel& lm_, int e, int l, const float* root, const float* q,
                                   const float* qd, int part = -1, float* foot_row_lds = nullptr, float* foot_z_dst = nullptr,
                                   int per_leg_ = -1, int B_ = 0, float LG_G* rigid_ = nullptr) {
  const int per_leg = per_leg_ >= 0 ? per_leg_ : C->per_leg, B = per_leg_ >= 0 ? B_ : C->B;
  const M3 Rb = quat_to_mat(root + 3);
  const V3 pb = v3(root[0], root[1], root[2]), vb = v3(root[7], root[8], root[9]), wb = v3(root[10], root[11], root[12]);
  LegKin k;
  leg_kinematics(lm_, Rb, pb, vb, wb, q, qd, k);
  float* rb = (per_leg_ >= 0 ? rigid_ : C->rigid) + (size_t)e * B * 13;
  if (l == 0 && part <= 0) {
#pragma unroll
    for (int i = 0; i < 13; ++i) rb[i] = root[i];
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (part >= 0 && part != j) continue;
    float* o = rb + (size_t)(1 + per_leg * l + j) * 13;
    float qq[4]; mat_to_quat(k.R[j], qq);
    o[0] = k.O[j].x; o[1] = k.O[j].y; o[2] = k.O[j].z; o[3] = qq[0]; o[4] = qq[1]; o[5] = qq[2]; o[6] = qq[3];
    o[7] = k.vO[j].x; o[8] = k.vO[j].y; o[9] = k.vO[j].z; o[10] = k.w[j].x; o[11] = k.w[j].y; o[12] = k.w[j].z;
  }
  if (per_leg == 4 && (part < 0 || part == 2)) {
    float* o = rb + (size_t)(1 + per_leg * l + 3) * 13;
    V3 r = mul(k.R[2], lm_.v(LM_FOOT_POS));
    V3 p = k.O[2] + r, v = k.vO[2] + cross(k.w[2], r);
    M3 fr;
#pragma unroll
    for (int i = 0; i < 9; ++i) fr.m[i] = lm_.f(LM_FOOT_ROT + i);
    float qq[4]; mat_to_quat(mul(k.R[2], fr), qq);
    o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = qq[0]; o[4] = qq[1]; o[5] = qq[2]; o[6] = qq[3];
    o[7] = v.x; o[8] = v.y; o[9] = v.z; o[10] = k.w[2].x; o[11] = k.w[2].y; o[12] = k.w[2].z;
    if (foot_row_lds) {       // the fused step's reward terms read the feet rows from LDS
#pragma unroll
      for (int i = 0; i < 13; ++i) foot_row_lds[i] = o[i];
    }
    if (foot_z_dst) *foot_z_dst = p.z;                  // gait_foot_z: the heights handed to GaitScheduler.step (anymal.py:107-110)
  } else if (per_leg == 3 && (part < 0 || part == 2)) {   // no separate foot body: "the foot" is the last link
    const float* o = rb + (size_t)(1 + per_leg * l + 2) * 13;
    if (foot_row_lds) {
#pragma unroll
      for (int i = 0; i < 13; ++i) foot_row_lds[i] = o[i];
    }
    if (foot_z_dst) *foot_z_dst = k.O[2].z;
  }
}

// persisted closest-point cache of one lane's slots [s0, s0 + 2): global [env][leg][slot][4] <-> LDS [slot][4][lane]
template <bool LOAD>
LG_DEV void mesh_cache_io(const DevCtx* __restrict__ C, float* cqc, int e, int l, int lane, int s0) {
  float4* g = (float4*)(C->mesh_cache + ((size_t)e * NLEG + l) * LG_MAX_CP * 4);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int sl = s0 + i;
    if (LOAD) {
      const float4 v = g[sl];
      cqc[(sl * 4 + 0) * 64 + lane] = v.x; cqc[(sl * 4 + 1) * 64 + lane] = v.y; cqc[(sl * 4 + 2) * 64 + lane] = v.z; cqc[(sl * 4 + 3) * 64 + lane] = v.w;
    } else {
      g[sl] = make_float4(cqc[(sl * 4 + 0) * 64 + lane], cqc[(sl * 4 + 1) * 64 + lane], cqc[(sl * 4 + 2) * 64 + lane], cqc[(sl * 4 + 3) * 64 + lane]);
    }
  }
}

LG_DEV void publish_state(float* rec, const float root[13], const float q[3], const float qd[3], float extra = 0.f) {
  float4* p = reinterpret_cast<float4*>(rec);
  p[0] = make_float4(root[0], root[1], root[2], root[3]); p[1] = make_float4(root[4], root[5], root[6], root[7]);
  p[2] = make_float4(root[8], root[9], root[10], root[11]); p[3] = make_float4(root[12], q[0], q[1], q[2]);
  p[4] = make_float4(qd[0], qd[1], qd[2], extra);      // (slot 19: free for the caller)
}
LG_DEV void fetch_state(const float* rec, float root[13], float q[3], float qd[3]) {
  const float4* p = reinterpret_cast<const float4*>(rec);
  const float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
  root[0] = a.x; root[1] = a.y; root[2] = a.z; root[3] = a.w; root[4] = b.x; root[5] = b.y; root[6] = b.z; root[7] = b.w;
  root[8] = c.x; root[9] = c.y; root[10] = c.z; root[11] = c.w; root[12] = d.x; q[0] = d.y; q[1] = d.z; q[2] = d.w;
  qd[0] = e.x; qd[1] = e.y; qd[2] = e.z;
}

LG_DEV void store_lstm_rows(const DevCtx* __restrict__ C, size_t row, size_t N12, const float* h0, const float* c0, const float* h1, const float* c1) {
  float LG_G* const sh = C->sea_h; float LG_G* const sc = C->sea_c;      // (both pointers in front of the first store: see fused_writeback_obs)
  float4* p = (float4*)(sh + row * 8);
  p[0] = make_float4(h0[0], h0[1], h0[2], h0[3]); p[1] = make_float4(h0[4], h0[5], h0[6], h0[7]);
  p = (float4*)(sc + row * 8);
  p[0] = make_float4(c0[0], c0[1], c0[2], c0[3]); p[1] = make_float4(c0[4], c0[5], c0[6], c0[7]);
  p = (float4*)(sh + (N12 + row) * 8);
  p[0] = make_float4(h1[0], h1[1], h1[2], h1[3]); p[1] = make_float4(h1[4], h1[5], h1[6], h1[7]);
  p = (float4*)(sc + (N12 + row) * 8);
  p[0] = make_float4(c1[0], c1[1], c1[2], c1[3]); p[1] = make_float4(c1[4], c1[5], c1[6], c1[7]);
}

#endif           // NJ == 3

// ============================================================================================ physics kernel
// MODE 0: fused step (clip actions, nsub x (actuator + physics)); MODE 1: lg_simulate (one dt, torques from LG_T_TORQUES);
// MODE 2: lg_compute_torques only.
// Optional second destinations of a step that feeds a rollout storage directly (lg_step_transition), see post_instance
struct PostSink { float* obs_out; const float* values; float* rewards; float* dones; float gamma;
                  float* rew_out; int rew_stride;          // (fused rollout steps: the reward column of lg_rollout_batch's (n, horizon) matrix)
                  uint8_t* reset_rows; uint8_t* tout_rows; int obs_by_row;      // lg_step_subset_rows: dense rows by position in the id list (obs_out then too)
                  int nsteps; };                                                // persistent rollout launch (SPEC & 3 == 3): steps this launch runs, action rows NDOF apart, reward columns 1 apart

// the post-physics step as the tail of this kernel (lg_fused_post.h, defined below the post-physics helpers)
struct FusedMainIn;
// ... and for an instance without that tail: post_instance<true> on the four waves of the workgroup, its LDS in the dead contact-slot table (defined behind post_instance)
LG_DEV void generic_fused_tail(const DevCtx* __restrict__ C, float* lds, const int32_t* __restrict__ ids, int n, int mode, const PostSink& K);
LG_DEV void fused_prefetch(const DevCtx* __restrict__ C, float* SR, float* UB, int blk, int n, int htid, int64_t step, const float* values, const int32_t* __restrict__ ids, bool ro);
// what the helper waves fetch in front of (F) for their work behind it (lg_fused_post.h): their scan point, their entries of the observation
// table, and every context member / config flag that work reads -- behind (F) each of those was a scalar round trip of its own on the way to (G2)
struct FusedPre { float bx, by; float4 tab[2];
                  int inject, gait_on, feet_early, heights_early, per_leg, B, P, plane; float vscale;
                  float LG_G *rigid, *gfz, *act, *tq, *heights; };
LG_DEV void fused_height_scan(const DevCtx* __restrict__ C, const float (*xst)[20], float* HB, int blk, int n, int htid, const int32_t* __restrict__ ids, bool ro, const FusedPre& F);
LG_DEV void fused_prefetch_static(const DevCtx* __restrict__ C, const float* hot, int htid, FusedPre& F);
enum { NZ_IT = 6 };          // Philox calls per helper lane that cover the observation noise of the workgroup's 16 envs (rows of up to 256 entries)
LG_DEV bool fused_noise_predrawn(const float* hot);
LG_DEV void fused_noise_draw(const float* hot, int blk, int n, int htid, int64_t step, float nz[NZ_IT][4], const int32_t* __restrict__ ids, bool ro);
LG_DEV void fused_noise_park(const float* hot, float* HB, int blk, int n, int htid, const float nz[NZ_IT][4]);
LG_DEV void fused_stage_obs_table(const float* hot, float* HB, int htid, const FusedPre& F);
LG_DEV void fused_main_and_serial(const DevCtx* __restrict__ C, const float* hot, const LegModel& lm_, float* xs, float* UB, float* HB, int lane, int e, bool valid,
                                  const float* root, const float* q, const float* qd, const float* tau, const float* last_qd, const V3* fbody,
                                  const float* act_or_null, bool fault, int64_t step, unsigned long long* stamps, const PostSink& K, bool ro, int krow);
// blockIdx -> env block with the blocks of one XCD contiguous (blockIdx % 8 = XCD, round-robin dispatch): a bijection on [0, nb)
LG_DEV int xcd_block(unsigned b, unsigned nb) {
  const unsigned x = b & 7u, i = b >> 3, per = nb >> 3, rem = nb & 7u;
  return (int)(x * per + min(x, rem) + i);
}
// The context pointer behind an opaque zero: loads through it cannot be scheduled in front of this point.  The tail of a fused step reads ~40
// row pointers and scalars of the context; hoisted to the top of the kernel as "invariant" scalar loads they sat in SGPRs through the whole
// substep loop -- i.e. were spilled to VGPR lanes in the wave prologues and read back in the tail (~140 of the kernel's ~200 SGPR spills).
LG_DEV const DevCtx* late_ctx(const DevCtx* C) {
  long zero;
  asm volatile("s_mov_b64 %0, 0" : "=s"(zero));
  return reinterpret_cast<const DevCtx*>(reinterpret_cast<const char*>(C) + zero);
}
#define FUSED_STATS_WAVE 1   // which wave of a fused workgroup adds the statistics, draws the arrival ticket and tests for the last arrival (a helper wave: with the rigid-body rows moved in front of (G2) the helpers reach the write-back with less left to do than the main wave; A/B -0.5 %)
LG_DEV bool fused_writeback_obs(const DevCtx* __restrict__ C, const float* hot, const float* SR, const float* HB, int blk, int n, int tid, int64_t step, unsigned long long* stamps, float* obs_out, bool obs_by_row,
                                const int32_t* __restrict__ ids, bool ro, bool arrive);
LG_DEV void fused_finalize(const DevCtx* __restrict__ C, int nblocks, int tid, bool ro, int nsteps);
LG_DEV bool fused_did_reset(const float* HB, int el);
LG_DEV void fused_reload_state(const float* SR, int lane, int l, float* root, float* q, float* qd);
LG_DEV float* fused_foot_row(float* xs, int lane);
LG_DEV float* fused_act_slot(float* xs, int lane, int d);
LG_DEV bool fused_needs_heights_early(const DevCtx* __restrict__ C);
LG_DEV bool fused_needs_feet_rows(const DevCtx* __restrict__ C);

#if NJ == 3
// HELPERS: the launch has the three helper waves (every policy step unless LG_SPLIT=0).  A separate instance, so that the kernel the
// headline runs does not carry the single-wave fallback (the whole LSTM inlined in the main wave, inline leg bias and contact
// detection): that dead code accounted for most of the register spills the compiler reported for the kernel.
// SPEC & 3: 1 = TGS + pyramid friction rows fixed at compile time (A/B build 13 only, see physics_substep); 2 = the fused tail in its rollout variant.
// SPEC >> 2 = FEAT of physics_substep: bit 0 capsule parts (sliding spheres), bit 1 the self-collision pass.  Launches with helper waves pick the
// instance the model / config needs (launch_physics); the single-wave instances always carry both (FEAT_ALL: a model without sliding spheres or
// pairs takes the same paths with an empty mask / list), so the kernel of a robot of fixed spheres without self-collision is what it was.
#define FEAT_ALL 12
#ifndef LG_LSTM_WARM
#define LG_LSTM_WARM 1       // helper waves warm the scalar cache with the LSTM weights at kernel entry (0: A/B)
#endif
#ifndef LG_CAPS_DEAL
#define LG_CAPS_DEAL (LG_LEGS == 4)      // the capsule-segment instance deals the contact slots 2 / 1 / 3 / 2 over main / waves 1-3 (0: 3 / 1 / 2 / 2 like the plain instance; A/B)
#endif
template <int MODE, bool TMESH, bool HELPERS = false, int SPEC = 0>
#if LG_AB == 9      // timing probe: cap the kernel at the 256 registers per wave that two workgroups per CU would leave (spills go to scratch)
__attribute__((amdgpu_num_vgpr(120)))
#endif
__global__ __launch_bounds__(256) void physics_kernel(const DevCtx* __restrict__ C, const float* __restrict__ actions_in, int nsub, int nact,
                                                      const int32_t* __restrict__ ids, int n, int act_stride, int fuse_arg, PostSink sink) {
  // fuse_arg != 0: the launch ends the policy step itself -- through the hand-tuned tail of lg_fused_post.h (four-legged instance: `fuse`), or, for an
  // instance without one (six legs, height grid / plane: `gfuse`, round 5), by running the post kernel's own code (post_instance<true>: one wave per two
  // envs) behind the write-back: the same arithmetic in the same order as the two-launch path, minus a kernel boundary.
  constexpr bool GFUSABLE = LG_LEGS == 6 && !TMESH && MODE == 0 && HELPERS;      // (triangle-mesh instances of the hexapod have no LDS left for the statistics step's arrays)
  const int fuse = LG_LEGS == 4 ? fuse_arg : 0;
  const bool gfuse = GFUSABLE && fuse_arg != 0;
  // act_stride: floats between consecutive action rows (12, or horizon * 12 when the rows are one step of a (n, horizon, 12) plan)
  // contact-detection split of a heightfield step with actuator waves (see the helper loop); mesh terrains: two slots a wave
  // (heightfield / plane steps; on triangle-mesh terrains every wave takes the slot pair [2 w, 2 w + 2), the main wave [0, 2))
  // Triangle-mesh steps: every wave takes one PAIR of slots (one traversal serves two neighbouring spheres).  The feet and the lowest shank spheres
  // (slots 0, 1: always near the ground, never skipped by the distance cache) are the expensive pair: they go to a helper wave -- wave 2, wave 1 has
  // the leg bias as well --, the main wave takes the cheapest pair (6, 7: trunk spheres, skipped by the cache in nearly every substep).  Round 4 had the
  // main wave on (0, 1): its queries were 76 % of the kernel on config 3 while the helper waves of a PD robot idled (A/B build 31 = that deal).
#if LG_AB == 31
#define MESH_PAIR0(wv_) (2 * (wv_))
#else
#define MESH_PAIR0(wv_) ((wv_) == 0 ? 6 : ((wv_) == 1 ? 2 : ((wv_) == 2 ? 0 : 4)))
#endif
  // (round 6, capsule-segment instance of the quadruped: 2 / 1 / 3 / 2 -- with a segment slot on it the main wave was the LAST at (A2): per-wave stamps,
  //  profiles/r06_schedule_experiments.txt, r06_phase_stamps_wave{1,2,3}.txt: helper waves 1 / 2 / 3 waited 1.5 / 2.8 / 1.4 k cycles for it there; one plain slot moved to wave 2)
  constexpr bool CAPS_DEAL = LG_CAPS_DEAL && ((SPEC >> 2) & 1) && !TMESH;
  constexpr int DS0 = CAPS_DEAL ? 2 : 3, DS1 = CAPS_DEAL ? 3 : 4, DS2 = 6;   // main 3 / wave 1 (which also has the leg bias) 1 / 2 / 2: the helpers are the last to arrive at (A2), the main wave has ~3 k cycles of slack there (A/B in one session: 2/2/2/2 +1.4 us, 4/0/2/2 +0.3 us; 1/2/2/3 and 0/2/3/3: worse still; the capsule-segment instance, round 5: 4/0/2/2 +3.5 us, 2/1/2/3 +0.3 us against this deal)
  // Workgroup = 16 envs.  Wave 0 ("main") runs the dynamics, one leg per lane.  With nact == 3 (fused step with the
  // LSTM actuator) waves 1..3 are actuator waves: wave w evaluates joint w-1 of every leg, concurrently with the main
  // wave's torque-independent work (kinematics, bias, mass matrix, contact set-up); they meet at two barriers per substep.
  __shared__ __attribute__((aligned(16))) float cst[LG_CST_FLOATS];
  __shared__ float lmod[LM_FIELDS * GRP];
#ifdef LG_STAMPS
  const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
  const float* __restrict__ wlstm = C->lstm_w;
  // state of the substep published by the main wave for the helpers: [lane][root 13 | q 3 | qd 3 | pad] = five 16-byte units per
  // lane (odd: conflict-free), written / read with ds_*_b128 -- a lone wave gets the full LDS rate only on 16-byte accesses
  __shared__ __attribute__((aligned(16))) float xst[64][20];
  __shared__ float xtau[3][64];
  __shared__ __attribute__((aligned(16))) float xbias[64][12];   // leg bias of wave 1: [lane][bk 3 | Fs 3 | Ns 3 | pad], three 16-byte units
  __shared__ __attribute__((aligned(16))) float xs[(TMESH ? XS_STRIDE : XS_STRIDE_PK) * 64];              // main wave's mass-matrix factors for the helpers' share of the contact set-up
  __shared__ float cqc[TMESH ? LG_MAX_CP * 4 * 64 : 1];   // mesh terrains: last closest-point query of every collision sphere
  __shared__ int s_last_f;                                 // fused step: this workgroup is the last of the launch to arrive
  constexpr bool FUSABLE = LG_LEGS == 4;                    // (can_fuse: the fused tail exists for the four-legged instance only; the others keep the LDS)
  __shared__ __attribute__((aligned(16))) float hot[FUSABLE ? ((HC_COUNT + 3) & ~3) : 4];             // fused step: the scalars of the post-physics tail (HC_*)
  // A/B build 7 only: an LDS copy of the 905 gate-interleaved LSTM weights for the actuator waves (one ds_read_b128 at a wave-uniform
  // address = four weights).  Measured against the scalar loads (s_load_dwordx16 -> SGPR pairs feeding the packed FMAs) in one session:
  // 0.0897 ms per step from LDS, 0.0818 ms through SGPRs -- the scalar unit fetches the weights beside the vector ALU, while LDS
  // reads take issue slots of the wave whose chain of gate evaluations is the critical path in front of rendezvous (A2).
  constexpr bool LSTM_LDS = !TMESH && LG_LSTM_LDS;
  __shared__ __attribute__((aligned(16))) float wlds[LSTM_LDS ? LW_COUNT + 3 : 4];
  // fused ROLLOUT step of an env subset (lg_step_subset rollout_mode = 1, lg_rollout_batch): its own instance (SPEC = 2), so that the
  // full step's tail carries none of the variant's selects (as run-time branches they cost the headline step 1.3 %: A/B in one session)
  // SPEC & 3 == 3: the same tail, and the launch runs sink.nsteps rollout steps of its envs back to back (lg_rollout_batch: one launch per horizon).  A
  // workgroup owns its envs for the whole horizon and rollout steps never reset, so the main wave keeps the robot state and the helper waves the LSTM state
  // in registers from step to step; what the tail exchanges through memory (history rows, written by one wave and read by another of the SAME workgroup one
  // step later) is ordered by the workgroup barrier at the end of a step.  Arrivals are counted in the last step only.
  constexpr bool ro = (SPEC & 3) >= 2;
  constexpr bool PERSIST = (SPEC & 3) == 3;
  const DevCtx* const Cq = C;                             // (the step loops shadow C)
  // ... and the three values every row address of the tail is formed from: loop-invariant otherwise, i.e. some sixty 64-bit addresses per lane computed in front
  // of the step loop and kept through all of it
  auto opaque_v = [](int v) { if (PERSIST) asm volatile("" : "+v"(v)); return v; };
  auto opaque_s = [](int v) { if (PERSIST) asm volatile("" : "+s"(v)); return v; };
  // the self-collision pair table (96 x 16 B) in LDS, except where the workgroup has no 1.5 KB left (six legs on a triangle mesh: 163 840 B taken)
  constexpr bool SC_LDS = ((SPEC >> 2) & 2) && !(TMESH && NLEG == 6);
  __shared__ uint4 sctab[SC_LDS ? LG_MAX_SC_PAIRS : 1];
  const int nsteps = PERSIST ? (sink.nsteps > 1 ? sink.nsteps : 1) : 1;
  constexpr int FEAT = SPEC >> 2;
  constexpr bool CAPS = !TMESH && (FEAT & 1);
  constexpr bool MCAPS = TMESH && (FEAT & 1);               // capsule segments against the edges of a GRID mesh (contact_detect_mesh<true>; launch_physics picks the instance when the terrain has TerrainView::SEG4)
  const int32_t* const fids = ro ? ids : nullptr;           // the tail's row -> env map: a literal null (rows = envs) in the full step's instance
  int64_t fstep = ro ? C->counters[3] + 1 : C->counters[0] + 1;         // LR:123 (the statistics step of the previous launch stored it)
  const int64_t gstep_f = C->counters[0] + 1;             // what the gait term's "has a scheduler step run yet" test sees (post_instance: gstep)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // Which block of 16 envs this workgroup steps.  Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8) and every XCD has its own
  // L2; on triangle-mesh terrains the BVH (13 MB of nodes + 34 MB of triangles on config 3) is what the contact queries read, and envs with
  // neighbouring indices stand next to each other on the terrain -- so XCD x takes the x-th CONTIGUOUS eighth of the env blocks and its
  // L2 sees one eighth of the map instead of all of it.
  const int bid = TMESH ? xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  // row kq of the launch <-> env e (identity, or ids[kq] for subset stepping: main-only / rollout-only steps)
  const int kq = bid * EPB + lane / GRP;
  const int lg_ = lane % GRP;               // lane of the env's group
  const bool has_leg = lg_ < NLEG;          // six legs on eight lanes: lanes 6 and 7 shadow leg 0 of their env and store nothing
  const int l = has_leg ? lg_ : 0;          // the leg whose rows this lane reads (and, with `valid`, writes)
  const bool env_ok = kq < n;
  const bool valid = env_ok && has_leg;
  const int krow = env_ok ? kq : n - 1;     // whole groups are (in)valid together; invalid groups compute on a copy and store nothing
  const int e = ids ? ids[krow] : krow;
  const int e_q = e, krow_q = krow, bid_q = bid;
  const lg_robot_model* __restrict__ m = &C->model; (void)m;
  const lg_config& g = C->cfg;
  const bool net = g.control_type == LG_CTRL_ACTUATOR_NET;
  // The rows this wave starts from are requested in FRONT of the table copy and its barrier (loads return in order: by the time the
  // table words can be stored to LDS these have landed too): one memory round trip at kernel entry instead of two.
  const bool helper_wave = MODE == 0 && HELPERS && wv > 0;
  float pre_root[13], pre_dof[6], pre_lqd[3], pre_mu = 0.f, pre_madd = 0.f;
  float4 pre_sea[8];
  float pre_act = 0.f;
  if (!helper_wave) {
#pragma unroll
    for (int i = 0; i < 13; ++i) pre_root[i] = C->root[(size_t)e * 13 + i];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      pre_dof[2 * j] = C->dof[((size_t)e * NDOF + 3 * l + j) * 2];
      pre_dof[2 * j + 1] = C->dof[((size_t)e * NDOF + 3 * l + j) * 2 + 1];
      pre_lqd[j] = C->last_dof_vel[(size_t)e * NDOF + 3 * l + j];
    }
    if (MODE != 2) { pre_mu = C->friction[e]; pre_madd = C->mass_added[e]; }
  } else if (net) {
    const size_t N12p = (size_t)C->N * NDOF, rowp = (size_t)e * NDOF + 3 * l + (wv - 1);
    const float4* p0 = (const float4*)(C->sea_h + rowp * 8); const float4* p1 = (const float4*)(C->sea_c + rowp * 8);
    const float4* p2 = (const float4*)(C->sea_h + (N12p + rowp) * 8); const float4* p3 = (const float4*)(C->sea_c + (N12p + rowp) * 8);
    pre_sea[0] = p0[0]; pre_sea[1] = p0[1]; pre_sea[2] = p1[0]; pre_sea[3] = p1[1];
    pre_sea[4] = p2[0]; pre_sea[5] = p2[1]; pre_sea[6] = p3[0]; pre_sea[7] = p3[1];
    pre_act = actions_in[(size_t)krow * act_stride + 3 * l + (wv - 1)];
  }
  fill_leg_model(lmod, C->lmod, threadIdx.x, blockDim.x);
#if LG_LSTM_WARM
  // (round 6) The scalar cache is invalidated at every launch, and the helper waves' first recurrent half fetches 2 KB of LSTM weights through it in six
  // dependent batches of s_load_dwordx16 -- the main wave waited ~6 k cycles at the first barrier (A) for that.  A helper wave touches every 64-byte line
  // of the weight block now, while its state rows are still on their way from HBM: one round of independent scalar loads, and the recurrent half hits.
  if (helper_wave && net && !LSTM_LDS) {
    float warm = 0.f;
#pragma unroll
    for (int i = 0; i < LW_COUNT; i += 16) warm += wlstm[i];
    asm volatile("" :: "s"(warm));
  }
#endif
  if (SC_LDS) for (int i = threadIdx.x; i < C->n_sc; i += blockDim.x) sctab[i] = C->sc_tab[i];      // (first read by the main wave behind (A2) of the first substep)
  if (FUSABLE && fuse && wv == 1) for (int i = lane; i < HC_COUNT; i += 64) hot[i] = C->hot[i];
  if (LSTM_LDS && MODE == 0 && net && wv >= 2) for (int i = (wv - 2) * 64 + lane; i < LW_COUNT; i += 128) wlds[i] = wlstm[i];
  // With helper waves nobody reads these tables before rendezvous (A) of the first substep (every wave's first use is the kinematics behind it), and
  // every wave has filled its share before it gets there: (A) stands in for a barrier here, and the helper waves start their first recurrent half
  // as soon as their own rows have landed instead of waiting for every wave's (A/B: -0.7 % on the step)
  if (!(MODE == 0 && HELPERS && !LSTM_LDS)) lds_barrier();
#ifdef LG_STAMPS
  const unsigned long long t_bar0 = __builtin_amdgcn_s_memtime();
#endif
  const LegModel lm_{lmod, lg_};

  if (MODE == 0 && HELPERS && wv > 0) {
    // ---------------------------------------------------------------- actuator wave: joint j of leg l of env e
    const int j = wv - 1, d = 3 * l + j;
    // with the actuator network this wave also evaluates joint j of every leg; with PD control (helpers are then only
    // present for triangle-mesh terrains) the main wave keeps the torques and barrier (B) does not exist
    float a = net ? pre_act : 0.f;
    a = fminf(fmaxf(a, -g.clip_actions), g.clip_actions);
    const size_t N12_o = (size_t)C->N * NDOF, row_o = (size_t)e * NDOF + d;
    float h0[8], c0[8], h1[8], c1[8];
    if (net) {
      float4 u = pre_sea[0], v = pre_sea[1];
      h0[0] = u.x; h0[1] = u.y; h0[2] = u.z; h0[3] = u.w; h0[4] = v.x; h0[5] = v.y; h0[6] = v.z; h0[7] = v.w;
      u = pre_sea[2]; v = pre_sea[3];
      c0[0] = u.x; c0[1] = u.y; c0[2] = u.z; c0[3] = u.w; c0[4] = v.x; c0[5] = v.y; c0[6] = v.z; c0[7] = v.w;
      u = pre_sea[4]; v = pre_sea[5];
      h1[0] = u.x; h1[1] = u.y; h1[2] = u.z; h1[3] = u.w; h1[4] = v.x; h1[5] = v.y; h1[6] = v.z; h1[7] = v.w;
      u = pre_sea[6]; v = pre_sea[7];
      c1[0] = u.x; c1[1] = u.y; c1[2] = u.z; c1[3] = u.w; c1[4] = v.x; c1[5] = v.y; c1[6] = v.z; c1[7] = v.w;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) { h0[i] = 0.f; c0[i] = 0.f; h1[i] = 0.f; c1[i] = 0.f; }
    }
#pragma unroll 1
    for (int st = 0;; ++st) {                            // (one pass unless PERSIST: the steps of a persistent rollout launch)
    // PERSIST: every step reads the context afresh through an opaque pointer, as a launch of its own would -- otherwise every context member the
    // step reads is loop-invariant, gets hoisted in front of the step loop and stays live through all of it (138-156 VGPR spills, measured)
    const DevCtx* __restrict__ const C = PERSIST ? late_ctx(Cq) : Cq;
    const lg_config& g = C->cfg;
    const int e = opaque_v(e_q), krow = opaque_v(krow_q), bid = opaque_s(bid_q);
    const int htid = opaque_v((wv - 1) * 64 + lane), tid_it = opaque_v((int)threadIdx.x);      // (what the tail's lane -> (env, entry) maps are formed from)
    const size_t N12 = PERSIST ? (size_t)C->N * NDOF : N12_o, row = PERSIST ? (size_t)e * NDOF + d : row_o;      // (PERSIST: formed from this step's opaque e)
    if (PERSIST && st > 0 && net) {                      // this step's action row of the plan; the LSTM state stays in the registers
      a = actions_in[(size_t)krow * act_stride + (size_t)st * NDOF + 3 * l + (wv - 1)];
      a = fminf(fmaxf(a, -g.clip_actions), g.clip_actions);
    }
    PhysParams P;
    P.dt = g.sim_dt; P.grav = v3(g.gravity[0], g.gravity[1], g.gravity[2]); P.iters = g.solver_iterations;
    P.contact_offset = g.contact_offset; P.max_depen = g.max_depenetration_velocity; P.erp = g.erp; P.cfm = g.cfm; P.solver = g.solver_type; P.fric = g.friction_model;
    P.terrain_mu = C->terrain_mu; P.slide_mask = (CAPS || (MCAPS && C->ter.SEG4)) ? C->slide_mask : 0u; P.slot_perm = CAPS ? C->slot_perm : (TMESH ? C->mesh_perm : 0x76543210u); P.cache_reach = TMESH ? C->mesh_reach : 0.f;
#if LG_AB == 21
    P.slide_mask = 0u;
#endif
    const TerrainView T = C->ter;
    if (TMESH && st == 0) mesh_cache_io<true>(C, cqc, e, l, lane, MESH_PAIR0(wv));   // this wave's two slots of the persisted query cache -> LDS
#ifdef LG_STAMPS
#ifndef LG_STAMP_WAVE
#define LG_STAMP_WAVE 2          // which helper wave the diagnostic build watches
#endif
    unsigned long long* stamps = (blockIdx.x == 0 && lane == 0 && wv == LG_STAMP_WAVE) ? C->stamps : nullptr;
    unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
#endif
    LstmPre lpre;
    if (net) {
      if (LSTM_LDS) lstm_recurrent_part(wlds, h0, h1, lpre);
      else { int zero; asm volatile("s_mov_b32 %0, 0" : "=s"(zero)); lstm_recurrent_part(wlstm + zero, h0, h1, lpre); }
    }
#pragma unroll 1
    for (int sub = 0; sub < nsub; ++sub) {
      lds_barrier();                                   // (A) main wave has published root, q, qd of this substep
      STAMP(40);
      // torque-independent share of the dynamics: this leg's kinematics, then the leg bias (wave 1) or the contact
      // detection of half of the slots (waves 2, 3), straight into the LDS the main wave reads after barrier (A2);
      // then this wave's joint of the actuator network, all while the main wave factorises the mass matrix
      float r13[13], qq[3], qdd[3];
      fetch_state(xst[lane], r13, qq, qdd);
      const M3 Rb = quat_to_mat(r13 + 3);
      const V3 pb = v3(r13[0], r13[1], r13[2]), vb = v3(r13[7], r13[8], r13[9]), wb = v3(r13[10], r13[11], r13[12]);
      LegKin k;
      leg_kinematics(lm_, Rb, pb, vb, wb, qq, qdd, k);
      if (sub == 0) STAMP(53); else STAMP(41);
      // heightfield terrains: two slots per wave (the main wave takes slots 0, 1); this wave issues its height-sample
      // loads now and uses them after the actuator network
      // slot ranges [DS0, DS1) wave 1 (which also has the leg bias), [DS1, DS2) wave 2, [DS2, 8) wave 3; the main wave
      // takes [0, DS0) while it waits for nothing else
      constexpr int DS1P = DS1 > DS0 ? DS1 : DS0 + 1;   // (wave 1 may have no slot at all: its probe type still needs a size)
      ContactProbe<DS0, DS1P> pr1; ContactProbe<DS1, DS2> pr2; ContactProbe<DS2, 8> pr3;
      ContactProbeC<DS0, DS1P> pc1; ContactProbeC<DS1, DS2> pc2; ContactProbeC<DS2, 8> pc3;     // (capsule-part instances; dead otherwise)
      static_assert(LG_MAX_CP == 8, "slot split assumes 8 contact slots");
      if (wv == 1) {
        float bk[3]; V3 Fs, Ns;
        leg_bias(lm_, k, pb, wb, qdd, P.grav, bk, Fs, Ns);
        {
          float4* pb4 = reinterpret_cast<float4*>(xbias[lane]);
          pb4[0] = make_float4(bk[0], bk[1], bk[2], Fs.x); pb4[1] = make_float4(Fs.y, Fs.z, Ns.x, Ns.y); pb4[2] = make_float4(Ns.z, 0.f, 0.f, 0.f);
        }
        if (!TMESH) { if (DS0 < DS1) { if (CAPS) contact_detect_begin_caps<DS0, DS1P>(lm_, T, k, Rb, pb, P.slide_mask, pc1, P.slot_perm); else contact_detect_begin<DS0, DS1P>(lm_, T, k, Rb, pb, pr1); } }
        else contact_detect_mesh<MCAPS>(MESH_PAIR0(1), MESH_PAIR0(1) + 2, lm_, T, P, k, Rb, pb, cst, lane, cqc);
      } else if (TMESH) {
#ifdef LG_STAMPS
        contact_detect_mesh<MCAPS>(MESH_PAIR0(wv), MESH_PAIR0(wv) + 2, lm_, T, P, k, Rb, pb, cst, lane, cqc,
#ifdef LG_STAMP_MAIN_MESH
                            nullptr);
#else
                            (blockIdx.x == 0 && wv == 2) ? C->stamps : nullptr);
#endif
#else
        contact_detect_mesh<MCAPS>(MESH_PAIR0(wv), MESH_PAIR0(wv) + 2, lm_, T, P, k, Rb, pb, cst, lane, cqc);
#endif
      } else if (wv == 2) {
        if (CAPS) contact_detect_begin_caps<DS1, DS2>(lm_, T, k, Rb, pb, P.slide_mask, pc2, P.slot_perm); else contact_detect_begin<DS1, DS2>(lm_, T, k, Rb, pb, pr2);
      } else {
        if (CAPS) contact_detect_begin_caps<DS2, 8>(lm_, T, k, Rb, pb, P.slide_mask, pc3, P.slot_perm); else contact_detect_begin<DS2, 8>(lm_, T, k, Rb, pb, pr3);
      }
      if (sub == 0) STAMP(54); else STAMP(42);
      if (net) {
        const float tgt = a * g.action_scale + lm_.f(LM_DEFAULT_POS + j);      // (the model table: first read behind (A), see the kernel's first barrier)
        const float x0 = (tgt - qq[j]) * g.actuator_in_scale[0], x1 = qdd[j] * g.actuator_in_scale[1];
        // an opaque zero keeps the ~60 weight addresses from being hoisted out of the substep loop as loop invariants
        // (they would fill the SGPR file and spill): inside the loop they fold into the s_load immediate offsets
        if (LSTM_LDS) xtau[j][lane] = lstm_input_part(wlds, x0, x1, lpre, h0, c0, h1, c1, g.actuator_out_scale);
        else {
          int zero; asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
          xtau[j][lane] = lstm_input_part(wlstm + zero, x0, x1, lpre, h0, c0, h1, c1, g.actuator_out_scale);
        }
      }
      if (!TMESH && wv == 1) { if (DS0 < DS1) { if (CAPS) contact_detect_finish_caps<DS0, DS1P>(lm_, T, P, pb, P.slide_mask, pc1, cst, lane, P.slot_perm); else contact_detect_finish<DS0, DS1P>(lm_, T, P, pb, pr1, cst, lane); } }
      else if (!TMESH && wv == 2) { if (CAPS) contact_detect_finish_caps<DS1, DS2>(lm_, T, P, pb, P.slide_mask, pc2, cst, lane, P.slot_perm); else contact_detect_finish<DS1, DS2>(lm_, T, P, pb, pr2, cst, lane); }
      else if (!TMESH && wv == 3) { if (CAPS) contact_detect_finish_caps<DS2, 8>(lm_, T, P, pb, P.slide_mask, pc3, cst, lane, P.slot_perm); else contact_detect_finish<DS2, 8>(lm_, T, P, pb, pr3, cst, lane); }
      if (sub == 0) STAMP(55); else STAMP(43);
      lds_barrier();                                   // (A2) bias, contact detection, torques | mass-matrix factors
      if (sub == 0) STAMP(56); else STAMP(44);
      // this wave's share of the contact set-up (every fourth active slot)
      {
        MassFactorsP MF;
        float Mi[6], Mbk[6][3], Y[3][6], Si[21];
        const unsigned slot_mask = active_slot_mask(cst, lane);
        if (slot_mask) {
          if (TMESH) fetch_mass_factors(xs, lane, Mi, Mbk, Y, Si); else fetch_mass_factors_pk(xs, lane, MF);
          int seen = 0;
#pragma unroll 1
          for (int sl = 0; sl < LG_MAX_CP; ++sl) {
            if (!((slot_mask >> sl) & 1u)) continue;
            if ((seen++ & 3) != wv - 1) continue;       // set-up order: waves 1, 2, 3, then the main wave (A/B: -1.9 us on the kernel)
            if (TMESH) contact_setup_slot(sl, lm_, k, pb, Mi, Mbk, Y, Si, P.cfm, P.fric, cst, lane);
            else contact_setup_slot_pk(sl, lm_, k, pb, MF, P.cfm, P.fric, cst, lane);
          }
        }
      }
      // this wave's quarter of the self-collision pair filter (the records it reads are complete since (A2); the published state in xst is dead since every
      // helper wave fetched it behind (A): its first 768 bytes carry the masks to the main wave)
      if (FEAT & 2) reinterpret_cast<unsigned*>(&xst[0][0])[htid] = sc_prefilter(cst, SC_LDS ? sctab : C->sc_tab, C->n_sc, lane, wv, 4);
      if (sub == 0) STAMP(57); else STAMP(45);
      lds_barrier();                                   // (A3) slot table complete
      STAMP(22);                                       // (diagnostic)
      if (net && sub + 1 < nsub) {                     // while the main wave sweeps: recurrent half of the next substep's network
        if (LSTM_LDS) lstm_recurrent_part(wlds, h0, h1, lpre);
        else { int zero; asm volatile("s_mov_b32 %0, 0" : "=s"(zero)); lstm_recurrent_part(wlstm + zero, h0, h1, lpre); }
      }
      STAMP(23);                                       // (diagnostic)
      if (fuse && sub + 1 == nsub) {                   // ... in the last substep: what the post-physics tail needs from HBM
        if (net) *fused_act_slot(xs, lane, d) = a;     // (the mass-factor table is dead after (A3): the env rows live there)
        STAMP(46);                                     // (diagnostic: (A3) of the last substep)
        fused_prefetch(late_ctx(C), xs, &xbias[0][0], bid, n, htid, fstep, sink.values, fids, ro);
#ifdef LG_STAMPS
        __builtin_amdgcn_s_waitcnt(0);
#endif
        STAMP(47);
      }
    }
    float nz[NZ_IT][4];
    const bool predraw = fuse && fused_noise_predrawn(hot);
    if (predraw) fused_noise_draw(hot, bid, n, htid, fstep, nz, fids, ro);   // (these waves would wait for the main wave's last sweeps now)
    FusedPre fpre;
    if (fuse) fused_prefetch_static(late_ctx(C), hot, htid, fpre);
    lds_barrier();                                     // (F) main wave has published the final state of the step
    const DevCtx* const Ct = late_ctx(C);                // (everything behind the last substep reads the context through this: see late_ctx)
    STAMP(48);
    if (predraw) fused_noise_park(hot, cst, bid, n, htid, nz);
    if (fuse) fused_stage_obs_table(hot, cst, htid, fpre);
    if (TMESH && valid) mesh_cache_io<false>(Ct, cqc, e, l, lane, MESH_PAIR0(wv));
    bool zero_state = false;
    if (!fuse) {
      if (valid) {                                       // wave w stores link w-1 of every leg (+ base / + foot body)
        float r13[13], qq[3], qdd[3];
        fetch_state(xst[lane], r13, qq, qdd);
        write_rigid_body_state(Ct, lm_, e, l, r13, qq, qdd, wv - 1);
      }
    } else {
      // fused step: first what the rest of the tail waits for (the height scan; the feet rows when a reward term reads them),
      // the rigid-body rows -- stores nobody in this launch reads -- come last
      const bool feet_early = fpre.feet_early != 0;       // (kernel-uniform)
      if (feet_early && wv == 3 && valid) {
        if (fpre.inject) {                               // parity tests: the feet rows the caller injected
          const float* o = fpre.rigid + ((size_t)e * fpre.B + 1 + fpre.per_leg * l + (fpre.per_leg == 4 ? 3 : 2)) * 13;
          float* fr = fused_foot_row(xs, lane);
#pragma unroll
          for (int i = 0; i < 13; ++i) fr[i] = o[i];
        } else {
          float r13[13], qq[3], qdd[3];
          fetch_state(xst[lane], r13, qq, qdd);
          write_rigid_body_state(Ct, lm_, e, l, r13, qq, qdd, 2, fused_foot_row(xs, lane), nullptr, fpre.per_leg, fpre.B, fpre.rigid);   // (gait_foot_z is stored late: the serial part still reads the old one)
        }
      }
      fused_height_scan(Ct, xst, cst, bid, n, htid, fids, ro, fpre);
      STAMP(35);                                       // (diagnostic)
#if defined(LG_STAMPS) && defined(LG_SCAN_TWICE)
      { const DevCtx* Cx = Ct; asm volatile("" : "+s"(Cx));     // (diagnostic: the same code a second time, now warm in the instruction cache)
        fused_height_scan(Cx, xst, cst, bid, n, htid, fids, ro, fpre); }
      STAMP(27);
#endif
      // the rigid-body rows (stores nobody in this launch reads) while the main wave runs the serial part: these waves wait ~6 k cycles for
      // it at (G2); behind the write-back, where they used to be, they were on the tail of the launch
      if (valid && fpre.inject) {                                // parity tests: the injected rows stay; gait_foot_z from the injected foot row
        if (wv == 3 && fpre.gait_on && !ro) fpre.gfz[(size_t)e * NLEG + l] = fpre.rigid[((size_t)e * fpre.B + 1 + fpre.per_leg * l + (fpre.per_leg == 4 ? 3 : 2)) * 13 + 2];
      } else if (valid && !(feet_early && wv == 3)) {            // rigid-body rows of the post-physics (pre-reset) pose, LR:118-120
        float r13[13], qq[3], qdd[3];
        fetch_state(xst[lane], r13, qq, qdd);
        write_rigid_body_state(Ct, lm_, e, l, r13, qq, qdd, wv - 1, nullptr, (wv == 3 && fpre.gait_on && !ro) ? fpre.gfz + (size_t)e * NLEG + l : nullptr,
                               fpre.per_leg, fpre.B, fpre.rigid);
      } else if (valid && fpre.gait_on && !ro) {
        fpre.gfz[(size_t)e * NLEG + l] = fused_foot_row(xs, lane)[2];
      }
      // the LSTM state, action and torque rows while the main wave is still in the serial part (these waves wait for it at (G2)); what a reset
      // changes -- a zero LSTM state, anymal.py:78-82 -- is stored over it behind (G2)
      if (valid && net) {
        store_lstm_rows(Ct, row, N12, h0, c0, h1, c1);
        fpre.act[(size_t)e * NDOF + d] = a;
        if (!fpre.inject) fpre.tq[(size_t)e * NDOF + d] = xtau[j][lane];
      }
      STAMP(51);
      if (feet_early || fpre.heights_early) lds_barrier();   // (G1) only when the serial part reads a helper's product
      lds_barrier();                                   // (G2) serial part + height scan done
      STAMP(52);
      zero_state = fused_did_reset(cst, lane / GRP);    // anymal.py:78-82: a reset env starts from the zero LSTM state
    }
    if (valid && net && (!fuse || zero_state)) {         // (fused step: stored in front of (G2) already; a reset env's rows are overwritten with zeros here --
      if (zero_state) {                                  //  same lane, same addresses: the stores of one wave stay in order)
#pragma unroll
        for (int i = 0; i < 8; ++i) { h0[i] = 0.f; c0[i] = 0.f; h1[i] = 0.f; c1[i] = 0.f; }
      }
      store_lstm_rows(Ct, row, N12, h0, c0, h1, c1);
    }
    if (valid && net && !fuse) {
      Ct->actions[(size_t)e * NDOF + d] = a;
      Ct->torques[(size_t)e * NDOF + d] = xtau[j][lane];
    }
    if (fuse) {
      const bool arrive = !PERSIST || st + 1 >= nsteps;  // (the launch's last step counts the workgroup's arrival)
      const bool last = fused_writeback_obs(Ct, hot, xs, cst, bid, n, tid_it, fstep, nullptr, sink.obs_out, sink.obs_by_row != 0, fids, ro, arrive);
      if (!arrive) { __syncthreads(); fstep += 1; continue; }     // (the step's rows are stored and visible to the workgroup; its LDS is free again)
      if (tid_it == 64 * FUSED_STATS_WAVE) s_last_f = last ? 1 : 0;
      __syncthreads();
      if (s_last_f) fused_finalize(Ct, gridDim.x, tid_it, ro, nsteps);
    }
    if (GFUSABLE && gfuse) {                           // (every row this workgroup's envs own is stored; the barrier makes them visible to its four waves)
      __syncthreads();
      generic_fused_tail(Ct, cst, ids, n, ro ? 1 : 0, sink);
    }
    return;
    }
  }
  const bool helpers = MODE == 0 && HELPERS;             // helper waves present (leg bias + contact detection); the host launches this instance with nact == 3
  const bool split = helpers && net;                     // ... and they evaluate the actuator network too

  QuadState s;
#pragma unroll
  for (int i = 0; i < 13; ++i) s.root[i] = pre_root[i];
  float last_qd[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) { s.q[j] = pre_dof[2 * j]; s.qd[j] = pre_dof[2 * j + 1]; last_qd[j] = pre_lqd[j]; }
  float act[3] = {0, 0, 0};
  if (MODE != 1 && !split) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float a = actions_in ? actions_in[(size_t)krow * act_stride + 3 * l + j] : C->actions[(size_t)e * NDOF + 3 * l + j];
      a = fminf(fmaxf(a, -g.clip_actions), g.clip_actions);        // LR:93-94
      act[j] = a;
      if (valid && actions_in) C->actions[(size_t)e * NDOF + 3 * l + j] = a;
    }
  }
  LegActuator A;
  if (MODE != 1 && net && !split) load_lstm(C, e, l, A);

  float tau[3] = {0, 0, 0};
  if (MODE == 2) {
    leg_torques(C, lm_, wlstm, act, s.q, s.qd, last_qd, A, tau);
    if (valid) {
#pragma unroll
      for (int j = 0; j < 3; ++j) C->torques[(size_t)e * NDOF + 3 * l + j] = tau[j];
      if (net) store_lstm(C, e, l, A);
    }
    return;
  }

#pragma unroll 1
  for (int st = 0;; ++st) {                              // (one pass unless PERSIST)
  const DevCtx* __restrict__ const C = PERSIST ? late_ctx(Cq) : Cq;      // (see the helper waves' step loop)
  const lg_config& g = C->cfg;
  const lg_robot_model* __restrict__ m = &C->model;
  const int e = opaque_v(e_q), krow = opaque_v(krow_q), bid = opaque_s(bid_q), tid_it = opaque_v((int)threadIdx.x);
  PhysParams P;
  P.dt = g.sim_dt; P.grav = v3(g.gravity[0], g.gravity[1], g.gravity[2]); P.iters = g.solver_iterations;
  P.contact_offset = g.contact_offset; P.max_depen = g.max_depenetration_velocity; P.erp = g.erp; P.cfm = g.cfm; P.solver = g.solver_type; P.fric = g.friction_model;
  P.terrain_mu = C->terrain_mu; P.slide_mask = (CAPS || (MCAPS && C->ter.SEG4)) ? C->slide_mask : 0u; P.slot_perm = CAPS ? C->slot_perm : (TMESH ? C->mesh_perm : 0x76543210u); P.cache_reach = TMESH ? C->mesh_reach : 0.f;
#if LG_AB == 21
  P.slide_mask = 0u;
#endif
  const TerrainView T = C->ter;
  const SelfCol scol{C->sc_pairs, (FEAT & 2) ? C->n_sc : 0, SC_LDS ? sctab : C->sc_tab, ((FEAT & 2) && helpers) ? reinterpret_cast<unsigned*>(&xst[0][0]) : nullptr};
  const float mu_robot = pre_mu, madd = pre_madd;
  V3 fbody[5];
  bool fault = false;
  if (TMESH && helpers && st == 0) mesh_cache_io<true>(C, cqc, e, l, lane, MESH_PAIR0(0));
#ifdef LG_STAMPS
  unsigned long long* stamps = (blockIdx.x == 0 && lane == 0) ? C->stamps : nullptr;
  __builtin_amdgcn_s_waitcnt(0);                      // (diagnostic: the state loads have landed)
  unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
  if (stamps) { stamps[37] += t_bar0 - t_entry; stamps[38] += stamp_t - t_bar0; }
#else
  unsigned long long* stamps = nullptr;
#endif
  PostSink sk = sink;
  if (PERSIST && st > 0) {                               // the next rollout step of this launch: state in registers, this step's action rows and reward column
    if (!split) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float a = actions_in[(size_t)krow * act_stride + (size_t)st * NDOF + 3 * l + j];
        a = fminf(fmaxf(a, -g.clip_actions), g.clip_actions);
        act[j] = a;
        if (valid) C->actions[(size_t)e * NDOF + 3 * l + j] = a;
      }
    }
    // the state the previous step ended with, from the env rows its tail left in LDS (what it stored to the state tensors): re-read rather than kept,
    // so that these 19 registers are free during the tail -- the kernel has none to spare there
    fused_reload_state(xs, lane, l, s.root, s.q, s.qd);
#pragma unroll
    for (int j = 0; j < 3; ++j) { last_qd[j] = s.qd[j]; tau[j] = 0.f; }      // (what the tail stored as last_dof_vel: the joint speeds the step ended with)
    fault = false;
    if (sink.rew_out) sk.rew_out = sink.rew_out + st;
  }
#pragma unroll 1
  for (int sub = 0; sub < nsub; ++sub) {
    STAMP(15);
    if (helpers) {
      publish_state(xst[lane], s.root, s.q, s.qd);
      STAMP(24);                                       // (diagnostic)
      // Measured and dropped (round 4): the first substep without this rendezvous (the helper waves load the rows themselves).  The main wave
      // waits 6.5 k cycles here in the first substep of a launch -- the helper waves' LSTM state loads and first recurrent half, all of it cold
      // -- but without it it waits as long at (A2) instead (their first substep takes 17 k cycles against 12 k later on): +1 % on the step.
      lds_barrier();                                   // (A) root, q, qd of this substep visible to the helper waves
      if (sub == 0) STAMP(26); else STAMP(25);         // (diagnostic: the first substep's rendezvous apart)
    }
    if (split) {
    } else if (MODE == 0) {
      leg_torques<!HELPERS>(C, lm_, wlstm, act, s.q, s.qd, last_qd, A, tau);
    } else {
#pragma unroll
      for (int j = 0; j < 3; ++j) tau[j] = C->torques[(size_t)e * NDOF + 3 * l + j];
    }
    float root0[7], q0[3];
#pragma unroll
    for (int i = 0; i < 7; ++i) root0[i] = s.root[i];
#pragma unroll
    for (int j = 0; j < 3; ++j) q0[j] = s.q[j];
    STAMP(0);
    auto tau_fn = [&](float* t) {
      if (split) {                                       // written by the helper waves before barrier (A2)
#pragma unroll
        for (int j = 0; j < 3; ++j) tau[j] = xtau[j][lane];
      }
      t[0] = tau[0]; t[1] = tau[1]; t[2] = tau[2];
    };
    auto prep_fn = [&](float* bk, V3& Fs, V3& Ns) -> bool {
      if (!helpers) return false;
      lds_barrier();                                   // (A2) helper waves have written the leg bias and the slot table
      const float4* pb4 = reinterpret_cast<const float4*>(xbias[lane]);
      const float4 b0 = pb4[0], b1 = pb4[1], b2 = pb4[2];
      bk[0] = b0.x; bk[1] = b0.y; bk[2] = b0.z;
      Fs = v3(b0.w, b1.x, b1.y);
      Ns = v3(b1.z, b1.w, b2.x);
      return true;
    };
    auto share_fn = [&]() { if (helpers) lds_barrier(); };   // (A3) every wave has finished its slots
    const SlotShare share{helpers ? 4 : 1, helpers ? 3 : 0, helpers};     // set-up order: wave 1, 2, 3, then this wave
    physics_substep<TMESH, TMESH ? MESH_PAIR0(0) + 100 : DS0, !(MODE == 0 && HELPERS), (SPEC & 3) == 1 ? 1 : 0, FEAT>(m, lm_, T, P, lane, cst, s, tau_fn, prep_fn, share_fn, share, xs, mu_robot, madd,
                              sub == nsub - 1 ? fbody : nullptr, stamps, (TMESH && helpers) ? cqc : nullptr, scol);
#ifdef LG_STAMPS
    stamp_t = __builtin_amdgcn_s_memtime();
#endif
    // fault guard: a non-finite or diverged state is rolled back to the pre-step pose at rest and flagged for termination
    float acc = 0.f, acc0 = 0.f;
#pragma unroll
    for (int i = 0; i < 13; ++i) acc += s.root[i] * 0.f;
#pragma unroll
    for (int i = 7; i < 13; ++i) acc += fabsf(s.root[i]) < 1e3f ? 0.f : 1.f;   // finite but diverged (> 1 km/s, > 1000 rad/s)
    if (TMESH) {
      // a base that has left the collision mesh (walked off its edge, or got through a surface and is in free fall below it)
      // can never come back: same treatment as a fault -- back to the pre-step pose at rest, episode terminated.  PhysX has
      // no such rule (its bodies fall forever); the reference's tasks never leave their terrain because it is bordered.
      acc += (s.root[0] < C->mesh_lo[0] - LG_MESH_OOB_MARGIN || s.root[0] > C->mesh_hi[0] + LG_MESH_OOB_MARGIN ||
              s.root[1] < C->mesh_lo[1] - LG_MESH_OOB_MARGIN || s.root[1] > C->mesh_hi[1] + LG_MESH_OOB_MARGIN ||
              s.root[2] < C->mesh_lo[2] - LG_MESH_OOB_MARGIN) ? 1.f : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) { acc += s.q[j] * 0.f + s.qd[j] * 0.f; acc0 += q0[j] * 0.f; }
#pragma unroll
    for (int i = 0; i < 7; ++i) acc0 += root0[i] * 0.f;
    acc = grp_sum(acc); acc0 = grp_sum(acc0);
    if (!(acc == 0.f)) {
      const bool ok0 = acc0 == 0.f;
      fault = true;
#pragma unroll
      for (int i = 0; i < 13; ++i)
        s.root[i] = ok0 ? (i < 7 ? root0[i] : 0.f) : g.base_init_state[i] + (i < 3 ? C->origins[(size_t)e * 3 + i] : 0.f);
#pragma unroll
      for (int j = 0; j < 3; ++j) { s.q[j] = ok0 ? q0[j] : lm_.f(LM_DEFAULT_POS + j); s.qd[j] = 0.f; }
#pragma unroll
      for (int b = 0; b < 5; ++b) fbody[b] = v3(0, 0, 0);
    }
  }
  STAMP(9);
  const DevCtx* const Ct = late_ctx(C);                  // (see late_ctx)
  if (fuse && g.inject_sim_state) {
    // parity tests (lg_config.inject_sim_state): the post-physics half starts from the post-simulation state the caller left in the
    // tensors -- this workgroup's rows still hold it, nothing of this launch has stored to them yet
    const int per_leg = Ct->per_leg, B = Ct->B;
    const int ee = env_ok ? e : 0;
#pragma unroll
    for (int i = 0; i < 13; ++i) s.root[i] = Ct->root[(size_t)ee * 13 + i];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      s.q[j] = Ct->dof[((size_t)ee * NDOF + 3 * l + j) * 2]; s.qd[j] = Ct->dof[((size_t)ee * NDOF + 3 * l + j) * 2 + 1];
      tau[j] = Ct->torques[(size_t)ee * NDOF + 3 * l + j];
    }
    const float* cf = Ct->cforce + (size_t)ee * B * 3;
    fbody[0] = v3(cf[0], cf[1], cf[2]);
#pragma unroll
    for (int b = 0; b < 3; ++b) { const float* cl = cf + (size_t)(1 + per_leg * l + b) * 3; fbody[1 + b] = v3(cl[0], cl[1], cl[2]); }
    if (per_leg == 4) { const float* cl = cf + (size_t)(1 + per_leg * l + 3) * 3; fbody[4] = v3(cl[0], cl[1], cl[2]); } else fbody[4] = v3(0, 0, 0);
    fault = false;
  }
  if (helpers) {
    float qn = 0.f;
    if (fuse) {      // the height scan's yaw-only quaternion (math_utils.quat_apply_yaw), normalised here once per env instead of once per scan point
      const float qz = s.root[5], qw = s.root[6];
      const float nrm = fmaxf(sqrtf(qz * qz + qw * qw), 1e-9f);
      qn = (l == 0 ? qz : qw) / nrm;
    }
    publish_state(xst[lane], s.root, s.q, s.qd, qn);
    lds_barrier();                                     // (F) final state visible to the helper waves, which write the body states
  }
  STAMP(39);
  if (fuse) {
    // ---- fused step: the post-physics step of the workgroup's envs, from the registers of this wave (lg_fused_post.h)
    fused_main_and_serial(Ct, hot, lm_, xs, &xbias[0][0], cst, lane, e, valid, s.root, s.q, s.qd, tau, last_qd, fbody, split ? nullptr : act, fault, ro ? gstep_f : fstep, stamps, sk, ro, krow);
#ifdef LG_STAMPS
    stamp_t = __builtin_amdgcn_s_memtime();
#endif
    if (valid) {
      if (TMESH && helpers) mesh_cache_io<false>(Ct, cqc, e, l, lane, MESH_PAIR0(0));
#pragma unroll
      for (int j = 0; j < 3; ++j) if (!split) Ct->torques[(size_t)e * NDOF + 3 * l + j] = tau[j];
      // (the net contact forces go out with the env rows: fused_writeback_obs, from the LDS rows fused_main_part1 wrote)
    }
    STAMP(12);
    const bool arrive = !PERSIST || st + 1 >= nsteps;
    const bool last_wg = fused_writeback_obs(Ct, hot, xs, cst, bid, n, tid_it, fstep, stamps, sink.obs_out, sink.obs_by_row != 0, fids, ro, arrive);
#ifdef LG_STAMPS
    stamp_t = __builtin_amdgcn_s_memtime();
#endif
    STAMP(13);
    if (!arrive) { __syncthreads(); fstep += 1; continue; }
    if (tid_it == 64 * FUSED_STATS_WAVE) s_last_f = last_wg ? 1 : 0;
    __syncthreads();
    if (s_last_f) fused_finalize(Ct, gridDim.x, tid_it, ro, nsteps);
    STAMP(14);
#ifdef LG_STAMPS
    if (stamps) stamps[36] += __builtin_amdgcn_s_memtime() - t_entry;      // the main wave's whole kernel
#endif
    return;
  }
  if (valid) {
  if (TMESH && helpers) mesh_cache_io<false>(C, cqc, e, l, lane, MESH_PAIR0(0));
  if (fault && l == 0) C->reset_buf[e] = 2;

  // ---- write back state, torques, contact forces
  if (l == 0) {
#pragma unroll
    for (int i = 0; i < 13; ++i) C->root[(size_t)e * 13 + i] = s.root[i];
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    C->dof[((size_t)e * NDOF + 3 * l + j) * 2] = s.q[j];
    C->dof[((size_t)e * NDOF + 3 * l + j) * 2 + 1] = s.qd[j];
    if (MODE == 0 && !split) C->torques[(size_t)e * NDOF + 3 * l + j] = tau[j];
  }
  if (MODE == 0 && net && !split) store_lstm(C, e, l, A);
  const int per_leg = C->per_leg, B = C->B;
  {
    float* cf = C->cforce + (size_t)e * B * 3;
    if (l == 0) { cf[0] = fbody[0].x; cf[1] = fbody[0].y; cf[2] = fbody[0].z; }
    float* cl = cf + (size_t)(1 + per_leg * l) * 3;
    V3 last = fbody[3];
    if (per_leg == 3) last = last + fbody[4];      // no separate foot body: its spheres report on the last link
    cl[0] = fbody[1].x; cl[1] = fbody[1].y; cl[2] = fbody[1].z;
    cl[3] = fbody[2].x; cl[4] = fbody[2].y; cl[5] = fbody[2].z;
    cl[6] = last.x; cl[7] = last.y; cl[8] = last.z;
    if (per_leg == 4) { cl[9] = fbody[4].x; cl[10] = fbody[4].y; cl[11] = fbody[4].z; }
  }
  // ---- rigid-body state of the post-step configuration (LR:118-120 refresh_rigid_body_state_tensor); with helper
  // waves, they write it (one link each) from the published final state while this wave stores the rest
  if (!helpers) write_rigid_body_state(C, lm_, e, l, s.root, s.q, s.qd);
  }
  STAMP(10);
  if (GFUSABLE && gfuse) {
    __syncthreads();
    generic_fused_tail(C, cst, ids, n, ro ? 1 : 0, sink);
  }
  return;
  }
}


#else            // NJ != 3
// Body states of one leg (+ the base from leg 0) from the generalised state: the (N, B, 13) rows the reference reads from refresh_rigid_body_state_tensor
LG_DEV void write_rigid_body_state(const DevCtx* __restrict__ C, const LegModel& lm_, int e, int l, const float* root, const float* q, const float* qd) {
  const int per_leg = C->per_leg, B = C->B;
  const M3 Rb = quat_to_mat(root + 3);
  const V3 pb = v3(root[0], root[1], root[2]), vb = v3(root[7], root[8], root[9]), wb = v3(root[10], root[11], root[12]);
  LegKin k;
  leg_kinematics(lm_, Rb, pb, vb, wb, q, qd, k);
  float* rb = C->rigid + (size_t)e * B * 13;
  if (l == 0) {
#pragma unroll
    for (int i = 0; i < 13; ++i) rb[i] = root[i];
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    float* o = rb + (size_t)(1 + per_leg * l + j) * 13;
    float qq[4]; mat_to_quat(k.R[j], qq);
    o[0] = k.O[j].x; o[1] = k.O[j].y; o[2] = k.O[j].z; o[3] = qq[0]; o[4] = qq[1]; o[5] = qq[2]; o[6] = qq[3];
    o[7] = k.vO[j].x; o[8] = k.vO[j].y; o[9] = k.vO[j].z; o[10] = k.w[j].x; o[11] = k.w[j].y; o[12] = k.w[j].z;
  }
  if (per_leg == NJ + 1) {
    float* o = rb + (size_t)(1 + per_leg * l + NJ) * 13;
    const V3 r = mul(k.R[NJ - 1], lm_.v(LM_FOOT_POS));
    const V3 p = k.O[NJ - 1] + r, v = k.vO[NJ - 1] + cross(k.w[NJ - 1], r);
    M3 fr;
#pragma unroll
    for (int i = 0; i < 9; ++i) fr.m[i] = lm_.f(LM_FOOT_ROT + i);
    float qq[4]; mat_to_quat(mul(k.R[NJ - 1], fr), qq);
    o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = qq[0]; o[4] = qq[1]; o[5] = qq[2]; o[6] = qq[3];
    o[7] = v.x; o[8] = v.y; o[9] = v.z; o[10] = k.w[NJ - 1].x; o[11] = k.w[NJ - 1].y; o[12] = k.w[NJ - 1].z;
  }
}

// The physics of a policy step (MODE 0: clip actions, nsub x (PD torques + one dt)), one dt with the torques of LG_T_TORQUES (MODE 1: lg_simulate), or the
// torques alone (MODE 2: lg_compute_torques) for the chain instance: ONE wave per workgroup, a lane per leg, EPW envs per wave; post_kernel ends the step.
// TMESH: contacts against a grid mesh (closest-point queries by cell index; the instance has no BVH walk).
// HELP (round 5): a 256-thread launch whose waves 1-3 detect the contact slots -- kinematics of the published state, then their slots' queries, on a grid mesh
// ~10 k cycles each -- while the main wave runs the bias, the mass matrix and its factorisation (Cassie, 4096 envs, trimesh: 0.281 -> ms per step, see the A/B log).
template <int MODE, bool TMESH, bool HELP = false>
__global__ __launch_bounds__(HELP ? 256 : 64) void physics_kernel_chain(const DevCtx* __restrict__ C, const float* __restrict__ actions_in, int nsub, const int32_t* __restrict__ ids, int n, int act_stride,
                                                           int epb) {
  // epb: envs this workgroup steps (<= EPW; chain_epb).  The kernel is a chain of dependent latencies, one workgroup per CU (86 KB of LDS): a launch of few
  // workgroups leaves CUs idle, so the host deals the envs over up to 256 of them and the lanes past epb groups compute on a copy and store nothing.
  __shared__ __attribute__((aligned(16))) float cst[CH_CST_FLOATS];
  __shared__ float lmod[LM_FIELDS * GRP];
  constexpr int XST = 13 + 2 * NJ;                       // state a helper wave needs, per lane: root 13 | q NJ | qd NJ (an odd stride: conflict-free)
  __shared__ float xst[HELP ? 64 * XST : 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // HALVES (the helper-wave instance at <= 16 envs per workgroup, i.e. N <= 4096): the upper 32 lanes of every wave would be idle copies; instead they MIRROR
  // the lower 32 -- same env, same leg, same arithmetic, same slot-record rows -- and the work that is per contact slot is split between the halves:
  // a helper wave detects two slots at once, the main wave sets up the even slots on its lower and the odd slots on its upper half.
  const bool halves = HELP && epb * GRP <= 32;
  const int le = halves ? (lane & 31) : lane;              // the lane's row in the slot records and in the published state
  const int half = halves ? (lane >> 5) : -1;
  const int kq = blockIdx.x * epb + le / GRP;
  const int l = lane % GRP;
  const bool live = kq < n && le / GRP < epb;
  const bool valid = live && half <= 0;                   // (the lower half stores)
  const int krow = live ? kq : n - 1;
  const int e = ids ? ids[krow] : krow;
  const lg_robot_model* __restrict__ m = &C->model;
  const lg_config& g = C->cfg;
  fill_leg_model(lmod, C->lmod, threadIdx.x, blockDim.x);
  lds_barrier();
  const LegModel lm_{lmod, l};
  if (HELP && wv > 0) {
    // ---- helper wave: per substep, the kinematics of the state the main wave published and the detection of this wave's slots
    PhysParams P;
    P.dt = g.sim_dt; P.grav = v3(g.gravity[0], g.gravity[1], g.gravity[2]); P.iters = g.solver_iterations;
    P.contact_offset = g.contact_offset; P.max_depen = g.max_depenetration_velocity; P.erp = g.erp; P.cfm = g.cfm; P.solver = g.solver_type; P.fric = g.friction_model;
    P.terrain_mu = C->terrain_mu; P.slide_mask = C->slide_mask; P.slot_perm = 0x76543210u; P.cache_reach = LG_MESH_CACHE_REACH;
    const TerrainView T = C->ter;
#pragma unroll 1
    for (int sub = 0; sub < nsub; ++sub) {
      lds_barrier();                                     // (A) the main wave has published root, q, qd of this substep
      float r13[13], qq[NJ], qdd[NJ];
      // (halves: a helper wave's two halves detect two slots of the same 32 rows at once -- the four slots take ONE slot's time on three waves instead of two)
      const float* x = xst + le * XST;
#pragma unroll
      for (int i = 0; i < 13; ++i) r13[i] = x[i];
#pragma unroll
      for (int j = 0; j < NJ; ++j) { qq[j] = x[13 + j]; qdd[j] = x[13 + NJ + j]; }
      const M3 Rb = quat_to_mat(r13 + 3);
      const V3 pb = v3(r13[0], r13[1], r13[2]), vb = v3(r13[7], r13[8], r13[9]), wb = v3(r13[10], r13[11], r13[12]);
      LegKin k;
      leg_kinematics(lm_, Rb, pb, vb, wb, qq, qdd, k);
      // slots dealt round-robin over the three helper waves (CH_NCP = 4: wave 1 takes slots 0 and 3)
      if (halves) {
        const int sl = lane < 32 ? wv - 1 : wv + 2;
        if (sl < CH_NCP) ch_detect_slot<TMESH>(sl, lm_, T, P, k, Rb, pb, cst, le);
      } else {
#pragma unroll 1
        for (int sl = wv - 1; sl < CH_NCP; sl += 3) ch_detect_slot<TMESH>(sl, lm_, T, P, k, Rb, pb, cst, lane);
      }
      lds_barrier();                                     // (A2) detection blocks complete
    }
    return;
  }
  QuadState s;
#pragma unroll
  for (int i = 0; i < 13; ++i) s.root[i] = C->root[(size_t)e * 13 + i];
  float last_qd[NJ], act[NJ], tau[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    s.q[j] = C->dof[((size_t)e * NDOF + NJ * l + j) * 2]; s.qd[j] = C->dof[((size_t)e * NDOF + NJ * l + j) * 2 + 1];
    last_qd[j] = C->last_dof_vel[(size_t)e * NDOF + NJ * l + j];
    act[j] = 0.f; tau[j] = 0.f;
  }
  if (MODE != 1) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      float a = actions_in ? actions_in[(size_t)krow * act_stride + NJ * l + j] : C->actions[(size_t)e * NDOF + NJ * l + j];
      a = fminf(fmaxf(a, -g.clip_actions), g.clip_actions);        // LR:93-94
      act[j] = a;
      if (valid && actions_in) C->actions[(size_t)e * NDOF + NJ * l + j] = a;
    }
  }
  if (MODE == 2) {
    ch_leg_torques(g, lm_, act, s.q, s.qd, last_qd, tau);
    if (valid) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) C->torques[(size_t)e * NDOF + NJ * l + j] = tau[j];
    }
    return;
  }
  PhysParams P;
  P.dt = g.sim_dt; P.grav = v3(g.gravity[0], g.gravity[1], g.gravity[2]); P.iters = g.solver_iterations;
  P.contact_offset = g.contact_offset; P.max_depen = g.max_depenetration_velocity; P.erp = g.erp; P.cfm = g.cfm; P.solver = g.solver_type; P.fric = g.friction_model;
  P.terrain_mu = C->terrain_mu; P.slide_mask = C->slide_mask; P.slot_perm = 0x76543210u; P.cache_reach = LG_MESH_CACHE_REACH;
  const TerrainView T = C->ter;
  const SelfCol scol{C->sc_pairs, C->n_sc, nullptr, nullptr};
  const float mu_robot = C->friction[e], madd = C->mass_added[e];
  V3 fbody[NJ + 2];
#pragma unroll
  for (int b = 0; b < NJ + 2; ++b) fbody[b] = v3(0, 0, 0);
  bool fault = false;
#pragma unroll 1
  for (int sub = 0; sub < nsub; ++sub) {
    if (MODE == 0) ch_leg_torques(g, lm_, act, s.q, s.qd, last_qd, tau);
    else {
#pragma unroll
      for (int j = 0; j < NJ; ++j) tau[j] = C->torques[(size_t)e * NDOF + NJ * l + j];
    }
    float root0[7], q0[NJ];
#pragma unroll
    for (int i = 0; i < 7; ++i) root0[i] = s.root[i];
#pragma unroll
    for (int j = 0; j < NJ; ++j) q0[j] = s.q[j];
    if (HELP) {
      float* x = xst + le * XST;                           // (halves: both mirrors write the same values)
#pragma unroll
      for (int i = 0; i < 13; ++i) x[i] = s.root[i];
#pragma unroll
      for (int j = 0; j < NJ; ++j) { x[13 + j] = s.q[j]; x[13 + NJ + j] = s.qd[j]; }
      lds_barrier();                                     // (A)
    }
    chain_substep<TMESH, HELP>(m, lm_, T, P, le, cst, s, tau, mu_robot, madd, sub == nsub - 1 ? fbody : nullptr, scol, half);
    // fault guard: a non-finite or diverged state is rolled back to the pre-step pose at rest and flagged for termination
    float acc = 0.f, acc0 = 0.f;
#pragma unroll
    for (int i = 0; i < 13; ++i) acc += s.root[i] * 0.f;
#pragma unroll
    for (int i = 7; i < 13; ++i) acc += fabsf(s.root[i]) < 1e3f ? 0.f : 1.f;
    if (TMESH) {
      acc += (s.root[0] < C->mesh_lo[0] - LG_MESH_OOB_MARGIN || s.root[0] > C->mesh_hi[0] + LG_MESH_OOB_MARGIN ||
              s.root[1] < C->mesh_lo[1] - LG_MESH_OOB_MARGIN || s.root[1] > C->mesh_hi[1] + LG_MESH_OOB_MARGIN ||
              s.root[2] < C->mesh_lo[2] - LG_MESH_OOB_MARGIN) ? 1.f : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) { acc += s.q[j] * 0.f + s.qd[j] * 0.f; acc0 += q0[j] * 0.f; }
#pragma unroll
    for (int i = 0; i < 7; ++i) acc0 += root0[i] * 0.f;
    acc = grp_sum(acc); acc0 = grp_sum(acc0);
    if (!(acc == 0.f)) {
      const bool ok0 = acc0 == 0.f;
      fault = true;
#pragma unroll
      for (int i = 0; i < 13; ++i)
        s.root[i] = ok0 ? (i < 7 ? root0[i] : 0.f) : g.base_init_state[i] + (i < 3 ? C->origins[(size_t)e * 3 + i] : 0.f);
#pragma unroll
      for (int j = 0; j < NJ; ++j) { s.q[j] = ok0 ? q0[j] : lm_.f(LM_DEFAULT_POS + j); s.qd[j] = 0.f; }
#pragma unroll
      for (int b = 0; b < NJ + 2; ++b) fbody[b] = v3(0, 0, 0);
    }
  }
  if (!valid) return;
  if (fault && l == 0) C->reset_buf[e] = 2;
  if (l == 0) {
#pragma unroll
    for (int i = 0; i < 13; ++i) C->root[(size_t)e * 13 + i] = s.root[i];
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    C->dof[((size_t)e * NDOF + NJ * l + j) * 2] = s.q[j];
    C->dof[((size_t)e * NDOF + NJ * l + j) * 2 + 1] = s.qd[j];
    if (MODE == 0) C->torques[(size_t)e * NDOF + NJ * l + j] = tau[j];
  }
  const int per_leg = C->per_leg, B = C->B;
  {
    float* cf = C->cforce + (size_t)e * B * 3;
    if (l == 0) { cf[0] = fbody[0].x; cf[1] = fbody[0].y; cf[2] = fbody[0].z; }
    float* cl = cf + (size_t)(1 + per_leg * l) * 3;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      V3 f = fbody[1 + j];
      if (j == NJ - 1 && per_leg == NJ) f = f + fbody[NJ + 1];      // no separate foot body: its spheres report on the last link
      cl[3 * j] = f.x; cl[3 * j + 1] = f.y; cl[3 * j + 2] = f.z;
    }
    if (per_leg == NJ + 1) { cl[3 * NJ] = fbody[NJ + 1].x; cl[3 * NJ + 1] = fbody[NJ + 1].y; cl[3 * NJ + 2] = fbody[NJ + 1].z; }
  }
  write_rigid_body_state(C, lm_, e, l, s.root, s.q, s.qd);
}
#endif           // NJ == 3

// ============================================================================================ post-physics helpers
LG_DEV float wrap_to_pi(float a) {   // math_utils.py:55-58
  const float two_pi = 6.28318530717958647692f, pi = 3.14159265358979323846f;
  float r = fmodf(a, two_pi);
  if (r != 0.f && r < 0.f) r += two_pi;
  if (r > pi) r -= two_pi;
  return r;
}

// Where one env's data lives while the post-physics logic runs: LDS copies inside post_kernel (staged with coalesced
// loads, so the one-lane-per-env phase never waits on global memory), plain global rows for lg_reset_idx.
struct EnvView {
  float *root, *dof, *cmd, *air, *ctime, *blv, *bav, *pg;
  const float *tq, *act, *lact, *ldv, *cf, *rb, *bla;
  uint8_t* lastc;
  int feet_rows = 0;      // rb holds only the feet rows (foot f at rb + 13 f) instead of all bodies: the fused step's rows
};
LG_DEV EnvView global_view(const DevCtx* __restrict__ C, int e) {
  EnvView V;
  V.root = C->root + (size_t)e * 13; V.dof = C->dof + (size_t)e * 2 * NDOF; V.cmd = C->commands + (size_t)e * 4;
  V.air = C->feet_air + (size_t)e * NLEG; V.ctime = C->feet_ctime + (size_t)e * NLEG;
  V.blv = C->base_lin_vel + (size_t)e * 3; V.bav = C->base_ang_vel + (size_t)e * 3; V.pg = C->proj_grav + (size_t)e * 3;
  V.tq = C->torques + (size_t)e * NDOF; V.act = C->actions + (size_t)e * NDOF; V.lact = C->last_actions + (size_t)e * NDOF;
  V.ldv = C->last_dof_vel + (size_t)e * NDOF; V.cf = C->cforce + (size_t)e * C->B * 3; V.rb = C->rigid + (size_t)e * C->B * 13;
  V.lastc = C->last_contacts + (size_t)e * NLEG; V.bla = C->base_lin_acc + (size_t)e * 3;
  return V;
}

// `U` = this env's uniforms for slots 0 .. LG_RS_NOISE-1 (drawn up front, one Philox call per group of four)
LG_DEV void resample_commands(const DevCtx* __restrict__ C, float* cmd, const float* U, int slot0) {  // LR:405-423
  const lg_config& g = C->cfg;
  const float* R = C->cmd_ranges;   // rows lin_vel_x, lin_vel_y, ang_vel_yaw, heading
  float c0 = rand_float(R[0], R[1], U[slot0]);
  float c1 = rand_float(R[2], R[3], U[slot0 + 1]);
  float u2 = U[slot0 + 2];
  if (g.heading_command) cmd[3] = rand_float(R[6], R[7], u2);
  else cmd[2] = rand_float(R[4], R[5], u2);
  float keep = (g.keep_small_commands || sqrtf(c0 * c0 + c1 * c1) > 0.2f) ? 1.f : 0.f;
  cmd[0] = c0 * keep; cmd[1] = c1 * keep;
}

// LR:900-938 + math_utils.quat_apply_yaw: the index arithmetic must round exactly like the reference's separate
// fp32 torch kernels, so contraction into FMAs is disabled for this function.
#pragma clang fp contract(off)
// The scan takes min(H[i][j], H[i+1][j], H[i][j+1]) (LR:929-936): the table TerrainView::Hmin holds that minimum per cell, so a probe is ONE
// 2-byte gather.  As three gathers per probe the scan of a fused workgroup was 16 x 187 x 3 = 8 976 scattered 2-byte reads -- one L1 request
// each: the texture path does not merge them -- i.e. ~9 k cycles of the CU's L1 for 6 k cycles of arithmetic.
struct HeightProbe { int16_t h; };
LG_DEV HeightProbe terrain_height_probe(const DevCtx* __restrict__ C, float qz, float qw, float px0, float py0, float bx, float by) {
  float tx = (0.f - qz * by) * 2.f, ty = (qz * bx - 0.f) * 2.f;
  float rx = (bx + qw * tx) + (0.f - qz * ty);
  float ry = (by + qw * ty) + (qz * tx - 0.f);
  // (points + border) / horizontal_scale, then .long(): only the INTEGER PART of the correctly rounded quotient is used.  A product with the
  // rounded reciprocal is within 2.5 ulp of it, so both truncate alike unless an integer lies within that distance: the two IEEE divisions
  // (v_div_scale / v_rcp / five FMAs / v_div_fmas / v_div_fixup each, a serial chain) are taken only by wave rounds in which some lane is that
  // close to an integer -- a few per cent of them; the result is the reference's bit for bit either way.
  const float fx = (rx + px0) + C->ter.border, fy = (ry + py0) + C->ter.border;
  const float ih = 1.0f / C->ter.hscale;
  float px = fx * ih, py = fy * ih;
  const bool near_int = fabsf(px - rintf(px)) <= 8e-7f * fmaxf(fabsf(px), 1.f) || fabsf(py - rintf(py)) <= 8e-7f * fmaxf(fabsf(py), 1.f);
  if (__any(near_int)) { px = fx / C->ter.hscale; py = fy / C->ter.hscale; }
  int ix = (int)px, iy = (int)py;                       // trunc toward zero (tensor.long())
  ix = max(0, min(ix, C->ter.rows - 2)); iy = max(0, min(iy, C->ter.cols - 2));
  HeightProbe r; r.h = C->ter.Hmin[(size_t)ix * C->ter.cols + iy];                   // (plane: the 1 x 1 dummy grid)
  return r;
}
LG_DEV float terrain_height_value(const DevCtx* __restrict__ C, const HeightProbe& r) { return (float)r.h * C->ter.vscale; }
LG_DEV float terrain_height_at(const DevCtx* __restrict__ C, float qz, float qw, float px0, float py0, float bx, float by) {
  return terrain_height_value(C, terrain_height_probe(C, qz, qw, px0, py0, bx, by));
}
#pragma clang fp contract(fast)

// zero_sea: the caller is lg_reset_idx's kernel, which has no write-back stage of its own: clear the LSTM actuator state rows
// and the history rows (last_actions, last_dof_vel) in global memory here.  The post kernel passes false: its write-back
// stage stores both rows for every env (LR:148-150 runs after reset_idx) and clears the LSTM rows cooperatively; storing
// them here as well would race with those stores, which another wave of the workgroup issues.
LG_DEV void reset_env(const DevCtx* __restrict__ C, const EnvView& V, int e, int update_curriculum, const float* U, bool zero_sea) {  // LR:162-213
  const lg_config& g = C->cfg;
  float* root = V.root; float* dof = V.dof; float* cmd = V.cmd;
  float* org = C->origins + (size_t)e * 3;
  float o0 = org[0], o1 = org[1], o2 = org[2];
  if (g.curriculum && update_curriculum) {   // LR:498-518
    int64_t typ = C->types[e];
    float dx = root[0] - o0, dy = root[1] - o1;
    float dist = sqrtf(dx * dx + dy * dy);
    bool up = dist > C->env_length / 2;
    bool down = (dist < sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]) * g.max_episode_length_s * 0.5f) && !up;
    int64_t L = C->levels[e] + (up ? 1 : 0) - (down ? 1 : 0);
    if (L >= g.max_terrain_level) L = (int64_t)floorf(U[LG_RS_LEVEL] * (float)g.max_terrain_level);
    else if (L < 0) L = 0;
    C->levels[e] = L;
    const float* to = C->terrain_origins + ((size_t)L * C->num_types + typ) * 3;
    o0 = to[0]; o1 = to[1]; o2 = to[2];
    org[0] = o0; org[1] = o1; org[2] = o2;
  }
  for (int d = 0; d < NDOF; ++d) {   // LR:450-465
    dof[2 * d] = g.default_dof_pos[d] * rand_float(0.5f, 1.5f, U[LG_RS_DOF + d]);
    dof[2 * d + 1] = 0.f;
  }
  float r[13];
  for (int i = 0; i < 13; ++i) r[i] = g.base_init_state[i];   // LR:467-489
  r[0] += o0; r[1] += o1; r[2] += o2;
  if (g.custom_origins) {
    r[0] += rand_float(-0.5f, 0.5f, U[LG_RS_ROOT_XY]); r[1] += rand_float(-0.5f, 0.5f, U[LG_RS_ROOT_XY + 1]);
    if (g.reset_z_from_terrain && C->ter.mesh_type != LG_MESH_PLANE) {   // robot_batch_rollout.py:1379-1391
      const TerrainView& T = C->ter;
      int ix = (int)((r[0] + T.border) / T.hscale), iy = (int)((r[1] + T.border) / T.hscale);   // trunc, as .long()
      ix = max(0, min(ix, T.rows - 2)); iy = max(0, min(iy, T.cols - 2));
      {
#pragma clang fp contract(off)   // two roundings, as torch (heights * vertical_scale, then + init z)
        const float hz = (float)T.H[(size_t)ix * T.cols + iy] * T.vscale;
        r[2] = hz + g.base_init_state[2];
      }
    }
  }
  for (int i = 0; i < 6; ++i) r[7 + i] = rand_float(-0.5f, 0.5f, U[LG_RS_ROOT_VEL + i]);
  for (int i = 0; i < 13; ++i) root[i] = r[i];
  resample_commands(C, cmd, U, LG_RS_CMD_RESET);
  if (zero_sea) for (int d = 0; d < NDOF; ++d) { C->last_actions[(size_t)e * NDOF + d] = 0.f; C->last_dof_vel[(size_t)e * NDOF + d] = 0.f; }
  for (int f = 0; f < NLEG; ++f) { V.air[f] = 0.f; V.ctime[f] = 0.f; }
  C->ep_len[e] = 0;
  C->reset_buf[e] = 1;
  if (zero_sea && g.control_type == LG_CTRL_ACTUATOR_NET) {   // anymal.py:78-82
    const size_t N12 = (size_t)C->N * NDOF;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int lay = 0; lay < 2; ++lay) {
      float4* ph = (float4*)(C->sea_h + (lay * N12 + (size_t)e * NDOF) * 8);
      float4* pc = (float4*)(C->sea_c + (lay * N12 + (size_t)e * NDOF) * 8);
      for (int k = 0; k < 2 * NDOF; ++k) { ph[k] = z4; pc[k] = z4; }
    }
  }
}

// every _reward_* of RM:41-234 (+ anymal.py:112-114), selected by id
// `feat` = the eight per-DOF sums (F_* below), `fn` = contact-force norm of every body, `bh` = sum over the height points
// of (base z - height): computed lane-parallel by the post kernel before the terms are evaluated in config order.
enum { F_TQ2 = 0, F_QD2, F_ACC2, F_ARATE2, F_POSLIM, F_VELLIM, F_TQLIM, F_STILL, F_COUNT };
// AsyncGaitScheduler's three terms with the stage's weights (lg_config.async_*; anymal_c_batch_rollout.py:207-220 over
// utils/gait_scheduler.py:151-175).  dof: (12, 2) rows [pos, vel].  A rare term: its parameters are read from the config in global memory.
LG_DEV float async_gait_value(const lg_config& g, const float* dof) {
  float align = 0.f;
  for (int k = 0; k < g.async_num_dof_sets; ++k) {
    int n = 0; float mean = 0.f;
    for (int i = 0; i < 3; ++i) { const int d = g.async_dof_sets[k][i]; if (d >= 0) { mean += dof[2 * d]; ++n; } }
    if (n < 2) continue;
    mean /= (float)n;
    float ss = 0.f;
    for (int i = 0; i < 3; ++i) { const int d = g.async_dof_sets[k][i]; if (d >= 0) { const float x = dof[2 * d] - mean; ss += x * x; } }
    align += sqrtf(ss / (float)(n - 1));
  }
  float nominal = 0.f;
  for (int d = 0; d < NDOF; ++d) { const float x = dof[2 * d] - g.async_dof_nominal[d]; nominal += x * x * g.async_dof_weight[d]; }
  return align * g.async_weights[0] + nominal * g.async_weights[1] + g.async_foot_z_align * g.async_weights[2];
}
LG_DEV float reward_term(const DevCtx* __restrict__ C, const EnvView& V, int e, int id, const float* feat, const float* fn, float bh,
                         int64_t step) {
  const lg_config& g = C->cfg; const lg_robot_model& m = C->model; const float dt = g.sim_dt * g.decimation;
  const float* root = V.root;
  const float* blv = V.blv; const float* bav = V.bav; const float* pg = V.pg; const float* cmd = V.cmd;
  const float* cf = V.cf; const float* rb = V.rb;
  float* air = V.air; float* ctime = V.ctime; uint8_t* lastc = V.lastc;
  const float cmdn = sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]);
  const bool stand = g.reward_class == LG_RC_STAND;
#define SQ(x) ((x) * (x))
#define FNORM(b) sqrtf(SQ(cf[3 * (b)]) + SQ(cf[3 * (b) + 1]) + SQ(cf[3 * (b) + 2]))
#define FRB(f, k) rb[(size_t)(V.feet_rows ? (f) : m.feet_indices[f]) * 13 + (k)]      /* component k of foot f's rigid-body row */
  switch (id) {
    case LG_REW_LIN_VEL_Z: return SQ(blv[2]);
    case LG_REW_ANG_VEL_XY: return stand ? SQ(bav[1]) + SQ(bav[2]) : SQ(bav[0]) + SQ(bav[1]);        // anymal.py:264-266
    case LG_REW_ORIENTATION: return stand ? SQ(pg[1]) + SQ(pg[2]) : SQ(pg[0]) + SQ(pg[1]);             // anymal.py:268-271
    case LG_REW_ORIENTATION_LOAD_ADAPT:   // base perpendicular to gravity + acceleration (anymal.py:140-143)
      return SQ(pg[0] - V.bla[0] / 9.81f) + SQ(pg[1] - V.bla[1] / 9.81f);
    case LG_REW_BASE_HEIGHT: {
      float s = root[2];
      if (g.measure_heights) s = bh / (float)C->P;
      return SQ(s - g.base_height_target);
    }
    case LG_REW_BASE_FOOT_HEIGHT: {
      float s = 0.f; int n = 0;
      for (int f = 0; f < NLEG; ++f) if (ctime[f] > 1e-3f) { s += FRB(f, 2); ++n; }
      float est = n > 0 ? s / (float)n : root[2] - g.base_height_target;
      return SQ((root[2] - est) - g.base_height_target);
    }
    case LG_REW_TORQUES: return feat[F_TQ2];
    case LG_REW_DOF_VEL: return feat[F_QD2];
    case LG_REW_DOF_ACC: return feat[F_ACC2];
    case LG_REW_ACTION_RATE: return feat[F_ARATE2];
    case LG_REW_DOF_POS_LIMITS: return feat[F_POSLIM];
    case LG_REW_DOF_VEL_LIMITS: return feat[F_VELLIM];
    case LG_REW_TORQUE_LIMITS: return feat[F_TQLIM];
    case LG_REW_COLLISION: { float s = 0.f; for (int i = 0; i < m.num_penalised; ++i) s += fn[m.penalised_contact_indices[i]] > 0.1f ? 1.f : 0.f; return s; }
    case LG_REW_FEET_STUMBLE: {
      bool any = false;
      for (int f = 0; f < NLEG; ++f) { int b = m.feet_indices[f]; any |= sqrtf(SQ(cf[3 * b]) + SQ(cf[3 * b + 1])) > 5.f * fabsf(cf[3 * b + 2]); }
      return any ? 1.f : 0.f;
    }
    case LG_REW_FEET_STUMBLE_LIFTUP: {
      float s = 0.f;
      for (int f = 0; f < NLEG; ++f) { int b = m.feet_indices[f]; bool st = sqrtf(SQ(cf[3 * b]) + SQ(cf[3 * b + 1])) > 5.f * fabsf(cf[3 * b + 2]); s += (st ? 1.f : 0.f) * FRB(f, 9); }
      return s;
    }
    case LG_REW_FEET_SLIP: {
      float s = 0.f;
      for (int f = 0; f < NLEG; ++f) { int b = m.feet_indices[f]; bool cfl = (cf[3 * b + 2] > 1.f) || lastc[f]; float vn = sqrtf(SQ(FRB(f, 7)) + SQ(FRB(f, 8))); s += (cfl ? 1.f : 0.f) * SQ(vn); }
      return s;
    }
    case LG_REW_JUMP_AIR: {
      float s = 0.f;
      for (int f = 0; f < NLEG; ++f) { int b = m.feet_indices[f]; bool cfl = (cf[3 * b + 2] > 1.f) || lastc[f]; s += (cfl ? 0.f : 1.f) * (air[f] - 0.5f); }
      return fmaxf(s - NLEG / 2.f, 0.f);      // RM:148: len(feet_indices) / 2
    }
    case LG_REW_FEET_AIR_TIME: {   // RM:150-163, stateful; the stand classes run it on feet 1 and 3 only and leave feet_contact_time alone (anymal.py:287-299)
      float s = 0.f;
      for (int f = stand ? 1 : 0; f < (stand ? 4 : NLEG); f += stand ? 2 : 1) {     // (the stand classes of both robots: feet_indices[1] and [3], anymal.py:289, elspider.py:706)
        int b = m.feet_indices[f]; bool contact = cf[3 * b + 2] > 1.f; bool cfl = contact || lastc[f];
        lastc[f] = contact ? 1 : 0;
        float first = (air[f] > 0.f && cfl) ? 1.f : 0.f;
        float a = air[f] + dt, ct = ctime[f] + dt;
        s += (a - 0.5f) * first;
        air[f] = a * (cfl ? 0.f : 1.f); if (!stand) ctime[f] = ct * (cfl ? 1.f : 0.f);
      }
      return s * ((g.feet_air_time_ungated || cmdn > 0.1f) ? 1.f : 0.f);
    }
    case LG_REW_PENALTY_IN_THE_AIR: {   // anymal.py:301-308
      bool any = false;
      for (int f = 1; f < 4; f += 2) any |= (cf[3 * m.feet_indices[f] + 2] > 1.f) || lastc[f];
      return any ? 0.f : 1.f;
    }
    case LG_REW_FEET_CONTACT_FORCES: { float s = 0.f; for (int f = 0; f < NLEG; ++f) s += fmaxf(fn[m.feet_indices[f]] - g.max_contact_force, 0.f); return s; }
    case LG_REW_GAIT_2_STEP: {
#define SYNC(a, b) (fminf(SQ(air[a] - air[b]), 4.f) + fminf(SQ(ctime[a] - ctime[b]), 4.f))
#define ASYN(a, b) (fminf(SQ(air[a] - ctime[b]), 4.f) + fminf(SQ(ctime[a] - air[b]), 4.f))
#if NLEG == 6   // ElSpider._reward_gait_2_step (elspider.py:365-407): feet (alphabetical) 0 LB, 1 LF, 2 LM, 3 RB, 4 RF, 5 RM; tripods (0, 1, 5) and (2, 3, 4)
      float sr = ((SYNC(0, 1) + SYNC(0, 5) + SYNC(1, 5)) / 3 + (SYNC(2, 3) + SYNC(2, 4) + SYNC(3, 4)) / 3) / 2;
      float ar = (ASYN(0, 2) + ASYN(0, 3) + ASYN(0, 4) + ASYN(1, 2) + ASYN(1, 3) + ASYN(1, 4) + ASYN(5, 2) + ASYN(5, 3) + ASYN(5, 4)) / 9;
#else
      float sr = (SYNC(0, 3) + SYNC(1, 2)) / 2;
      float ar = (ASYN(0, 1) + ASYN(0, 2) + ASYN(3, 2) + ASYN(3, 1)) / 4;
#endif
      float other = g.heading_command ? cmd[3] : cmd[2];
      bool on = cmdn > 0.1f || fabsf(other) >= 0.05f;
      return (sr + ar) * (on ? 1.f : 0.f);
    }
    case LG_REW_FOUR_FOOTUP: { bool all = true; for (int f = 0; f < NLEG; ++f) all &= cf[3 * m.feet_indices[f] + 2] < 1.f; return 0.1f * (all ? 1.f : 0.f); }
    case LG_REW_TERMINATION: return (C->reset_buf[e] && !C->time_out[e]) ? 1.f : 0.f;
    case LG_REW_STAND_STILL: return feat[F_STILL] * (cmdn < 0.1f ? 1.f : 0.f);
    case LG_REW_ASYNC_GAIT_SCHEDULER: return async_gait_value(g, V.dof);
    case LG_REW_NO_FLY: { int nc = 0; for (int f = 0; f < NLEG; ++f) nc += cf[3 * m.feet_indices[f] + 2] > 0.1f ? 1 : 0; return nc == 1 ? 1.f : 0.f; }   // cassie.py:42-45
    case LG_REW_TRACKING_LIN_VEL:       // stand: commands against -base_lin_vel[1:] (anymal.py:277-281)
      return expf(-(stand ? SQ(cmd[0] + blv[1]) + SQ(cmd[1] + blv[2]) : SQ(cmd[0] - blv[0]) + SQ(cmd[1] - blv[1])) / g.tracking_sigma);
    case LG_REW_TRACKING_ANG_VEL: return expf(-SQ(cmd[2] - (stand ? bav[0] : bav[2])) / g.tracking_sigma);   // anymal.py:283-286
    case LG_REW_GAIT_SCHEDULER: {   // gait_scheduler.py:74-81 on the foot heights / phase stored by the previous step
      if (!g.gait_enabled || step <= 1) return 0.f;
      float gi = C->gait_idx[e]; const float* fz = C->gait_foot_z + (size_t)e * NLEG; float s = 0.f;
      for (int f = 0; f < NLEG; ++f) {
        float ph = fmodf(gi + g.gait_foot_phases[f], 1.0f);
        float tgt = ph < 0.5f ? g.gait_swing_height * sinf(6.28318530717958647692f * ph) : 0.f;
        s += SQ(tgt - fz[f]);
      }
      return s;
    }
  }
  return 0.f;
}

// Episode statistics of one step: fixed-order (ascending workgroup) sums of the per-workgroup partial rows -> extras
// (LR:200-206), step counters, running stats.  Only workgroups that reset an env have a row to add (flags), so the usual
// step touches a handful of rows; the order of the additions never depends on scheduling.  One 256-thread workgroup,
// launched after reset_idx_kernel (its single row); a policy step runs finalize_from_acc inside the post kernel instead.
//   bump: 1 = policy step (counters[0]), 0 = explicit reset (counters[2]), 2 = rollout step (counters[3], nothing else)
#define FIN_CHUNK 1024
// A policy step has no second launch for its statistics: the post kernel's workgroups count their arrival on sharded
// device-scope counters and the last one runs finalize_from_acc (a single counter for 1024 workgroups cost 12 us, a
// list walk over per-workgroup rows 6 us; sharded counters + integer accumulators 4.4 us, against 5.8 us + a kernel
// boundary for the separate launch).  finalize_kernel remains for lg_reset_idx (one row, no arrival counting).
// written by one workgroup, read by another (possibly on another XCD, behind another L2) within ONE launch: relaxed
// device-scope accesses (sc1: write-through / L2-bypassing), ordered by the arrival counters below
LG_DEV void st_dev(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
LG_DEV float ld_dev(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
LG_DEV void st_dev(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
LG_DEV unsigned ld_dev(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// the published half of the statistics step: extras, command curriculum, counters, running totals
// nthr: threads of the finishing workgroup, 256 or 128 (the hexapod's post kernel, round 6).  The level sum keeps the order of 256 threads either way -- a
// thread of a 128-thread workgroup stands for the virtual threads tid and tid + 128 (lvl_hi) -- so the mean level does not depend on who finishes the step.
LG_DEV void finalize_publish(const DevCtx* __restrict__ C, const float* tot, float* f_lvl, float lvl_acc, int bump, int tid, bool use_flags, float lvl_hi = 0.f, int nthr = 256) {
  const int K = C->cfg.num_reward_terms;
  const bool want_lvl = C->cfg.curriculum != 0;
  lds_barrier();
  const float cnt = tot[K];
  if (cnt > 0.f && want_lvl) {            // mean terrain level over all envs (LR:205-206): only reported with a reset
    if (!use_flags) lvl_acc = tid == 0 ? ld_dev(C->partials + K + 1) : 0.f;
    f_lvl[tid] = lvl_acc;
    if (nthr == 128) f_lvl[tid + 128] = use_flags ? lvl_hi : 0.f;
    lds_barrier();
    for (int off = 128; off > 0; off >>= 1) { for (int vt = tid; vt < off; vt += nthr) f_lvl[vt] += f_lvl[vt + off]; lds_barrier(); }
  }
  if (cnt > 0.f) {
    if (tid < K) C->extras[tid] = tot[tid] / cnt / C->cfg.max_episode_length_s;
    if (tid == K && C->cfg.curriculum) C->extras[K] = f_lvl[0] / (float)C->N;
  }
  if (tid == 0 && cnt > 0.f && C->cfg.command_curriculum) {
    // update_command_curriculum (LR:520-533), gated like LR:178: every max_episode_length steps, from the mean
    // tracking_lin_vel episode sum of the envs reset in this step.  The widened range serves every later draw (the
    // reference applies it already to the commands of those same envs: their reset runs before this statistics step).
    const lg_config& g = C->cfg;
    const int64_t common = C->counters[0] + (bump == 1 ? 1 : 0);
    int kt = -1;
    for (int k = 0; k < K; ++k) if (g.reward_term_ids[k] == LG_REW_TRACKING_LIN_VEL) kt = k;
    if (kt >= 0 && common % (int64_t)g.max_episode_length == 0 &&
        tot[kt] / cnt / g.max_episode_length > 0.8f * g.reward_scales[kt]) {
      C->cmd_ranges[0] = fminf(fmaxf(C->cmd_ranges[0] - 0.5f, -g.max_curriculum), 0.f);
      C->cmd_ranges[1] = fminf(fmaxf(C->cmd_ranges[1] + 0.5f, 0.f), g.max_curriculum);
    }
  }
  if (tid == 0) {
    if (bump == 1) C->counters[0] += 1; else if (bump == 0) C->counters[2] += 1;
    C->counters[1] = (int64_t)cnt;
    double ret = 0.0;
    for (int k = 0; k < K; ++k) ret += (double)tot[k];
    C->ep_stats[0] += ret; C->ep_stats[1] += (double)tot[K + 2]; C->ep_stats[2] += (double)cnt;
    if (bump == 1) C->ep_stats[3] += (double)C->n_stepped;
  }
}

LG_DEV void finalize_step(const DevCtx* __restrict__ C, int nblocks, int bump, int tid, bool use_flags) {
  const int K = C->cfg.num_reward_terms, KP = K + 3, lane = tid & 63, wv = tid >> 6;
  __shared__ float tot[PART_STRIDE];
  __shared__ int f_list[FIN_CHUNK];
  __shared__ int f_n;
  __shared__ float f_buf[8][32];
  __shared__ float f_lvl[256];
  if (bump == 2) { if (tid == 0) C->counters[3] += 1; return; }
  __shared__ unsigned char f_flag[FIN_CHUNK];
  if (tid < PART_STRIDE) tot[tid] = 0.f;
  float lvl_acc = 0.f;                     // this thread's share of the terrain-level sum (fixed assignment of rows)
  const bool want_lvl = C->cfg.curriculum != 0;
  for (int c0 = 0; c0 < nblocks; c0 += FIN_CHUNK) {
    const int cn = min(FIN_CHUNK, nblocks - c0);
    {                                      // flags and level sums of the chunk: independent loads, one latency
      unsigned fl[FIN_CHUNK / 256]; float lv[FIN_CHUNK / 256];
#pragma unroll
      for (int i = 0; i < FIN_CHUNK / 256; ++i) {
        const int b = tid + 256 * i;
        fl[i] = (b < cn && use_flags) ? ld_dev(C->part_flag + c0 + b) : (b < cn ? 1u : 0u);
        lv[i] = (b < cn && use_flags && want_lvl) ? ld_dev(C->lvl_part + c0 + b) : 0.f;
      }
      lds_barrier();
#pragma unroll
      for (int i = 0; i < FIN_CHUNK / 256; ++i) { f_flag[tid + 256 * i] = fl[i] ? 1 : 0; lvl_acc += lv[i]; }
    }
    lds_barrier();
    if (wv == 0) {                       // ascending list of the workgroups of this chunk that hold a row
      int cnt = 0;
      for (int base = 0; base < cn; base += 64) {
        const int b = base + lane;
        const bool f = b < cn && f_flag[b] != 0;
        const unsigned long long mk = __ballot(f);
        if (f) f_list[cnt + __popcll(mk & ((1ull << lane) - 1ull))] = c0 + b;
        cnt += __popcll(mk);
      }
      if (lane == 0) f_n = cnt;
    }
    lds_barrier();
    const int nl = f_n;
    for (int j0 = 0; j0 < nl; j0 += 8) {       // 8 rows per pass: loads in parallel, additions in row order
      const int j = j0 + (tid >> 5), c = tid & 31;
      f_buf[tid >> 5][c] = (j < nl && c < KP) ? ld_dev(C->partials + (size_t)f_list[j] * PART_STRIDE + c) : 0.f;
      lds_barrier();
      if (tid < KP) { float sacc = tot[tid]; for (int jj = 0; jj < 8; ++jj) sacc += f_buf[jj][tid]; tot[tid] = sacc; }
      lds_barrier();
    }
  }
  finalize_publish(C, tot, f_lvl, lvl_acc, bump, tid, use_flags);
}

// Statistics step of a policy step, run by the last workgroup of the post kernel: the workgroups that reset an env have
// added their rows to 64-bit fixed-point accumulators (integer atomics: the sum does not depend on arrival order), so
// this is one round trip (the accumulators and the per-workgroup level sums together) instead of a list walk.
#define ACC_SCALE 16777216.0
LG_DEV void finalize_from_acc(const DevCtx* __restrict__ C, int nblocks, int bump, int tid, bool subset, int nsteps = 1) {
  const int K = C->cfg.num_reward_terms, KP = K + 3;
  __shared__ float tot[PART_STRIDE];
  __shared__ float f_lvl[256];
  if (bump == 2) { if (tid == 0) C->counters[3] += nsteps; return; }      // (a persistent rollout launch has run nsteps steps)
  const int nthr = (int)blockDim.x;                     // 256, or 128 (the hexapod's post kernel): see finalize_publish
  float lvl_acc = 0.f, lvl_hi = 0.f;
  if (C->cfg.curriculum != 0) for (int b = tid; b < nblocks; b += 256) lvl_acc += ld_dev(C->lvl_part + b);
  if (C->cfg.curriculum != 0 && nthr == 128) for (int b = tid + 128; b < nblocks; b += 256) lvl_hi += ld_dev(C->lvl_part + b);
  if (C->cfg.curriculum != 0 && subset) {
    for (int b = tid; b < nblocks; b += 256) lvl_acc -= ld_dev(reinterpret_cast<const float*>(C->part_flag) + b);
    if (nthr == 128) for (int b = tid + 128; b < nblocks; b += 256) lvl_hi -= ld_dev(reinterpret_cast<const float*>(C->part_flag) + b);
    if (tid == 0) lvl_acc += C->lvl_total_before;
  }
  if (tid < PART_STRIDE) {
    long long a = 0;
    if (tid < KP) {
      a = __hip_atomic_load(C->acc + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a != 0) __hip_atomic_store(C->acc + tid, 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    tot[tid] = (float)((double)a * (1.0 / ACC_SCALE));
  }
  finalize_publish(C, tot, f_lvl, lvl_acc, bump, tid, true, lvl_hi, nthr);
}

// ============================================================================================ post-physics kernel
// LDS staging area per env (floats)
enum { S_ROOT = 0, S_DOF = 13, S_CF = S_DOF + 2 * NDOF, S_RB = S_CF + NBODY_MAX * 3, S_ACT = S_RB + NBODY_MAX * 13, S_LACT = S_ACT + NDOF,
       S_LDV = S_LACT + NDOF, S_TQ = S_LDV + NDOF, S_LRV = S_TQ + NDOF, S_CMD = S_LRV + 6, S_BLA = S_CMD + 4, S_BAA = S_BLA + 3,
       S_AIR = S_BAA + 3, S_CT = S_AIR + NLEG, S_BLV = S_CT + NLEG, S_BAV = S_BLV + 3, S_PG = S_BAV + 3, S_SUMS = S_PG + 3,
       S_GAIT = S_SUMS + LG_MAX_REWARD_TERMS, S_STRIDE = S_GAIT + 1 + 2 };

// LDS of one post-physics instance (EPBP envs): a workgroup of post_kernel, or one wave of the fused step kernel
struct alignas(16) PostLds {
  int64_t s_eplen[EPBP];
  float s_h[EPBP][MAX_P];
  float s_env[EPBP][S_STRIDE];
  float s_prop[EPBP][NPROP];
  float s_part[EPBP][PART_STRIDE];
  float s_u[EPBP][LG_RS_NOISE];      // uniforms of slots 0..31 (commands, push, curriculum, reset)
  float s_feat[EPBP][F_COUNT][NDOF];
  float s_fsum[EPBP][F_COUNT];
  float s_fn[EPBP][NBODY_MAX];
  float s_rk[EPBP][LG_MAX_REWARD_TERMS];
  float s_old[EPBP][2 * NLEG];
  float s_rootz[EPBP], s_bh[EPBP];
  float s_level[EPBP];               // terrain level at the start of the step (re-read after a reset)
  float s_level0[EPBP];              // ... and a copy that stays (subset steps: the finisher needs the sum before and after)
  int s_e[EPBP];
  uint8_t s_lastc[EPBP][NLEG], s_oldc[EPBP][NLEG];
  uint8_t s_flag[EPBP], s_did_reset[EPBP], s_root_dirty[EPBP], s_term[EPBP], s_tout[EPBP];
};

// The post-physics step of one instance = EPBP envs (4; 2 for the hexapod): rows [inst * EPBP, inst * EPBP + EPBP) of the launch.
//   FUSED = false: the instance is a workgroup of post_kernel, EPBP waves; in the wide stages wave w owns env w.
//   FUSED = true:  the instance is ONE wave of the fused step kernel (physics_kernel's tail: the four waves of a physics
//                  workgroup take four envs each of its sixteen); the wave owns all four envs in the wide stages (NQ = 4
//                  rounds of the same lane mapping) and runs the narrow stages for them as below.  The workgroup barriers
//                  stay (every wave passes the same ones); the instances never exchange data.
// ids/n: optional env subset (row k of the launch <-> env ids[k]); mode 0 = LeggedRobot.post_physics_step,
// mode 1 = RobotBatchRollout.post_physics_step_rollout (robot_batch_rollout.py:763-817: no callback, no termination,
// no reset, rewards without episode sums).
// rew_out (optional): the reward of launch row k also goes to rew_out[k * rew_stride] (lg_rollout_batch: column i of (n, horizon)).
// ninst: instances of the launch (rows of the per-instance statistics the finishing workgroup adds up).
// Optional second destinations of a step that feeds a rollout storage directly (lg_step_transition): row k of the launch
//   obs_out[k, :]  = the observation row (RolloutStorage.observations[t + 1]: what the policy acts on next),
//   rewards[k]     = rew + gamma * (values[k] * time_out)   (PPO.process_env_step, ppo.py:179-183: three roundings),
//   dones[k]       = reset flag as float (ppo.py:165).
template <bool FUSED>
LG_DEV void post_instance(const DevCtx* __restrict__ C, const int32_t* __restrict__ ids, int n, int mode, float* __restrict__ rew_out,
                          int rew_stride, PostLds& L, int inst, int ninst, const PostSink K) {
  constexpr int NQ = FUSED ? EPBP : 1;          // envs a wave owns in the wide stages
  const lg_config& g = C->cfg; const lg_robot_model& m = C->model;
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int itid = FUSED ? ln : tid;            // thread index within the instance
  const int e0 = inst * EPBP;
  const int nenv = max(0, min(EPBP, n - e0));
  const bool ro = mode == 1;
  if (itid < EPBP) L.s_e[itid] = itid < nenv ? (ids ? ids[e0 + itid] : e0 + itid) : 0;
  lds_barrier();
  const int P = g.measure_heights ? C->P : 0;
  const int64_t step = ro ? C->counters[3] + 1 : C->counters[0] + 1;   // LR:123 (finalize_kernel stores it)
  const uint32_t rstream = ro ? 2u : 0u;
  // what the gait term's "has a scheduler step run yet" test sees (reward_term: step <= 1 -> nothing): a rollout step of AnymalCBatchRollout
  // sees the phases left by the main steps before it
  const int64_t gstep = ro ? C->counters[0] + 1 : step;
  const float dt = g.sim_dt * g.decimation;
  const int B = C->B;
#ifdef LG_STAMPS
  unsigned long long* stamps = (inst == 0 && itid == 0) ? C->stamps : nullptr;
  unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
#endif

  // Wide stages: a wave owns an env from the first load to the last store, lane ln owns entry ln (+ 64 j) of each of its
  // rows.  No index needs an integer division (rows used to be dealt to the 256 threads of the workgroup by flat index:
  // ~25 instructions per division by a run-time row length, a dozen of them).
  const int wv_el = wv < nenv ? wv : 0;          // FUSED = false: this wave's env (waves without one shadow env 0 and store nothing)
  bool hv_[NQ]; int eq[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int el = FUSED ? q : wv_el;
    hv_[q] = FUSED ? q < nenv : wv < nenv;
    eq[q] = L.s_e[el];
  }

  // Per-DOF constants of the narrow stages (lane sl of an env's sixteen owns DOF sl there), requested now so that their
  // latency overlaps the staging loads instead of standing in the dependent chain of stage (2.1) / (2.4); likewise the
  // 64-bit modulo of the push schedule.
  const int sl_pf = ln & (LPE - 1), d_pf = min(sl_pf, NDOF - 1);
  const float pf_lim0 = g.dof_pos_limits[d_pf][0], pf_lim1 = g.dof_pos_limits[d_pf][1];
  const float pf_vlim = m.dof_vel_limit[d_pf], pf_tlim = m.torque_limit[d_pf], pf_dflt = g.default_dof_pos[d_pf];
  const float pf_dflt_r0 = g.default_dof_pos[min(max(sl_pf - 12, 0), NDOF - 1)], pf_dflt_r1 = g.default_dof_pos[min(sl_pf + LPE - 12, NDOF - 1)];   // entries 12..12+NDOF-1 of the observation: rounds 0 / 1 of stage (2.4)
  // (the step counter is 64-bit; its modulo is a ~150-instruction emulation, a 32-bit one a fifth of that: taken whenever it fits)
  const bool push_hit = (step >> 32) == 0 ? ((uint32_t)step % (uint32_t)g.push_interval == 0u) : (step % g.push_interval == 0);
  const bool push_now = !ro && g.push_robots && push_hit;                                                               // LR:402-403
  const float pf_value = K.values ? K.values[min(inst * EPBP + ln / LPE, n - 1)] : 0.f;   // critic value of the narrow-stage lane's env (time-out bootstrap)

  // ---- (1a) height scan from the post-physics root pose (LR:400-401).  Order of the memory traffic of this kernel's
  // first stage: [scan inputs: base pose + scan points] -> [all staging loads] -> wait for the scan inputs only ->
  // [height gathers] -> LDS stores of the staged rows -> barrier; the gathers are consumed after stage (2.1).
  // No branch around any of these loads: lanes without a point, and the plane, read a valid dummy address (a
  // conditional load makes the compiler drain the memory pipeline at the end of the branch).
  static_assert(MAX_P <= 3 * 64, "height scan assumes three points per lane");
  const bool scan = P > 0 && !ro;
  const bool plane = C->ter.mesh_type == LG_MESH_PLANE;
  HeightProbe hp_[NQ][3]; int hpi[NQ][3];
  float rq[NQ][4], hxy[3][2];
  {
    const float* hpts = scan ? C->height_points : C->root;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const float* rt = C->root + (size_t)eq[q] * 13;
      rq[q][0] = rt[0]; rq[q][1] = rt[1]; rq[q][2] = rt[5]; rq[q][3] = rt[6];
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int p = ln + 64 * u;
      const bool okp = scan && p < P;
#pragma unroll
      for (int q = 0; q < NQ; ++q) hpi[q][u] = (okp && hv_[q]) ? p : -1;
      const int pp = okp ? p : 0;
      hxy[u][0] = hpts[2 * pp]; hxy[u][1] = hpts[2 * pp + 1];
    }
  }
  float h_keep[NQ][3];                         // rollout steps keep the heights measured by the last main step
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int u = 0; u < 3; ++u) h_keep[q][u] = 0.f;
  if (P > 0 && ro) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int p = ln + 64 * u;
        if (hv_[q] && p < P) { hpi[q][u] = p; h_keep[q][u] = C->heights[(size_t)eq[q] * C->P + p]; }
      }
  }

  // ---- (0) stage the env rows in LDS.  Every global load is issued before the first LDS store, so the phase costs one
  // memory latency instead of one per tensor.
  constexpr int RBC = (NBODY_MAX * 13 + 63) / 64, CFC = (NBODY_MAX * 3 + 63) / 64;   // 64-lane chunks of the body-state / contact-force rows
  static_assert(2 * NDOF <= 64 && LG_MAX_REWARD_TERMS <= 64 && LG_RS_NOISE / 4 <= 64, "staging assumes one 64-lane chunk for every row but the per-body ones");
  const int LRB = B * 13;
  float v_rb[NQ][RBC], v_cf[NQ][CFC], v_row[NQ][14], v_sum[NQ];
  uint8_t v_lc[NQ], v_flag[NQ]; int64_t v_len[NQ], v_lvl[NQ];
#define LDV(i, SRC, LEN) v_row[q][i] = 0.f; if (ln < (LEN)) v_row[q][i] = (SRC)[(size_t)e * (LEN) + ln];
#define STV(i, OFF, LEN) if (hv_[q] && ln < (LEN)) S[(OFF) + ln] = v_row[q][i];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int e = eq[q];
#pragma unroll
    for (int j = 0; j < RBC; ++j) {
      const int k_ = ln + j * 64; v_rb[q][j] = 0.f;
      if (k_ < LRB) v_rb[q][j] = C->rigid[(size_t)e * LRB + k_];
    }
#pragma unroll
    for (int j = 0; j < CFC; ++j) {
      const int k_ = ln + j * 64; v_cf[q][j] = 0.f;
      if (k_ < B * 3) v_cf[q][j] = C->cforce[(size_t)e * (B * 3) + k_];
    }
    LDV(0, C->root, 13) LDV(1, C->dof, 2 * NDOF)
    LDV(3, C->actions, NDOF) LDV(4, C->last_actions, NDOF) LDV(5, C->last_dof_vel, NDOF) LDV(6, C->torques, NDOF)
    LDV(7, C->last_root_vel, 6) LDV(8, C->commands, 4) LDV(9, C->base_lin_acc, 3) LDV(10, C->base_ang_acc, 3)
    LDV(11, C->feet_air, NLEG) LDV(12, C->feet_ctime, NLEG) LDV(13, C->gait_idx, 1)
    v_sum[q] = 0.f;                                    // episode sums are (K, N): row k, envs contiguous
    if (ln < g.num_reward_terms) v_sum[q] = C->ep_sums[(size_t)ln * C->N + e];
    v_lc[q] = 0; if (ln < NLEG) v_lc[q] = C->last_contacts[(size_t)e * NLEG + ln];
    v_len[q] = C->ep_len[e]; v_flag[q] = C->reset_buf[e]; v_lvl[q] = C->levels[e];
    if (!(g.curriculum && !ro)) v_lvl[q] = 0;
  }
  float nv_pre[4] = {0.f, 0.f, 0.f, 0.f};            // noise scales of this lane's first four observation entries (stage 3)
  if (g.add_noise) {
#pragma unroll
    for (int i = 0; i < 4; ++i) if (4 * ln + i < g.num_obs) nv_pre[i] = C->noise_vec[4 * ln + i];
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int el = FUSED ? q : wv_el;
    if (hv_[q] && ln < LG_RS_NOISE / 4)               // one Philox call per (env, slot group): 8 lanes
      uniform_draw4(C, eq[q], ln, step, rstream, &L.s_u[el][4 * ln]);
  }
  // (1a, continued) the scan inputs are here: issue the height gathers
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const float qz = rq[q][2], qw = rq[q][3];
    const float nrm = fmaxf(sqrtf(qz * qz + qw * qw), 1e-9f);
    const float qzn = qz / nrm, qwn = qw / nrm;
#pragma unroll
    for (int u = 0; u < 3; ++u) hp_[q][u] = terrain_height_probe(C, qzn, qwn, rq[q][0], rq[q][1], hxy[u][0], hxy[u][1]);
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int el = FUSED ? q : wv_el;
    float* S = L.s_env[el];
#pragma unroll
    for (int j = 0; j < RBC; ++j) {
      const int k_ = ln + j * 64;
      if (hv_[q] && k_ < LRB) S[S_RB + k_] = v_rb[q][j];
    }
#pragma unroll
    for (int j = 0; j < CFC; ++j) {
      const int k_ = ln + j * 64;
      if (hv_[q] && k_ < B * 3) S[S_CF + k_] = v_cf[q][j];
    }
    STV(0, S_ROOT, 13) STV(1, S_DOF, 2 * NDOF)
    STV(3, S_ACT, NDOF) STV(4, S_LACT, NDOF) STV(5, S_LDV, NDOF) STV(6, S_TQ, NDOF)
    STV(7, S_LRV, 6) STV(8, S_CMD, 4) STV(9, S_BLA, 3) STV(10, S_BAA, 3)
    STV(11, S_AIR, NLEG) STV(12, S_CT, NLEG) STV(13, S_GAIT, 1)
    if (hv_[q] && ln < g.num_reward_terms) S[S_SUMS + ln] = v_sum[q];
    if (hv_[q] && ln < NLEG) L.s_lastc[el][ln] = v_lc[q];
    if (hv_[q] && ln == 0) { L.s_eplen[el] = v_len[q]; L.s_flag[el] = v_flag[q]; L.s_level[el] = (float)v_lvl[q]; L.s_level0[el] = (float)v_lvl[q]; }
  }
#undef LDV
#undef STV
  lds_barrier();
  STAMP(11);

  // (1b) the height samples have usually arrived by the end of stage (2.1): they are consumed after it (see below)
  // ---- (2) the post-physics logic of the workgroup's four envs on ONE wave, sixteen lanes per env.  The stages below are
  // a few lanes wide (five rotations, twelve DOFs, one lane per reward term ...), and their cost is instruction issue:
  // a wave per env made every SIMD issue each stage four times (four workgroups per CU, one wave of each per SIMD) with
  // at most a quarter of the lanes in use; now one wave per workgroup issues it once for all four envs, and which wave
  // that is rotates with the workgroup index so that the four workgroups of a CU load its four SIMDs evenly.
  // The reference's order of side effects is kept by stages separated by barriers:
  //   (2.1) base-frame velocities / accelerations / gravity (five rotations on five lanes), per-DOF reward features
  //         (one lane per DOF), contact-force norms (one lane per body);
  //   (2.2) feature sums (one lane per feature, DOF order = the serial order), callback + termination (one lane);
  //   (2.3) reward terms in config order, episode sums, reset;
  //   (2.4) proprioceptive observation entries, gait phase.
  static_assert(EPBP * LPE == 64 && F_COUNT <= 8 && NDOF <= LPE, "phase 2: LPE lanes per env on one wave, a lane per DOF");
  const unsigned term_mask = C->rew_term_mask;  // which reward terms are switched on (wave-uniform)
  const int kfat = C->rew_kfat, kterm = C->rew_kterm; const float term_scale = C->rew_term_scale;   // see reward_meta()
  {
    const bool mine = FUSED || wv == (int)((blockIdx.x >> 8) & (unsigned)(EPBP - 1));   // (workgroups 256 apart share a CU when all 1024 are resident)
    const int el = ln / LPE, sl = ln & (LPE - 1);           // env of the workgroup, lane within the env (shadow the wave-per-env names)
    const bool have = mine && el < nenv;
    float* S = L.s_env[have ? el : 0];
    EnvView V;
    V.root = S + S_ROOT; V.dof = S + S_DOF; V.cmd = S + S_CMD; V.air = S + S_AIR; V.ctime = S + S_CT;
    V.blv = S + S_BLV; V.bav = S + S_BAV; V.pg = S + S_PG; V.tq = S + S_TQ; V.act = S + S_ACT; V.lact = S + S_LACT; V.bla = S + S_BLA;
    V.ldv = S + S_LDV; V.cf = S + S_CF; V.rb = S + S_RB; V.lastc = L.s_lastc[have ? el : 0];
    float* root = V.root; float* dof = V.dof;

    // (2.1)
    if (have && sl < 5) {                                                           // LR:128-134
      const float* lrv = S + S_LRV;
      float q[4] = {root[3], root[4], root[5], root[6]};
      V3 lin = v3(root[7], root[8], root[9]), ang = v3(root[10], root[11], root[12]);
      V3 x = sl == 0 ? lin : sl == 1 ? lin - v3(lrv[0], lrv[1], lrv[2]) : sl == 2 ? ang : sl == 3 ? ang - v3(lrv[3], lrv[4], lrv[5]) : v3(0, 0, -1);
      V3 r = quat_rotate_inverse(q, x);
      const float ema = 0.9f, oma = (float)(1 - 0.9);
      float* dst = sl == 0 ? V.blv : sl == 1 ? S + S_BLA : sl == 2 ? V.bav : sl == 3 ? S + S_BAA : V.pg;
      if (sl == 1 || sl == 3) { dst[0] = dst[0] * ema + oma * r.x / dt; dst[1] = dst[1] * ema + oma * r.y / dt; dst[2] = dst[2] * ema + oma * r.z / dt; }
      else { dst[0] = r.x; dst[1] = r.y; dst[2] = r.z; }
    }
    if (have && sl < NDOF) {
      const int d = sl;
      const float q_ = dof[2 * d], qd = dof[2 * d + 1], tq = V.tq[d];
      float (*F)[NDOF] = L.s_feat[el];
      F[F_TQ2][d] = tq * tq;
      F[F_QD2][d] = qd * qd;
      { float a = (V.ldv[d] - qd) / dt; F[F_ACC2][d] = a * a; }
      { float a = V.lact[d] - V.act[d]; F[F_ARATE2][d] = a * a; }
      { float lo = q_ - pf_lim0, hi = q_ - pf_lim1; F[F_POSLIM][d] = -fminf(lo, 0.f) + fmaxf(hi, 0.f); }
      F[F_VELLIM][d] = fminf(fmaxf(fabsf(qd) - pf_vlim * g.soft_dof_vel_limit, 0.f), 1.f);
      F[F_TQLIM][d] = fmaxf(fabsf(tq) - pf_tlim * g.soft_torque_limit, 0.f);
      F[F_STILL][d] = fabsf(q_ - pf_dflt);
    }
    if (have) {
      for (int b = sl; b < B; b += LPE) {
        const float* cf = V.cf + 3 * b;
        L.s_fn[el][b] = sqrtf(cf[0] * cf[0] + cf[1] * cf[1] + cf[2] * cf[2]);
      }
    }
  }
  // (1b) the height samples: rows into LDS (and to the measured_heights tensor on main steps), every wave for its own envs
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int el = FUSED ? q : wv_el;
#pragma unroll
    for (int u = 0; u < 3; ++u) if (hpi[q][u] >= 0) {
      const float hv = ro ? h_keep[q][u] : (plane ? 0.f : terrain_height_value(C, hp_[q][u]));
      L.s_h[el][hpi[q][u]] = hv;
      if (!ro) C->heights[(size_t)eq[q] * C->P + hpi[q][u]] = hv;
    }
  }
  lds_barrier();
  STAMP(19);
  {
    const bool mine = FUSED || wv == (int)((blockIdx.x >> 8) & (unsigned)(EPBP - 1));
    const int el = ln / LPE, sl = ln & (LPE - 1);
    const bool have = mine && el < nenv;
    const int e = L.s_e[have ? el : 0];
    float* S = L.s_env[have ? el : 0];
    EnvView V;
    V.root = S + S_ROOT; V.dof = S + S_DOF; V.cmd = S + S_CMD; V.air = S + S_AIR; V.ctime = S + S_CT;
    V.blv = S + S_BLV; V.bav = S + S_BAV; V.pg = S + S_PG; V.tq = S + S_TQ; V.act = S + S_ACT; V.lact = S + S_LACT; V.bla = S + S_BLA;
    V.ldv = S + S_LDV; V.cf = S + S_CF; V.rb = S + S_RB; V.lastc = L.s_lastc[have ? el : 0];
    float* root = V.root; float* cmd = V.cmd;

    // (2.2)
    if (have && sl < F_COUNT) {
      float sacc = 0.f;
      for (int d = 0; d < NDOF; ++d) sacc += L.s_feat[el][sl][d];
      L.s_fsum[el][sl] = sacc;
    } else if (have && sl == 8) {
      float sacc = 0.f;
      if ((term_mask >> LG_REW_BASE_HEIGHT) & 1u) for (int p = 0; p < P; ++p) sacc += root[2] - L.s_h[el][p];
      L.s_bh[el] = sacc;
    } else if (have && sl == 9) {
      const int64_t eplen = L.s_eplen[el] + (ro ? 0 : 1);                             // LR:122 (not in rollout steps)
      bool root_dirty = false;
      // _post_physics_step_callback (LR:386-403)
      if (!ro && (int)eplen % g.resampling_steps == 0) resample_commands(C, cmd, L.s_u[el], LG_RS_CMD_CB);
      if (!ro && g.heading_command) {
        float q[4] = {root[3], root[4], root[5], root[6]};
        V3 f = quat_apply(q, v3(1, 0, 0));
        float x = 0.5f * wrap_to_pi(cmd[3] - atan2f(f.y, f.x));
        cmd[2] = fminf(fmaxf(x, -1.f), 1.f);
      }
      if (push_now) {                                                                // LR:402-403, 491-496
        root[7] = rand_float(-g.max_push_vel_xy, g.max_push_vel_xy, L.s_u[el][LG_RS_PUSH]);
        root[8] = rand_float(-g.max_push_vel_xy, g.max_push_vel_xy, L.s_u[el][LG_RS_PUSH + 1]);
        root_dirty = true;
      }
      // check_termination (LR:155-160)
      bool term = false;
      for (int i = 0; i < m.num_termination; ++i) term |= L.s_fn[el][m.termination_contact_indices[i]] > 1.f;
      term |= g.terminate_on_flip && V.pg[2] > 0.f;   // anymal_c_batch_rollout.py:192-198 (stage 2.1 left the new vector in LDS)
      term |= L.s_flag[el] == 2;        // physics fault flagged by physics_kernel
      if (!ro && C->extra_term) term |= C->extra_term[e] != 0;       // the env class's own reset rule (lg_set_extra_termination)
      bool tout = (float)eplen > g.max_episode_length;
      if (ro) {                       // rollout envs never terminate on their own: flags keep their last values
        tout = C->time_out[e] != 0; term = (L.s_flag[el] != 0) && !tout;
        if (L.s_flag[el] == 2) C->reset_buf[e] = 1;
      } else { C->time_out[e] = tout ? 1 : 0; C->reset_buf[e] = (term || tout) ? 1 : 0; }
      L.s_term[el] = term ? 1 : 0; L.s_tout[el] = tout ? 1 : 0;
      L.s_root_dirty[el] = root_dirty ? 1 : 0;
      L.s_eplen[el] = eplen;
      C->ep_len[e] = eplen;
    }
    lds_barrier();
    STAMP(20);

    // (2.3a) the one stateful term: _reward_feet_air_time rewrites air / contact times and last_contacts (RM:150-163).
    // Terms that come before it in the config order must still see the old values: keep a copy.
    if (have && sl == 0 && kfat < g.num_reward_terms) {
#pragma unroll
      for (int f = 0; f < NLEG; ++f) { L.s_old[el][f] = V.air[f]; L.s_old[el][NLEG + f] = V.ctime[f]; L.s_oldc[el][f] = V.lastc[f]; }
      L.s_rk[el][kfat] = reward_term(C, V, e, LG_REW_FEET_AIR_TIME, L.s_fsum[el], L.s_fn[el], L.s_bh[el], gstep) * g.reward_scales[kfat];
    }
    lds_barrier();
    // (2.3b) every other term on its own lane (they only read); more than sixteen terms take a second round
    for (int k = sl; have && k < g.num_reward_terms; k += LPE) {
      if (k == kfat) continue;
      const int id = g.reward_term_ids[k];
      EnvView Vk = V;
      if (k < kfat && kfat < g.num_reward_terms) { Vk.air = L.s_old[el]; Vk.ctime = L.s_old[el] + NLEG; Vk.lastc = L.s_oldc[el]; }
      L.s_rk[el][k] = id != LG_REW_TERMINATION ? reward_term(C, Vk, e, id, L.s_fsum[el], L.s_fn[el], L.s_bh[el], gstep) * g.reward_scales[k] : 0.f;
    }
    lds_barrier();
    STAMP(21);
    // (2.3c) total in config order, clip, termination term, reset (LR:215-232, 144-145)
    if (have && sl == 0) {
      const bool term = L.s_term[el] != 0, tout = L.s_tout[el] != 0;
      float rew = 0.f;
      for (int k = 0; k < g.num_reward_terms; ++k) rew += L.s_rk[el][k];
      if (g.only_positive_rewards) rew = fmaxf(rew, 0.f);
      if (kterm >= 0) {                                  // (found once at the top: no walk over the term list on this lane's chain)
        float r = ((term || tout) && !tout ? 1.f : 0.f) * term_scale;
        rew += r; L.s_rk[el][kterm] = r;
      }
      C->rew[e] = rew;
      if (rew_out) rew_out[(size_t)(e0 + el) * rew_stride] = rew;
      if (K.rewards) {
#pragma clang fp contract(off)
        const float boot = K.gamma * (pf_value * (tout ? 1.f : 0.f));
        K.rewards[e0 + el] = rew + boot;
        K.dones[e0 + el] = (term || tout) ? 1.f : 0.f;
      }
      const bool do_reset = !ro && (term || tout);
      if (do_reset) { reset_env(C, V, e, 1, L.s_u[el], false); L.s_root_dirty[el] = 1; if (g.curriculum) L.s_level[el] = (float)C->levels[e]; }
      L.s_rootz[el] = root[2];
      L.s_did_reset[el] = do_reset ? 1 : 0;
    }
    lds_barrier();

    // (2.4) proprioceptive part of the observation (LR:237-244), from the post-reset state; gait scheduler (anymal.py:107-110)
    if (have) {
      float* sp = L.s_prop[el];
      // entry i = (S[off_i] - sub_i) * scale_i, branch-free (sub is the default DOF position for entries 12..23, else 0;
      // x - 0 and x * 1 are exact, so this is the reference's arithmetic entry by entry)
      const float ls = g.obs_scale_lin_vel, as = g.obs_scale_ang_vel, ps = g.obs_scale_dof_pos, vs = g.obs_scale_dof_vel;
      static_assert(3 * LPE >= NPROP && 2 * LPE >= 12 + NDOF, "three rounds cover the proprioceptive entries, the first two the joint angles");
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int i = min(sl + LPE * r, NPROP - 1);       // (lanes past the row recompute its last entry)
        const int off = i < 3 ? S_BLV + i : i < 6 ? S_BAV + i - 3 : i < 9 ? S_PG + i - 6 : i < 12 ? S_CMD + i - 9
                      : i < 12 + NDOF ? S_DOF + 2 * (i - 12) : i < 12 + 2 * NDOF ? S_DOF + 2 * (i - 12 - NDOF) + 1 : S_ACT + i - 12 - 2 * NDOF;
        const float scale = i < 3 ? ls : i < 6 ? as : i < 9 ? 1.f : i < 11 ? ls : i < 12 ? as : i < 12 + NDOF ? ps : i < 12 + 2 * NDOF ? vs : 1.f;
        const float sub = (i >= 12 && i < 12 + NDOF) ? (r == 0 ? pf_dflt_r0 : pf_dflt_r1) : 0.f;
        sp[i] = (S[off] - sub) * scale;
      }
      if (sl == LPE - 1 && g.gait_enabled && !ro) {
        float x = fmodf(S[S_GAIT] + dt / g.gait_period, 1.0f); if (x < 0.f) x += 1.0f;
        S[S_GAIT] = x;
      }
      // episode sums and the statistics of LR:200-206, one lane per term
      const bool do_reset = L.s_did_reset[el] != 0;
      const int K_ = g.num_reward_terms;
      for (int k = sl; k < K_ + 3; k += LPE) {
        if (k < K_) {
          float tot = S[S_SUMS + k] + (ro ? 0.f : L.s_rk[el][k]);  // compute_reward_rollout does not touch the episode sums
          L.s_part[el][k] = do_reset ? tot : 0.f;
          S[S_SUMS + k] = do_reset ? 0.f : tot;        // written back to (K, N) by the env's own wave below
        } else if (k == K_) L.s_part[el][K_] = do_reset ? 1.f : 0.f;
        else if (k == K_ + 1) L.s_part[el][K_ + 1] = L.s_level[el];
        else L.s_part[el][K_ + 2] = do_reset ? (float)L.s_eplen[el] : 0.f;
      }
    }
  }
  lds_barrier();
  STAMP(13);

  // ---- (2b) write-back of everything phase (2) produced or changed; history buffers (LR:148-150)
#define UNSTAGE(DST, OFF, LEN) if (hv_[q] && ln < (LEN)) (DST)[(size_t)e * (LEN) + ln] = S[(OFF) + ln];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int el = FUSED ? q : wv_el, e = eq[q];
    const float* S = L.s_env[el];
    UNSTAGE(C->commands, S_CMD, 4) UNSTAGE(C->feet_air, S_AIR, NLEG) UNSTAGE(C->feet_ctime, S_CT, NLEG)
    UNSTAGE(C->base_lin_vel, S_BLV, 3) UNSTAGE(C->base_ang_vel, S_BAV, 3) UNSTAGE(C->proj_grav, S_PG, 3)
    UNSTAGE(C->base_lin_acc, S_BLA, 3) UNSTAGE(C->base_ang_acc, S_BAA, 3) UNSTAGE(C->gait_idx, S_GAIT, 1)
    UNSTAGE(C->last_actions, S_ACT, NDOF) UNSTAGE(C->last_root_vel, S_ROOT + 7, 6)
    if (hv_[q] && ln < NLEG) C->last_contacts[(size_t)e * NLEG + ln] = L.s_lastc[el][ln];
    if (hv_[q] && ln < 13 && L.s_root_dirty[el]) C->root[(size_t)e * 13 + ln] = S[S_ROOT + ln];
    if (hv_[q] && ln < 2 * NDOF && L.s_did_reset[el]) C->dof[(size_t)e * (2 * NDOF) + ln] = S[S_DOF + ln];
    if (hv_[q] && ln < NDOF) C->last_dof_vel[(size_t)e * NDOF + ln] = S[S_DOF + 2 * ln + 1];
    if (g.gait_enabled && !ro && hv_[q] && ln < NLEG) C->gait_foot_z[(size_t)e * NLEG + ln] = S[S_RB + m.feet_indices[ln] * 13 + 2];
    if (g.control_type == LG_CTRL_ACTUATOR_NET && hv_[q] && L.s_did_reset[el]) {   // anymal.py:78-82: clear the LSTM state of a reset env
      const size_t N12 = (size_t)C->N * NDOF;
      constexpr int PL = NDOF * 8;                     // floats per env and layer
#pragma unroll
      for (int j = 0; j < (2 * PL + 63) / 64; ++j) {
        const int idx = ln + 64 * j, lay = idx >= PL ? 1 : 0, k = idx - PL * lay;
        if (idx < 2 * PL) {
          C->sea_h[(lay * N12 + (size_t)e * NDOF) * 8 + k] = 0.f;
          C->sea_c[(lay * N12 + (size_t)e * NDOF) * 8 + k] = 0.f;
        }
      }
    }
    if (hv_[q] && ln < g.num_reward_terms) C->ep_sums[(size_t)ln * C->N + e] = S[S_SUMS + ln];   // back to the (K, N) rows
  }
#undef UNSTAGE

  // ---- per-instance episode statistics, summed in fixed env order (deterministic); written by the instance's first wave
  const int KP = g.num_reward_terms + 3;
  const bool stat_wave = FUSED || wv == 0;
  bool any_reset = false;
  for (int el2 = 0; el2 < nenv; ++el2) any_reset |= L.s_did_reset[el2] != 0;
  // rows of the reset envs: 64-bit fixed-point integer atomics (order-independent, hence deterministic)
  if (stat_wave && ln < KP && any_reset) {
    float sacc = 0.f;
    for (int el2 = 0; el2 < nenv; ++el2) sacc += L.s_part[el2][ln];
    __hip_atomic_fetch_add(C->acc + ln, __double2ll_rn((double)sacc * ACC_SCALE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (stat_wave && ln == KP + 1) {
    float sacc = 0.f;
    for (int el2 = 0; el2 < nenv; ++el2) sacc += L.s_part[el2][g.num_reward_terms + 1];
    st_dev(C->lvl_part + inst, sacc);
    if (ids) {      // subset step: the mean level is over ALL envs (LR:205-206, robot_batch_rollout.py:932) = total before - ours before + ours after
      float s0 = 0.f;
      for (int el2 = 0; el2 < nenv; ++el2) s0 += L.s_level0[el2];
      st_dev(reinterpret_cast<float*>(C->part_flag) + inst, s0);       // (the flag rows are only used by lg_reset_idx's finalize_kernel)
    }
  }
  // ---- arrival: the last workgroup to arrive finishes the step (statistics, extras, counters).  Arrivals are counted per
  // shard (blockIdx & 7: eight counters on eight cache lines, ~1/8 of the contention of one), the shard that fills up
  // counts itself on a ninth.  Everything the finishing workgroup reads (the accumulators, the level sums) was written
  // above as device-scope accesses by the statistics wave(s); each waits for them to be acknowledged (vmcnt(0)), and then
  // the workgroup counts its arrival.  The counter's reply is only looked at after the observation rows, which hide its
  // round trip.
  static_assert(LG_MAX_REWARD_TERMS + 3 + 2 <= 64, "statistics rows are written by one wave");
  unsigned arrival = 0;
  const unsigned shard = blockIdx.x & 7u, nsh = min(8u, gridDim.x);
  const unsigned want = (gridDim.x + 7u - shard) >> 3;
  if (stat_wave) __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): this wave's stores and atomics have completed
  if (FUSED) lds_barrier();                              // ... those of all four instances of the workgroup
  if (tid == 0) arrival = __hip_atomic_fetch_add(C->tickets + 32 * shard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // ---- (3) observation rows: proprio | heights | extra, + uniform noise, clipped (LR:245-252, :107-108); 4 entries per
  // lane, loads first (noise scales, caller's extra rows, injected uniforms), then arithmetic, then the row stores
  const int O = g.num_obs, G4 = (O + 3) >> 2;
  const bool inject = g.rng_mode == LG_RNG_INJECT;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int el3 = FUSED ? q : wv_el, e3 = eq[q];
    for (int gq = ln; hv_[q] && gq < G4; gq += 64) {
      float u[4] = {0.5f, 0.5f, 0.5f, 0.5f}, nv[4] = {0.f, 0.f, 0.f, 0.f}, ex[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = 4 * gq + i;
        const bool in = idx < O;
        if (g.add_noise && in) nv[i] = gq == ln ? nv_pre[i] : C->noise_vec[idx];
        if (g.add_noise && inject && in) u[i] = C->rand_inject[(size_t)e3 * (LG_RS_NOISE + O) + LG_RS_NOISE + idx];
        if (in && idx >= NPROP + P && C->extra_obs) ex[i] = C->extra_obs[(size_t)e3 * g.num_extra_obs + (idx - NPROP - P)];
      }
      if (g.add_noise && !inject) {
        uint32_t o4[4];
        philox4((uint32_t)e3, (uint32_t)step, (uint32_t)((LG_RS_NOISE >> 2) + gq), rstream, (uint32_t)g.seed, (uint32_t)(g.seed >> 32), o4);
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = u01(o4[i]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = 4 * gq + i;
        if (idx < O) {
          float o;
          if (idx < NPROP) o = L.s_prop[el3][idx];
          else if (idx >= NPROP + P) o = ex[i];
          else { float h = (L.s_rootz[el3] - 0.5f) - L.s_h[el3][idx - NPROP]; o = fminf(fmaxf(h, -1.f), 1.f) * g.obs_scale_height; }
          if (g.add_noise) o += (2.f * u[i] - 1.f) * nv[i];
          o = fminf(fmaxf(o, -g.clip_observations), g.clip_observations);
          C->obs[(size_t)e3 * O + idx] = o;
          if (K.obs_out) K.obs_out[(size_t)(e0 + el3) * O + idx] = o;
        }
      }
    }
  }
  STAMP(14);
  __shared__ int s_last;
  if (tid == 0) {
    int last = 0;
    if (arrival == want - 1u) {
      if (__hip_atomic_fetch_add(C->tickets + 32 * 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsh - 1u) {
        last = 1;
        for (int i = 0; i < 9; ++i) __hip_atomic_store(C->tickets + 32 * i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    s_last = last;
  }
  __syncthreads();
  if (s_last) finalize_from_acc(C, ninst, mode == 0 ? 1 : 2, tid, ids != nullptr);
}

__global__ __launch_bounds__(256) void post_kernel(const DevCtx* __restrict__ C, const int32_t* __restrict__ ids, int n, int mode,
                                                   float* __restrict__ rew_out, int rew_stride, PostSink sink) {
  __shared__ PostLds L;
  post_instance<false>(C, ids, n, mode, rew_out, rew_stride, L, (int)blockIdx.x, (int)gridDim.x, sink);
}

LG_DEV void generic_fused_tail(const DevCtx* __restrict__ C, float* lds, const int32_t* __restrict__ ids, int n, int mode, const PostSink& K) {
#if NJ == 3
  static_assert(4 * sizeof(PostLds) <= (size_t)LG_MAX_CP * CF_FIELDS * 64 * sizeof(float), "four post-physics instances must fit the contact-slot table");
#endif
  static_assert(EPB == 4 * EPBP || LG_LEGS != 6, "a physics workgroup's envs = four post-physics instances");
  const int wv = threadIdx.x >> 6;
  PostLds& L = reinterpret_cast<PostLds*>(lds)[wv];
  post_instance<true>(C, ids, n, mode, K.rew_out, K.rew_stride, L, (int)blockIdx.x * 4 + wv, (int)gridDim.x * 4, K);
}

#if LG_LEGS == 4
#include "lg_fused_post.h"

// Observation entry idx of the fused tail as one float4: x = bits(source offset | kind << 16) (kind 0: float of the env's LDS row, 1: height
// sample, 2: extra-observation column, 3: past the row), y = scale, z = subtracted offset, w = noise scale (0 without noise).  Per lane this
// was ~100 instructions of selects and four dependent loads at the top of the write-back of every step; it depends on the config only.
static std::vector<float> pack_obs_table(const lg_config& g, int P) {
  const int O = g.num_obs, Opad = ((O + 3) / 4) * 4 + 256;            // (+ 256: lanes past the last group read entry codes of kind 3)
  std::vector<float> t((size_t)Opad * 4, 0.f);
  for (int idx = 0; idx < Opad; ++idx) {
    int off = 0, kind = 3; float scale = 1.f, sub = 0.f, nv = 0.f;
    if (idx < O) {
      const float ls = g.obs_scale_lin_vel, as = g.obs_scale_ang_vel, ps = g.obs_scale_dof_pos, vs = g.obs_scale_dof_vel;
      off = idx < 3 ? FS_BLV + idx : idx < 6 ? FS_BAV + idx - 3 : idx < 9 ? FS_PG + idx - 6 : idx < 12 ? FS_CMD + idx - 9
          : idx < 12 + NDOF ? FS_DOF + 2 * (idx - 12) : idx < 12 + 2 * NDOF ? FS_DOF + 2 * (idx - 12 - NDOF) + 1 : idx < NPROP ? FS_ACT + idx - 12 - 2 * NDOF : 0;
      scale = idx < 3 ? ls : idx < 6 ? as : idx < 9 ? 1.f : idx < 11 ? ls : idx < 12 ? as : idx < 12 + NDOF ? ps : idx < 12 + 2 * NDOF ? vs : 1.f;
      sub = (idx >= 12 && idx < 12 + NDOF) ? g.default_dof_pos[idx - 12] : 0.f;
      kind = idx < NPROP ? 0 : (idx < NPROP + P ? 1 : 2);
      if (kind == 1) { off = std::min(std::max(idx - NPROP, 0), MAX_P - 1); scale = g.obs_scale_height; }
      if (kind == 2) off = idx - NPROP - P;
      nv = g.add_noise ? g.noise_scale_vec[idx] : 0.f;
    }
    const int code = off | (kind << 16);
    memcpy(&t[(size_t)idx * 4], &code, 4); t[(size_t)idx * 4 + 1] = scale; t[(size_t)idx * 4 + 2] = sub; t[(size_t)idx * 4 + 3] = nv;
  }
  return t;
}
// glue between physics_kernel (which only sees declarations) and the tail
LG_DEV void fused_finalize(const DevCtx* __restrict__ C, int nblocks, int tid, bool ro, int nsteps) { finalize_from_acc(C, nblocks, ro ? 2 : 1, tid, false, nsteps); }
LG_DEV bool fused_did_reset(const float* HB, int el) { return HB[FH_MISC + el * FM_STRIDE + FM_DID_RESET] != 0.f; }
// the robot state a step ended with, from the env rows of its tail (persistent rollout launches: the next step starts from them)
LG_DEV void fused_reload_state(const float* SR, int lane, int l, float* root, float* q, float* qd) {
  const float* S = SR + (lane / GRP) * FS_STRIDE;
#pragma unroll
  for (int i = 0; i < 13; ++i) root[i] = S[FS_ROOT + i];
#pragma unroll
  for (int j = 0; j < 3; ++j) { q[j] = S[FS_DOF + 2 * (3 * l + j)]; qd[j] = S[FS_DOF + 2 * (3 * l + j) + 1]; }
}
LG_DEV float* fused_foot_row(float* xs, int lane) { return xs + (lane / GRP) * FS_STRIDE + FS_FRB + 13 * (lane % GRP); }
LG_DEV float* fused_act_slot(float* xs, int lane, int d) { return xs + (lane / GRP) * FS_STRIDE + FS_ACT + d; }
LG_DEV bool fused_needs_heights_early(const DevCtx* __restrict__ C) {
  return ((C->rew_term_mask >> LG_REW_BASE_HEIGHT) & 1u) != 0u && C->cfg.measure_heights && C->P > 0;
}
LG_DEV bool fused_needs_feet_rows(const DevCtx* __restrict__ C) {    // reward terms that read the feet's rigid-body rows
  return (C->rew_term_mask & ((1u << LG_REW_BASE_FOOT_HEIGHT) | (1u << LG_REW_FEET_STUMBLE_LIFTUP) | (1u << LG_REW_FEET_SLIP))) != 0u;
}
LG_DEV void fused_main_and_serial(const DevCtx* __restrict__ C, const float* hot, const LegModel& lm_, float* xs, float* UB, float* HB, int lane, int e, bool valid,
                                  const float* root, const float* q, const float* qd, const float* tau, const float* last_qd, const V3* fbody,
                                  const float* act_or_null, bool fault, int64_t step, unsigned long long* stamps, const PostSink& K, bool ro, int krow) {
  STAMP_DECL
  const int l = lane % GRP, el = lane / GRP;
  float* S = xs + el * FS_STRIDE;
  if (act_or_null) {        // PD-controlled robot with helper waves: the clipped actions are in this wave's registers
#pragma unroll
    for (int j = 0; j < 3; ++j) S[FS_ACT + 3 * l + j] = act_or_null[j];
  }
  float feat[F_COUNT];
  const FusedMainIn in{root, q, qd, tau, last_qd, fbody, fault};
  fused_main_part1(C, hot, lm_, S, l, in, feat);
  STAMP(11);
  if (fused_needs_feet_rows(C) || fused_needs_heights_early(C)) lds_barrier();   // (G1) only when the serial part reads a helper's product
  STAMP(19);
  if (l == 0 && valid)
    fused_env_serial(C, hot, e, S, UB + FU_U + el * LG_RS_NOISE, UB + FU_PRE + el * FU_PRE_STRIDE, HB + FH_MISC + el * FM_STRIDE,
                     HB + FH_HEIGHTS + el * MAX_P, feat, fault, step, K, ro, krow, K.rew_out, K.rew_stride);
  STAMP(20);
  lds_barrier();                                         // (G2) serial part + height scan done
  STAMP(21);
}

#else
// The post-physics step as the tail of the physics kernel (lg_fused_post.h) exists for the four-legged instance; the six-legged one ends
// its policy step in post_kernel (can_fuse() is false there, these are never called).
static std::vector<float> pack_obs_table(const lg_config&, int) { return std::vector<float>(4, 0.f); }
LG_DEV void fused_prefetch(const DevCtx* __restrict__, float*, float*, int, int, int, int64_t, const float*, const int32_t* __restrict__, bool) {}
LG_DEV void fused_height_scan(const DevCtx* __restrict__, const float (*)[20], float*, int, int, int, const int32_t* __restrict__, bool, const FusedPre&) {}
LG_DEV void fused_prefetch_static(const DevCtx* __restrict__, const float*, int, FusedPre& F) { F = FusedPre{}; }
LG_DEV bool fused_noise_predrawn(const float*) { return false; }
LG_DEV void fused_noise_draw(const float*, int, int, int, int64_t, float (*)[4], const int32_t* __restrict__, bool) {}
LG_DEV void fused_noise_park(const float*, float*, int, int, int, const float (*)[4]) {}
LG_DEV void fused_stage_obs_table(const float*, float*, int, const FusedPre&) {}
LG_DEV void fused_main_and_serial(const DevCtx* __restrict__, const float*, const LegModel&, float*, float*, float*, int, int, bool, const float*, const float*,
                                  const float*, const float*, const float*, const V3*, const float*, bool, int64_t, unsigned long long*, const PostSink&, bool, int) {}
LG_DEV bool fused_writeback_obs(const DevCtx* __restrict__, const float*, const float*, const float*, int, int, int, int64_t, unsigned long long*, float*, bool, const int32_t* __restrict__, bool, bool) { return false; }
LG_DEV void fused_finalize(const DevCtx* __restrict__, int, int, bool, int) {}
LG_DEV bool fused_did_reset(const float*, int) { return false; }
LG_DEV void fused_reload_state(const float*, int, int, float*, float*, float*) {}
LG_DEV float* fused_foot_row(float* xs, int) { return xs; }
LG_DEV float* fused_act_slot(float* xs, int, int) { return xs; }
LG_DEV bool fused_needs_heights_early(const DevCtx* __restrict__) { return false; }
LG_DEV bool fused_needs_feet_rows(const DevCtx* __restrict__) { return false; }
#endif

__global__ __launch_bounds__(256) void finalize_kernel(const DevCtx* __restrict__ C, int nblocks, int bump_step, int use_flags) {
  finalize_step(C, nblocks, bump_step, threadIdx.x, use_flags != 0);
}

// lg_set_state_indexed: one lane group per listed env (lane = leg): rows of the caller's full tensors -> simulation state, then the
// rigid-body rows of the new pose
// What a subset step hands back (robot_batch_rollout.py:598-600, 714-716: `obs_buf[ids]`, `rew_buf[ids]`, `reset_buf[ids]`, `time_outs[ids]`) as dense rows in
// ONE launch: the four index kernels the host framework runs for the same thing cost more than a quarter of a rollout step of 4096 envs
__global__ __launch_bounds__(256) void gather_step_rows_kernel(const DevCtx* __restrict__ C, const int32_t* __restrict__ ids, int n, float* __restrict__ obs_out,
                                                               float* __restrict__ rew_out, uint8_t* __restrict__ reset_out, uint8_t* __restrict__ time_out_out) {
  const int O = C->cfg.num_obs;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (obs_out && idx < (int64_t)n * O) {
    const int k = (int)(idx / O), col = (int)(idx - (int64_t)k * O);
    obs_out[idx] = C->obs[(size_t)ids[k] * O + col];
  }
  if (idx < n) {
    const int e = ids[idx];
    if (rew_out) rew_out[idx] = C->rew[e];
    if (reset_out) reset_out[idx] = C->reset_buf[e];
    if (time_out_out) time_out_out[idx] = C->time_out[e];
  }
}

__global__ __launch_bounds__(64) void set_state_kernel(const DevCtx* __restrict__ C, const float* __restrict__ root_src, const float* __restrict__ dof_src,
                                                       const int32_t* __restrict__ ids, int n) {
  __shared__ float lmod[LM_FIELDS * GRP];
  const int lane = threadIdx.x, l = lane % GRP;
  fill_leg_model(lmod, C->lmod, lane, 64);
  lds_barrier();
  const int kq = blockIdx.x * EPW + lane / GRP;
  if (kq >= n || l >= NLEG) return;
  const int e = ids[kq];
  if (e < 0 || e >= C->N) return;
  const LegModel lm_{lmod, l};
  const float* rs = (root_src ? root_src : C->root) + (size_t)e * 13;
  const float* ds = (dof_src ? dof_src : C->dof) + ((size_t)e * NDOF + NJ * l) * 2;
  float r13[13], q[NJ], qd[NJ];
#pragma unroll
  for (int i = 0; i < 13; ++i) r13[i] = rs[i];
#pragma unroll
  for (int j = 0; j < NJ; ++j) { q[j] = ds[2 * j]; qd[j] = ds[2 * j + 1]; }
  if (root_src && root_src != C->root && l == 0) {
#pragma unroll
    for (int i = 0; i < 13; ++i) C->root[(size_t)e * 13 + i] = r13[i];
  }
  if (dof_src && dof_src != C->dof) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) { C->dof[((size_t)e * NDOF + NJ * l + j) * 2] = q[j]; C->dof[((size_t)e * NDOF + NJ * l + j) * 2 + 1] = qd[j]; }
  }
  write_rigid_body_state(C, lm_, e, l, r13, q, qd);
  const int B = C->B, per_leg = C->per_leg;
  float* cf = C->cforce + (size_t)e * B * 3;
  if (l == 0) { cf[0] = 0.f; cf[1] = 0.f; cf[2] = 0.f; }
  for (int i = 0; i < per_leg * 3; ++i) cf[(1 + per_leg * l) * 3 + i] = 0.f;
}

// lg_reset_idx: one workgroup, loops over the id list; statistics summed in list order
__global__ __launch_bounds__(256) void reset_idx_kernel(const DevCtx* __restrict__ C, const int32_t* __restrict__ ids, int n, int update_curriculum) {
  const int tid = threadIdx.x, K = C->cfg.num_reward_terms;
  const int64_t step = C->counters[2] + 1;
  if (tid == K + 2) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += (float)C->ep_len[ids[i]];
    C->partials[K + 2] = s;
  }
  __syncthreads();   // global data crosses these barriers
  for (int i = tid; i < n; i += 256) {
    const EnvView V = global_view(C, ids[i]);
    float U[LG_RS_NOISE];
#pragma unroll
    for (int gq = 0; gq < LG_RS_NOISE / 4; ++gq) uniform_draw4(C, ids[i], gq, step, 1, U + 4 * gq);
    reset_env(C, V, ids[i], update_curriculum, U, true);
  }
  __syncthreads();   // global data crosses these barriers
  __threadfence_block();
  if (tid < K) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) { size_t o = (size_t)tid * C->N + ids[i]; s += C->ep_sums[o]; C->ep_sums[o] = 0.f; }
    C->partials[tid] = s;
  }
  if (tid == K) C->partials[K] = (float)n;
  if (tid == K + 1) {
    float s = 0.f;
    if (C->cfg.curriculum) for (int e = 0; e < C->N; ++e) s += (float)C->levels[e];
    C->partials[K + 1] = s;
  }
}

// ============================================================================================ host side
static size_t dtype_size(int d) { return d == LG_F32 ? 4 : (d == LG_I64 || d == LG_F64) ? 8 : d == LG_U8 ? 1 : d == LG_I16 ? 2 : 4; }

static size_t build_layout(const lg_config* cfg, const lg_robot_model* model, const lg_terrain* ter, TensorInfo* t) {
  const int64_t N = cfg->num_envs, B = model->num_bodies, P = cfg->num_height_points, O = cfg->num_obs;
  auto set = [&](int id, int dtype, std::initializer_list<int64_t> shp) {
    TensorInfo& T = t[id]; T.dtype = dtype; T.ndim = (int)shp.size(); int i = 0;
    for (auto s : shp) T.shape[i++] = s;
    for (; i < 4; ++i) T.shape[i] = 1;
  };
  set(LG_T_ROOT_STATES, LG_F32, {N, 13}); set(LG_T_DOF_STATE, LG_F32, {N, NDOF, 2});
  set(LG_T_RIGID_BODY_STATE, LG_F32, {N, B, 13}); set(LG_T_CONTACT_FORCES, LG_F32, {N, B, 3});
  set(LG_T_TORQUES, LG_F32, {N, NDOF}); set(LG_T_ACTIONS, LG_F32, {N, NDOF}); set(LG_T_LAST_ACTIONS, LG_F32, {N, NDOF});
  set(LG_T_LAST_DOF_VEL, LG_F32, {N, NDOF}); set(LG_T_LAST_ROOT_VEL, LG_F32, {N, 6}); set(LG_T_COMMANDS, LG_F32, {N, 4});
  set(LG_T_BASE_LIN_VEL, LG_F32, {N, 3}); set(LG_T_BASE_ANG_VEL, LG_F32, {N, 3}); set(LG_T_PROJECTED_GRAVITY, LG_F32, {N, 3});
  set(LG_T_BASE_LIN_ACC, LG_F32, {N, 3}); set(LG_T_BASE_ANG_ACC, LG_F32, {N, 3});
  set(LG_T_FEET_AIR_TIME, LG_F32, {N, NLEG}); set(LG_T_FEET_CONTACT_TIME, LG_F32, {N, NLEG}); set(LG_T_LAST_CONTACTS, LG_U8, {N, NLEG});
  set(LG_T_MEASURED_HEIGHTS, LG_F32, {N, P > 0 ? P : 1}); set(LG_T_OBS_BUF, LG_F32, {N, O}); set(LG_T_REW_BUF, LG_F32, {N});
  set(LG_T_RESET_BUF, LG_U8, {N}); set(LG_T_TIME_OUT_BUF, LG_U8, {N}); set(LG_T_EPISODE_LENGTH_BUF, LG_I64, {N});
  set(LG_T_EPISODE_SUMS, LG_F32, {LG_MAX_REWARD_TERMS, N}); set(LG_T_TERRAIN_LEVELS, LG_I64, {N}); set(LG_T_TERRAIN_TYPES, LG_I64, {N});
  set(LG_T_ENV_ORIGINS, LG_F32, {N, 3}); set(LG_T_FRICTION_COEFFS, LG_F32, {N}); set(LG_T_BASE_MASS_ADDED, LG_F32, {N});
  set(LG_T_SEA_HIDDEN_STATE, LG_F32, {2, N * NDOF, 8}); set(LG_T_SEA_CELL_STATE, LG_F32, {2, N * NDOF, 8});
  set(LG_T_GAIT_IDX, LG_F32, {N}); set(LG_T_GAIT_FOOT_Z, LG_F32, {N, NLEG}); set(LG_T_EXTRAS_EPISODE, LG_F32, {LG_MAX_REWARD_TERMS + 1});
  set(LG_T_RAND_INJECT, LG_F32, {N, LG_RS_NOISE + O}); set(LG_T_STEP_COUNTERS, LG_I64, {4});
  int64_t r = ter->rows > 0 ? ter->rows : 1, cc = ter->cols > 0 ? ter->cols : 1;
  set(LG_T_HEIGHT_SAMPLES, LG_I16, {r, cc});
  set(LG_T_TERRAIN_ORIGINS, LG_F32, {ter->num_levels > 0 ? ter->num_levels : 1, ter->num_types > 0 ? ter->num_types : 1, 3});
  set(LG_T_EPISODE_STATS, LG_F64, {4});
  set(LG_T_COMMAND_RANGES, LG_F32, {4, 2});
  size_t off = 0;
  for (int i = 0; i < LG_T_COUNT; ++i) {
    TensorInfo& T = t[i]; size_t n = dtype_size(T.dtype);
    for (int d = 0; d < T.ndim; ++d) n *= (size_t)T.shape[d];
    T.off = off; off += (n + 255) & ~(size_t)255;
  }
  return off;
}

static const char* validate(const lg_config* cfg, const lg_robot_model* model, const lg_terrain* ter) {
  if (!cfg || !model || !ter) return "null argument";
  if (cfg->abi_version != LG_ABI_VERSION) return "abi_version mismatch";
  if (cfg->num_envs <= 0) return "num_envs must be positive";
  if (cfg->num_reward_terms < 0 || cfg->num_reward_terms > LG_MAX_REWARD_TERMS) return "too many reward terms";
  if (cfg->num_height_points > MAX_P) return "num_height_points exceeds the 192 points the post kernel stages in LDS";
  if (cfg->num_extra_obs < 0) return "num_extra_obs is negative";
  if (model->num_legs != NLEG) return "num_legs does not match this kernel instance";
  if (model->num_joints_per_leg != NJ) return "num_joints_per_leg does not match this kernel instance (4 x 3, 6 x 3, 2 x 6)";
#if NJ != 3
  if (cfg->control_type == LG_CTRL_ACTUATOR_NET) return "the 2 x 6 instance has no actuator network (cassie_config.py: PD control)";
  if (ter->mesh_type == LG_MESH_TRIMESH && !ter->grid_vertices) return "the 2 x 6 instance collides with grid meshes (lg_terrain.grid_vertices) only: it has no BVH walk";
  for (int l = 0; l < NLEG; ++l) if (model->cp_count[l] > CH_NCP) return "the 2 x 6 instance holds four collision spheres per leg";
#endif
  if (cfg->num_obs != NPROP + (cfg->measure_heights ? cfg->num_height_points : 0) + cfg->num_extra_obs) return "num_obs does not match the observation layout";
  if (model->num_bodies != 1 + NLEG * (NJ + model->has_foot_body) || model->num_bodies > NBODY_MAX) return "unsupported body count";
  if (cfg->decimation <= 0 || cfg->sim_dt <= 0.f) return "bad dt / decimation";
  if (cfg->resampling_steps <= 0) return "resampling_steps must be positive";
  if (cfg->push_robots && cfg->push_interval <= 0) return "push_interval must be positive";
  if (ter->mesh_type < LG_MESH_PLANE || ter->mesh_type > LG_MESH_TRIMESH) return "unknown terrain mesh_type";
  if (ter->mesh_type != LG_MESH_PLANE && (ter->rows < 2 || ter->cols < 2 || !ter->height_samples)) return "rough terrain without height samples";
  if (ter->mesh_type == LG_MESH_HEIGHTFIELD && (ter->rows > 65535 || ter->cols > 32767)) {       // capsule edge pieces travel as L | j << 16 (lg_physics.h, ContactProbeC::lj)
    for (int l = 0; l < model->num_legs; ++l) for (int s = 0; s < model->cp_count[l]; ++s)
      if (model->cp_slide[l][s][0] != 0.f || model->cp_slide[l][s][1] != 0.f || model->cp_slide[l][s][2] != 0.f) return "height grid too large for the capsule-segment instance (rows <= 65535, cols <= 32767)";
  }
  if (ter->mesh_type == LG_MESH_TRIMESH && !ter->collision_mesh) return "trimesh terrain without a collision mesh (lg_mesh_create)";
  if (cfg->curriculum && (ter->num_levels <= 0 || ter->num_types <= 0 || !ter->terrain_origins)) return "curriculum needs terrain_origins";
  if (model->num_penalised < 0 || model->num_penalised > NBODY_MAX || model->num_termination < 0 || model->num_termination > NBODY_MAX)
    return "more penalised / termination bodies than the robot has";
  for (int l = 0; l < NLEG; ++l) if (model->cp_count[l] < 0 || model->cp_count[l] > LG_MAX_CP) return "bad cp_count";
  if (model->num_sc_pairs < 0 || model->num_sc_pairs > LG_MAX_SC_PAIRS) return "bad num_sc_pairs";
  for (int i = 0; i < model->num_sc_pairs; ++i) {
    const int32_t* q = model->sc_pairs[i];
    if (q[0] < 0 || q[0] >= NLEG || q[2] < 0 || q[2] >= NLEG || q[1] < 0 || q[1] >= model->cp_count[q[0]] || q[3] < 0 || q[3] >= model->cp_count[q[2]])
      return "sc_pairs names a collision sphere the model does not have";
  }
  for (int k = 0; k < cfg->num_reward_terms; ++k) if (cfg->reward_term_ids[k] < 0 || cfg->reward_term_ids[k] >= LG_REW_COUNT) return "unknown reward term id";
  for (int k = 0; k < cfg->num_reward_terms; ++k)
    if (NLEG != 4 && cfg->reward_term_ids[k] == LG_REW_PENALTY_IN_THE_AIR) return "penalty_in_the_air is the four-legged stand classes' term (feet 1 and 3 of four)";
  if (cfg->reward_class == LG_RC_STAND && NLEG < 4) return "the stand reward class reads feet 1 and 3: it needs at least four legs";
  if (!cfg->noise_scale_vec) return "noise_scale_vec is null";
  if (cfg->num_height_points > 0 && !cfg->height_points) return "height_points is null";
  return nullptr;
}

#define HIP_TRY(c, expr)                                                                         \
  do { hipError_t _e = (expr);                                                                   \
       if (_e != hipSuccess) { (c)->err = std::string(#expr) + ": " + hipGetErrorString(_e); return LG_ERR_HIP; } } while (0)

extern "C" {

void lg_abi_sizes(int32_t out[4]) {
  out[0] = LG_ABI_VERSION; out[1] = (int32_t)sizeof(lg_config); out[2] = (int32_t)sizeof(lg_robot_model); out[3] = (int32_t)sizeof(lg_terrain);
}

size_t lg_arena_bytes(const lg_config* cfg, const lg_robot_model* model, const lg_terrain* terrain) {
  if (validate(cfg, model, terrain)) return 0;
  TensorInfo t[LG_T_COUNT];
  return build_layout(cfg, model, terrain, t);
}

const char* lg_last_error(lg_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

// diagnostic builds only: copy out the 16 phase counters (not part of the public ABI)
int lg_debug_read_stamps(lg_ctx* c, unsigned long long out[64]) {
  if (!c) return LG_ERR_INVALID;
  if (hipDeviceSynchronize() != hipSuccess) return LG_ERR_HIP;
  return hipMemcpy(out, c->h.stamps, 512, hipMemcpyDeviceToHost) == hipSuccess ? LG_OK : LG_ERR_HIP;
}

void lg_destroy(lg_ctx* c) {
  if (!c) return;
  DeviceScope ds_(c->device);
  if (c->d) (void)hipFree(c->d);
  if (c->aux) (void)hipFree(c->aux);
  if (c->obs_tab) (void)hipFree(c->obs_tab);
  if (c->mesh_cache) (void)hipFree(c->mesh_cache);
  if (c->grid_verts) (void)hipFree(c->grid_verts);
  if (c->hmin) (void)hipFree(c->hmin);
  if (c->own_arena && c->arena) (void)hipFree(c->arena);
  for (auto e : c->ev) (void)hipEventDestroy(e);
  delete c;
}

lg_ctx* lg_create(const lg_config* cfg, const lg_robot_model* model, const lg_terrain* ter, int device_id, void* arena) {
  if (const char* why = validate(cfg, model, ter)) { g_err = why; return nullptr; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_err = "no HIP device: the env step has no CPU path"; return nullptr; }
  if (device_id < 0 || device_id >= ndev) { g_err = "device_id out of range"; return nullptr; }
  DeviceScope ds_(device_id);                      // (restores the caller's current device on every return path)
  if (!ds_.ok) { g_err = "hipSetDevice failed"; return nullptr; }
  lg_ctx* c = new lg_ctx();
  c->device = device_id;
  c->arena_bytes = build_layout(cfg, model, ter, c->t);
  auto fail = [&](const std::string& why) { g_err = why; lg_destroy(c); return (lg_ctx*)nullptr; };
  if (arena) c->arena = arena;
  else { if (hipMalloc(&c->arena, c->arena_bytes) != hipSuccess) return fail("hipMalloc(arena) failed"); c->own_arena = true; }
  if (hipMemset(c->arena, 0, c->arena_bytes) != hipSuccess) return fail("hipMemset(arena) failed");

  DevCtx& h = c->h;
  memset(&h, 0, sizeof(h));
  h.cfg = *cfg; h.model = *model;
  h.cfg.noise_scale_vec = nullptr; h.cfg.height_points = nullptr;
  h.N = cfg->num_envs; h.B = model->num_bodies; h.K = cfg->num_reward_terms; h.P = cfg->num_height_points;
  h.per_leg = NJ + model->has_foot_body;
  pack_lstm_weights(h.lstm_w, cfg->actuator_net);
  reward_meta(h);
  hot_config(h);
  h.slide_mask = 0u;
  for (int l = 0; l < NLEG; ++l)
    for (int sl = 0; sl < model->cp_count[l]; ++sl)
      if (model->cp_slide[l][sl][0] != 0.f || model->cp_slide[l][sl][1] != 0.f || model->cp_slide[l][sl][2] != 0.f) h.slide_mask |= 1u << sl;
  if (const char* ev = getenv("LG_CAPS")) { if (atoi(ev) == 0) h.slide_mask = 0u; }      // (diagnostic / A-B: every sphere stays in the middle of its part)
  if (ter->mesh_type == LG_MESH_PLANE) h.slide_mask = 0u;       // a plane has no edges: the self-collision instances (which carry the capsule code) skip it at run time (config 5: -2 %)
  // Who detects which slot (capsule instances on a height grid).  A slot with a segment costs about twice a plain one (two edge pieces), an empty slot
  // nothing; the waves take positions 0-2 (main) / 3 (wave 1, which also has the leg bias) / 4-5 / 6-7 -- the quadruped's capsule instance since round 6:
  // 0-1 / 2 / 3-5 / 6-7 (LG_CAPS_DEAL).  Segment slots go to wave 3's first position, then the main wave's last, then wave 2's; slots no leg has fill up
  // behind them (wave 3's second position first); plain slots take what is left in ascending order.  ANYmal-C (foot, three shank spheres of which two carry
  // segments, two thigh-drive spheres, a trunk sphere): {0, 3 | 1 | 4, 5, 6 | 2, -}.  LG_DEAL=0: the identity (A/B).
  {
    int max_cp = 0;
    for (int l = 0; l < NLEG; ++l) max_cp = model->cp_count[l] > max_cp ? model->cp_count[l] : max_cp;
    int at[LG_MAX_CP]; bool used[LG_MAX_CP] = {false};
    for (int p = 0; p < LG_MAX_CP; ++p) at[p] = -1;
#if LG_CAPS_DEAL      // positions 0-1 main / 2 wave 1 / 3-5 wave 2 / 6-7 wave 3: ANYmal-C {0, 3 | 1 | 4, 5, 6 | 2, -}
    const int seg_pos[LG_MAX_CP] = {6, 1, 3, 4, 7, 0, 5, 2}, empty_pos[LG_MAX_CP] = {7, 5, 2, 4, 3, 6, 1, 0};
#else
    const int seg_pos[LG_MAX_CP] = {6, 2, 4, 5, 7, 0, 1, 3}, empty_pos[LG_MAX_CP] = {7, 3, 5, 1, 6, 4, 2, 0};
#endif
    int ns = 0;
    for (int sl = 0; sl < max_cp; ++sl) if ((h.slide_mask >> sl) & 1u) { at[seg_pos[ns++]] = sl; used[sl] = true; }
    int ne = 0;
    for (int sl = max_cp; sl < LG_MAX_CP; ++sl) { while (at[empty_pos[ne]] >= 0) ++ne; at[empty_pos[ne]] = sl; used[sl] = true; }
    int pp = 0;
    for (int sl = 0; sl < max_cp; ++sl) if (!used[sl]) { while (at[pp] >= 0) ++pp; at[pp] = sl; }
    h.slot_perm = 0u;
    for (int p = 0; p < LG_MAX_CP; ++p) h.slot_perm |= (unsigned)at[p] << (4 * p);
    if (h.slide_mask == 0u) h.slot_perm = 0x76543210u;
    if (const char* ev = getenv("LG_DEAL")) {            // (A/B: "0" = the identity; eight digits = the slots at positions 0..7)
      if (strlen(ev) == 8) { h.slot_perm = 0u; for (int p = 0; p < 8; ++p) h.slot_perm |= (unsigned)((ev[p] - '0') & 7) << (4 * p); }
      else if (atoi(ev) == 0) h.slot_perm = 0x76543210u;
    }
  }
  // Triangle-mesh terrains: every wave answers the closest-point queries of a PAIR of positions per substep, and the step waits for the slowest pair at
  // rendezvous (A2).  LG_MESH_DEAL: eight digits = the slots at positions 0..7 (wave 2 takes positions 0-1, wave 1 2-3, wave 3 4-5, the main wave 6-7: MESH_PAIR0).
  // Measured on config 3 (A1, seven slots: foot, two shank, two thigh spheres, two trunk corners; profiles/r06_schedule_experiments.txt): with the spheres in model order
  // -- foot and lowest shank sphere on one wave -- 0.392 ms per step; with ONE ground-near sphere per wave 0.363-0.367 (fifteen deals tried: the level is the same
  // whichever far sphere joins which near one).  So: slots ranked by how near the ground they work (distal link first, then lower in the link), ranks 0-3 lead the four
  // pairs (wave 2 / wave 1 / wave 3 / main wave), ranks 4-7 fill them in the same order; a robot of seven slots leaves the main wave with one query.
  // LG_MESH_DEAL: eight digits = the slots at positions 0..7; "0" = model order (A/B, tests).
  {
    int max_cp = 0, lref = 0;
    for (int l = 0; l < NLEG; ++l) if (model->cp_count[l] > max_cp) { max_cp = model->cp_count[l]; lref = l; }
    int rk[LG_MAX_CP];
    for (int i = 0; i < LG_MAX_CP; ++i) rk[i] = i;
    std::stable_sort(rk, rk + LG_MAX_CP, [&](int a, int b) {
      const bool ea = a < max_cp, eb = b < max_cp;
      if (ea != eb) return ea;
      if (!ea) return false;
      const int la = model->cp_link[lref][a], lb = model->cp_link[lref][b];
      if (la != lb) return la > lb;
      return model->cp_pos[lref][a][2] < model->cp_pos[lref][b][2];
    });
    h.mesh_perm = 0u;
    for (int i = 0; i < 4; ++i) h.mesh_perm |= (unsigned)rk[i] << (4 * (2 * i)) | (unsigned)rk[4 + i] << (4 * (2 * i + 1));
  }
  if (const char* ev = getenv("LG_MESH_DEAL")) {
    if (strlen(ev) == 8) { h.mesh_perm = 0u; for (int p = 0; p < 8; ++p) h.mesh_perm |= (unsigned)((ev[p] - '0') & 7) << (4 * p); }
    else if (atoi(ev) == 0) h.mesh_perm = 0x76543210u;
  }
  h.mesh_reach = LG_MESH_CACHE_REACH;
  if (const char* ev = getenv("LG_MESH_REACH")) { const float v = (float)atof(ev); if (v >= 0.f && v <= 1.f) h.mesh_reach = v; }
  h.n_sc = cfg->self_collisions ? model->num_sc_pairs : 0;
  for (int i = 0; i < h.n_sc; ++i) {
    const int32_t* q = model->sc_pairs[i];
    h.sc_pairs[i] = (unsigned)q[0] | (unsigned)q[1] << 8 | (unsigned)q[2] << 16 | (unsigned)q[3] << 24;
#if NJ == 3
    const float ra = model->cp_radius[q[0]][q[1]], rb = model->cp_radius[q[2]][q[3]];
    const float reach = ra + rb + cfg->contact_offset, thr2 = reach * reach * 1.0001f;
    unsigned ua, ub, ut; memcpy(&ua, &ra, 4); memcpy(&ub, &rb, 4); memcpy(&ut, &thr2, 4);
    const unsigned oa = (unsigned)(q[1] * 64 + q[0]), ob = (unsigned)(q[3] * 64 + q[2]);      // record index slot * 64 + leg (the lane group's first lane is added in the kernel)
    h.sc_tab[i] = make_uint4(oa | ob << 16, ua, ub, ut);
#endif
  }
  h.mesh_cache = nullptr;
  if (ter->mesh_type == LG_MESH_TRIMESH) {
    const size_t nf = (size_t)cfg->num_envs * NLEG * LG_MAX_CP * 4;
    std::vector<float> init(nf, -1.f);                   // distance < 0: no entry
    if (hipMalloc((void**)&c->mesh_cache, nf * 4) != hipSuccess ||
        hipMemcpy(c->mesh_cache, init.data(), nf * 4, hipMemcpyHostToDevice) != hipSuccess) return fail("mesh contact cache allocation failed");
    h.mesh_cache = (float LG_G*)c->mesh_cache;
  }
  h.terrain_mu = ter->static_friction; h.env_length = ter->env_length; h.num_levels = ter->num_levels; h.num_types = ter->num_types;
  char* base = (char*)c->arena;
  auto P = [&](int id) { return (void*)(base + c->t[id].off); };
  h.root = (float LG_G*)P(LG_T_ROOT_STATES); h.dof = (float LG_G*)P(LG_T_DOF_STATE); h.rigid = (float LG_G*)P(LG_T_RIGID_BODY_STATE);
  h.cforce = (float LG_G*)P(LG_T_CONTACT_FORCES); h.torques = (float LG_G*)P(LG_T_TORQUES); h.actions = (float LG_G*)P(LG_T_ACTIONS);
  h.last_actions = (float LG_G*)P(LG_T_LAST_ACTIONS); h.last_dof_vel = (float LG_G*)P(LG_T_LAST_DOF_VEL); h.last_root_vel = (float LG_G*)P(LG_T_LAST_ROOT_VEL);
  h.commands = (float LG_G*)P(LG_T_COMMANDS); h.base_lin_vel = (float LG_G*)P(LG_T_BASE_LIN_VEL); h.base_ang_vel = (float LG_G*)P(LG_T_BASE_ANG_VEL);
  h.proj_grav = (float LG_G*)P(LG_T_PROJECTED_GRAVITY); h.base_lin_acc = (float LG_G*)P(LG_T_BASE_LIN_ACC); h.base_ang_acc = (float LG_G*)P(LG_T_BASE_ANG_ACC);
  h.feet_air = (float LG_G*)P(LG_T_FEET_AIR_TIME); h.feet_ctime = (float LG_G*)P(LG_T_FEET_CONTACT_TIME); h.last_contacts = (uint8_t LG_G*)P(LG_T_LAST_CONTACTS);
  h.heights = (float LG_G*)P(LG_T_MEASURED_HEIGHTS); h.obs = (float LG_G*)P(LG_T_OBS_BUF); h.rew = (float LG_G*)P(LG_T_REW_BUF);
  h.reset_buf = (uint8_t LG_G*)P(LG_T_RESET_BUF); h.time_out = (uint8_t LG_G*)P(LG_T_TIME_OUT_BUF); h.ep_len = (int64_t LG_G*)P(LG_T_EPISODE_LENGTH_BUF);
  h.ep_sums = (float LG_G*)P(LG_T_EPISODE_SUMS); h.levels = (int64_t LG_G*)P(LG_T_TERRAIN_LEVELS); h.types = (int64_t LG_G*)P(LG_T_TERRAIN_TYPES);
  h.origins = (float LG_G*)P(LG_T_ENV_ORIGINS); h.friction = (float LG_G*)P(LG_T_FRICTION_COEFFS); h.mass_added = (float LG_G*)P(LG_T_BASE_MASS_ADDED);
  h.sea_h = (float LG_G*)P(LG_T_SEA_HIDDEN_STATE); h.sea_c = (float LG_G*)P(LG_T_SEA_CELL_STATE); h.gait_idx = (float LG_G*)P(LG_T_GAIT_IDX);
  h.gait_foot_z = (float LG_G*)P(LG_T_GAIT_FOOT_Z); h.extras = (float LG_G*)P(LG_T_EXTRAS_EPISODE); h.rand_inject = (float LG_G*)P(LG_T_RAND_INJECT);
  h.counters = (int64_t LG_G*)P(LG_T_STEP_COUNTERS); h.cmd_ranges = (float LG_G*)P(LG_T_COMMAND_RANGES); h.ep_stats = (double LG_G*)P(LG_T_EPISODE_STATS); h.terrain_origins = (const float LG_G*)P(LG_T_TERRAIN_ORIGINS);
  h.ter.mesh_type = ter->mesh_type; h.ter.rows = ter->rows; h.ter.cols = ter->cols;
  h.ter.hscale = ter->horizontal_scale; h.ter.vscale = ter->vertical_scale; h.ter.border = ter->border_size;
  h.ter.H = (const int16_t LG_G*)P(LG_T_HEIGHT_SAMPLES);
  h.ter.L = LatticeView{nullptr, nullptr, nullptr, 0, 0, 0.f, 0.f, 1.f, 1.f, LATP_CAP};
  h.ter.M = MeshView{nullptr, nullptr}; h.ter.GV = nullptr; h.ter.GV4 = nullptr; h.ter.GM = nullptr; h.ter.mcols = 0; h.ter.SEG4 = nullptr;
  if (const char* ev = getenv("LG_GRID_MESH")) c->grid_mesh = atoi(ev) != 0;
  bool mesh_caps = true;                                   // LG_MESH_CAPS=0: the spheres alone on grid meshes (A/B, the tests' checker)
  if (const char* ev = getenv("LG_MESH_CAPS")) mesh_caps = atoi(ev) != 0;
  if (ter->mesh_type == LG_MESH_TRIMESH && ter->grid_vertices && (c->grid_mesh || mesh_caps)) {      // grid mesh: contact queries by cell index, capsule segments against its edges
    // vertices as (x, y, z, 0) -- one 16-byte load each --, then the max z of every 2 x 2 block of vertices (the clearance test in closest_point_grid)
    const size_t nvert = (size_t)ter->rows * ter->cols, nv = nvert * 4;
    const int mr = (ter->rows + 1) / 2, mc = (ter->cols + 1) / 2;
    std::vector<float> v4(nv, 0.f), top((size_t)mr * mc, -1e30f);
    std::vector<uint8_t> moved((size_t)mr * mc, 0);       // block holds a vertex the slope correction moved off its lattice position
    for (int i = 0; i < ter->rows; ++i)
      for (int j = 0; j < ter->cols; ++j) {
        const size_t k = (size_t)i * ter->cols + j;
        for (int a = 0; a < 3; ++a) v4[4 * k + a] = ter->grid_vertices[3 * k + a];
        float& t = top[(size_t)(i >> 1) * mc + (j >> 1)];
        t = std::max(t, ter->grid_vertices[3 * k + 2]);
        const float nx = (float)i * ter->horizontal_scale - ter->border_size, ny = (float)j * ter->horizontal_scale - ter->border_size, tol = 1e-3f * ter->horizontal_scale;
        if (std::fabs(ter->grid_vertices[3 * k] - nx) > tol || std::fabs(ter->grid_vertices[3 * k + 1] - ny) > tol) moved[(size_t)(i >> 1) * mc + (j >> 1)] = 1;
      }
    // the flag rides in the last mantissa bit of the block's height, which only ever moves UP for it (the height is an upper bound: one ulp more is still one)
    for (size_t b = 0; b < top.size(); ++b) {
      uint32_t u; memcpy(&u, &top[b], 4);
      if ((u & 1u) != (uint32_t)moved[b]) { top[b] = std::nextafter(top[b], 1e38f); }
    }
    if (hipMalloc((void**)&c->grid_verts, (nv + top.size()) * sizeof(float)) != hipSuccess ||
        hipMemcpy(c->grid_verts, v4.data(), nv * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy((float*)c->grid_verts + nv, top.data(), top.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
      return fail("grid-mesh vertex upload failed");
    if (c->grid_mesh) { h.ter.GV = (const float*)c->grid_verts; h.ter.GV4 = (const float4*)c->grid_verts; h.ter.GM = (const float*)c->grid_verts + nv; h.ter.mcols = mc; }
    if (mesh_caps && ter->rows <= 65535 && ter->cols <= 32767) h.ter.SEG4 = (const float4*)c->grid_verts;      // (caps_edge_piece's packed indices; larger grids: spheres alone)
  }
  if (ter->mesh_type == LG_MESH_TRIMESH) {
    if (ter->collision_mesh->device != device_id) return fail("collision mesh lives on another device");
    h.ter.M = MeshView{ter->collision_mesh->d_nodes, ter->collision_mesh->d_tris};
    if (const lg_mesh* cm = ter->collision_mesh; cm->d_gcz && cm->d_gcr && !h.ter.GV)      // a lattice mesh: contact queries by cell (closest_point_lattice)
    {
      int cap = LATP_CAP;                                  // LG_LATTICE_CAP: a smaller query table (never below the longest run of faces: a run must fit an empty table)
      if (const char* ev = getenv("LG_LATTICE_CAP")) cap = std::min(LATP_CAP, std::max(atoi(ev), std::max(cm->gmaxrun, 1)));
      h.ter.L = LatticeView{cm->d_gcz, cm->d_gcr, cm->d_gtris, cm->gnx, cm->gny, cm->gx0, cm->gy0, cm->ghx, cm->ghy, cap};
    }
    for (int k = 0; k < 3; ++k) { h.mesh_lo[k] = ter->collision_mesh->bmin[k]; h.mesh_hi[k] = ter->collision_mesh->bmax[k]; }
  }
  h.nblocks_post = (h.N + EPBP - 1) / EPBP;
  h.n_stepped = h.N;

  // aux buffer: noise_vec | height_points | partials
  size_t n_noise = (size_t)cfg->num_obs, n_hp = (size_t)2 * cfg->num_height_points;
  size_t n_part = (size_t)h.nblocks_post * PART_STRIDE;
  size_t n_fin = (size_t)2 * h.nblocks_post + 4 + 32 + 9 * 32 + 2 * PART_STRIDE + 2;   // level sums | reset flags | (pad to a 128-B line) | arrival counters | accumulators
  size_t aux_floats = n_noise + n_hp + n_part + 130 + n_fin;   // + 64 x u64 stamp counters
  if (hipMalloc(&c->aux, aux_floats * 4) != hipSuccess) return fail("hipMalloc(aux) failed");
  if (hipMemset(c->aux, 0, aux_floats * 4) != hipSuccess) return fail("hipMemset(aux) failed");
  float* aux = (float*)c->aux;
  h.noise_vec = (const float LG_G*)aux; h.height_points = (const float LG_G*)(aux + n_noise); h.partials = (float LG_G*)(aux + n_noise + n_hp);
  h.stamps = (unsigned long long LG_G*)(aux + ((n_noise + n_hp + n_part + 1) & ~(size_t)1));
  {
    float* fin = aux + n_noise + n_hp + n_part + 130;
    h.lvl_part = (float LG_G*)fin; h.part_flag = (unsigned LG_G*)(fin + h.nblocks_post);
    h.tickets = (unsigned LG_G*)(((uintptr_t)(fin + 2 * h.nblocks_post + 4) + 127) & ~(uintptr_t)127);
    h.acc = (long long LG_G*)(h.tickets + 9 * 32);
  }
  if (hipMemcpy(aux, cfg->noise_scale_vec, n_noise * 4, hipMemcpyHostToDevice) != hipSuccess) return fail("copy noise_scale_vec failed");
  {
    std::vector<float> tab = pack_obs_table(*cfg, h.P);
    if (hipMalloc(&c->obs_tab, tab.size() * 4) != hipSuccess) return fail("hipMalloc(obs_tab) failed");
    if (hipMemcpy(c->obs_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return fail("copy obs_tab failed");
    h.obs_tab = (const float4 LG_G*)c->obs_tab;
  }
  if (n_hp && hipMemcpy(aux + n_noise, cfg->height_points, n_hp * 4, hipMemcpyHostToDevice) != hipSuccess) return fail("copy height_points failed");
  if (ter->mesh_type != LG_MESH_PLANE &&
      hipMemcpy(P(LG_T_HEIGHT_SAMPLES), ter->height_samples, (size_t)ter->rows * ter->cols * 2, hipMemcpyHostToDevice) != hipSuccess)
    return fail("copy height_samples failed");
  {   // the height scan's table: per cell the minimum of the three samples LR:929-936 takes (the last row / column are never indexed: the
      // scan clamps its cell to rows - 2, cols - 2); the plane's 1 x 1 dummy grid is its own minimum
    const int R = ter->mesh_type != LG_MESH_PLANE ? ter->rows : 1, Cc = ter->mesh_type != LG_MESH_PLANE ? ter->cols : 1;
    std::vector<int16_t> hm((size_t)R * Cc, 0);
    if (ter->mesh_type != LG_MESH_PLANE)
      for (int i = 0; i < R; ++i)
        for (int j = 0; j < Cc; ++j) {
          const int16_t* Hs = ter->height_samples;
          int16_t v = Hs[(size_t)i * Cc + j];
          if (i + 1 < R) v = std::min(v, Hs[(size_t)(i + 1) * Cc + j]);
          if (j + 1 < Cc) v = std::min(v, Hs[(size_t)i * Cc + j + 1]);
          hm[(size_t)i * Cc + j] = v;
        }
    if (hipMalloc(&c->hmin, hm.size() * 2) != hipSuccess || hipMemcpy(c->hmin, hm.data(), hm.size() * 2, hipMemcpyHostToDevice) != hipSuccess)
      return fail("height-scan table upload failed");
    h.ter.Hmin = (const int16_t LG_G*)c->hmin;
  }
  if (ter->num_levels > 0 && ter->terrain_origins &&
      hipMemcpy(P(LG_T_TERRAIN_ORIGINS), ter->terrain_origins, (size_t)ter->num_levels * ter->num_types * 12, hipMemcpyHostToDevice) != hipSuccess)
    return fail("copy terrain_origins failed");
  {
    float cr[8] = {cfg->cmd_lin_vel_x[0], cfg->cmd_lin_vel_x[1], cfg->cmd_lin_vel_y[0], cfg->cmd_lin_vel_y[1],
                   cfg->cmd_ang_vel_yaw[0], cfg->cmd_ang_vel_yaw[1], cfg->cmd_heading[0], cfg->cmd_heading[1]};
    if (hipMemcpy(P(LG_T_COMMAND_RANGES), cr, sizeof(cr), hipMemcpyHostToDevice) != hipSuccess) return fail("copy command ranges failed");
  }
  // initial values: identity base quaternion, reset_buf = 1 (base_task.py:73)
  {
    std::vector<float> root((size_t)h.N * 13, 0.f);
    for (int e = 0; e < h.N; ++e) root[(size_t)e * 13 + 6] = 1.f;
    if (hipMemcpy(h.root, root.data(), root.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return fail("init root_states failed");
    if (hipMemset(h.reset_buf, 1, (size_t)h.N) != hipSuccess) return fail("init reset_buf failed");
  }
  if (hipMalloc((void**)&c->d, sizeof(DevCtx)) != hipSuccess) return fail("hipMalloc(ctx) failed");
  if (hipMemcpy(c->d, &h, sizeof(DevCtx), hipMemcpyHostToDevice) != hipSuccess) return fail("copy ctx failed");
  if (hipDeviceSynchronize() != hipSuccess) return fail("device sync failed");
  if (const char* ev = getenv("LG_SPLIT")) c->split = atoi(ev) != 0;
  if (const char* ev = getenv("LG_FUSE")) c->fuse = atoi(ev) != 0;
  if (const char* ev = getenv("LG_PERSIST")) c->persist = atoi(ev) != 0;
  if (const char* ev = getenv("LG_GFUSE")) c->gfuse = atoi(ev) != 0;
  if (const char* ev = getenv("LG_SPEC")) c->spec = atoi(ev) != 0;
  return c;
}

int lg_get_tensor(lg_ctx* c, int id, void** dptr, int64_t shape[4], int32_t* ndim, int32_t* dtype) {
  if (!c) return LG_ERR_INVALID;
  if (id < 0 || id >= LG_T_COUNT) { c->err = "tensor id out of range"; return LG_ERR_INVALID; }
  *dptr = (char*)c->arena + c->t[id].off;
  for (int i = 0; i < 4; ++i) shape[i] = c->t[id].shape[i];
  *ndim = c->t[id].ndim; *dtype = c->t[id].dtype;
  return LG_OK;
}

static __global__ void set_n_stepped(DevCtx* C, int n) { C->n_stepped = n; }
// sum of all terrain levels before a subset step (exact in float up to 2^24)
static __global__ __launch_bounds__(256) void level_total_kernel(DevCtx* C) {
  __shared__ float red[256];
  float s = 0.f;
  for (int e = threadIdx.x; e < C->N; e += 256) s += (float)C->levels[e];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) { if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off]; __syncthreads(); }
  if (threadIdx.x == 0) C->lvl_total_before = red[0];
}

static int launch_post(lg_ctx* c, hipStream_t st, hipEvent_t* ev, const int32_t* ids, int n, int mode, float* rew_out = nullptr, int rew_stride = 0,
                       PostSink sink = PostSink{nullptr, nullptr, nullptr, nullptr, 0.f}) {
  const int nb = (n + EPBP - 1) / EPBP;
  if (n != c->h.n_stepped) { c->h.n_stepped = n; hipLaunchKernelGGL(set_n_stepped, dim3(1), dim3(1), 0, st, c->d, n); }   // (a post step without its physics launch)
  if (ids && mode == 0 && c->h.cfg.curriculum) hipLaunchKernelGGL(level_total_kernel, dim3(1), dim3(256), 0, st, c->d);
  // a wave per env in the wide stages: 64 * EPBP threads (the hexapod's two envs per workgroup ran on 256 threads until round 6 -- two of the four waves
  // shadowed env 0 and stored nothing, and took the slots of another workgroup: 2048 workgroups in two rounds of four per CU instead of one round of eight)
  hipLaunchKernelGGL(post_kernel, dim3(nb), dim3(64 * EPBP), 0, st, c->d, ids, n, mode, rew_out, rew_stride, sink);
  if (ev) (void)hipEventRecord(ev[2], st);
  if (ev) (void)hipEventRecord(ev[3], st);
  HIP_TRY(c, hipGetLastError());
  return LG_OK;
}

// fuse: the post-physics step runs as the tail of the physics kernel (full steps of all envs with helper waves; LG_FUSE=0 keeps
// the two-launch path, which every split / subset / rollout entry point uses anyway)
static bool can_fuse(const lg_ctx* c) { return LG_LEGS == 4 && c->fuse && !c->h.extra_term && !c->h.cfg.keep_small_commands && !c->h.cfg.feet_air_time_ungated && (c->split || c->h.ter.mesh_type == LG_MESH_TRIMESH) && c->h.P <= MAX_P; }
// The six-legged instance can end lg_step / lg_step_transition in the physics launch too, through the post kernel's own code (generic_fused_tail): OFF unless
// LG_GFUSE=1.  Measured (round 5, 4096 envs, same session): 0.159 / 0.162 ms in one launch against 0.158 / 0.153 ms in two -- the post-physics chain is latency-bound,
// and on one wave per SIMD it takes as long as the separate kernel (eight workgroups per CU) plus its launch gap; round 1 found the same for the quadruped.  The
// path stays as the bit-exact checker of a future hand-tuned tail (tests/test_elspider.py).
static bool can_gfuse(const lg_ctx* c) { return LG_LEGS == 6 && NJ == 3 && c->gfuse && c->fuse && c->split && c->h.ter.mesh_type != LG_MESH_TRIMESH && !c->h.cfg.inject_sim_state; }
// fuse: 0 = physics only (a post kernel follows), 1 = full policy step with the fused tail, 2 = fused ROLLOUT step of the listed envs, 3 = sink.nsteps of them in one launch
// envs per workgroup of the chain instance's physics kernel: EPW, halved while the launch would have fewer workgroups than the chip has CUs (LG_CHAIN_EPB sets it)
static int chain_epb(int n) {
  int epb = EPW;
  while (epb > 4 && (n + epb - 1) / epb < 256) epb >>= 1;
  if (const char* ev = getenv("LG_CHAIN_EPB")) { const int v = atoi(ev); if (v >= 1 && v <= EPW) epb = v; }
  return epb;
}
static void launch_physics(lg_ctx* c, hipStream_t st, const float* actions, const int32_t* ids, int n, int act_stride = NDOF, int fuse = 0,
                           PostSink sink = PostSink{nullptr, nullptr, nullptr, nullptr, 0.f}) {
#if NJ != 3
  (void)fuse; (void)sink;
  const int epb = chain_epb(n), nb = (n + epb - 1) / epb;
  if (n != c->h.n_stepped) { c->h.n_stepped = n; hipLaunchKernelGGL(set_n_stepped, dim3(1), dim3(1), 0, st, c->d, n); }
  // (helper waves for the contact detection: LG_SPLIT=0 keeps the single-wave launch, the checker of that path)
  if (c->split) {
    if (c->h.ter.mesh_type == LG_MESH_TRIMESH) hipLaunchKernelGGL((physics_kernel_chain<0, true, true>), dim3(nb), dim3(256), 0, st, c->d, actions, c->h.cfg.decimation, ids, n, act_stride, epb);
    else hipLaunchKernelGGL((physics_kernel_chain<0, false, true>), dim3(nb), dim3(256), 0, st, c->d, actions, c->h.cfg.decimation, ids, n, act_stride, epb);
  } else if (c->h.ter.mesh_type == LG_MESH_TRIMESH) hipLaunchKernelGGL((physics_kernel_chain<0, true>), dim3(nb), dim3(64), 0, st, c->d, actions, c->h.cfg.decimation, ids, n, act_stride, epb);
  else hipLaunchKernelGGL((physics_kernel_chain<0, false>), dim3(nb), dim3(64), 0, st, c->d, actions, c->h.cfg.decimation, ids, n, act_stride, epb);
  return;
#else
  const int nb = (n + EPB - 1) / EPB;
  // helper waves (leg bias, contact detection, a share of the contact set-up; with the actuator network also its three
  // joints per leg): always, unless LG_SPLIT=0 (diagnostic) -- and even then on triangle-mesh terrains, whose contact
  // detection is a BVH traversal per collision sphere that should not sit on the main wave.  PD-controlled robots gain
  // as well: the main wave alone took 0.134 ms per rollout step of 4096 envs on the plane
  const int nact = (c->split || c->h.ter.mesh_type == LG_MESH_TRIMESH) ? 3 : 0;
  if (n != c->h.n_stepped) { c->h.n_stepped = n; hipLaunchKernelGGL(set_n_stepped, dim3(1), dim3(1), 0, st, c->d, n); }
  // Which instance (physics_kernel's SPEC): rollout tail or not, and the model features the launch needs -- capsule segments (height grids against their grid lines, grid meshes against their own edges; other meshes and planes: spheres alone)
  // and the self-collision pass.  A robot of fixed spheres without self-collision runs
  // the plain instance.
  const bool tm = c->h.ter.mesh_type == LG_MESH_TRIMESH;
  const bool caps = c->h.ter.mesh_type == LG_MESH_HEIGHTFIELD && c->h.slide_mask != 0u, selfc = c->h.n_sc > 0;       // (a plane has no grid lines: the plain instance)
  const bool mcaps = tm && c->h.ter.SEG4 && c->h.slide_mask != 0u;      // grid meshes: the segments against the mesh's edges (other meshes: spheres alone)
#define LG_LAUNCH_PK(TM, HELP, SPEC_, THREADS) \
  hipLaunchKernelGGL((physics_kernel<0, TM, HELP, SPEC_>), dim3(nb), dim3(THREADS), 0, st, c->d, actions, c->h.cfg.decimation, nact, ids, n, act_stride, fuse, sink)
#if LG_LEGS == 4
  if (fuse == 3 && !tm) {                                // persistent rollout launch: sink.nsteps steps (heightfield / plane terrains; mesh terrains go step by step)
    if (selfc) LG_LAUNCH_PK(false, true, 3 + 12, 256);
    else if (caps) LG_LAUNCH_PK(false, true, 3 + 4, 256);
    else LG_LAUNCH_PK(false, true, 3, 256);
    return;
  }
  if (fuse == 2) {                                       // (can_fuse() held: helper waves are present)
    if (tm) { if (selfc) { if (mcaps) LG_LAUNCH_PK(true, true, 2 + 12, 256); else LG_LAUNCH_PK(true, true, 2 + 8, 256); } else if (mcaps) LG_LAUNCH_PK(true, true, 2 + 4, 256); else LG_LAUNCH_PK(true, true, 2, 256); }
    else if (selfc) LG_LAUNCH_PK(false, true, 2 + 12, 256);
    else if (caps) LG_LAUNCH_PK(false, true, 2 + 4, 256);
    else LG_LAUNCH_PK(false, true, 2, 256);
    return;
  }
#endif
  if (tm) { if (selfc) { if (mcaps) LG_LAUNCH_PK(true, true, 12, 256); else LG_LAUNCH_PK(true, true, 8, 256); } else if (mcaps) LG_LAUNCH_PK(true, true, 4, 256); else LG_LAUNCH_PK(true, true, 0, 256); }
  else
#if LG_AB == 13
    if (nact == 3 && c->h.cfg.solver_type == LG_SOLVER_TGS && c->h.cfg.friction_model == LG_FRICTION_PYRAMID && c->spec && !caps && !selfc)
      LG_LAUNCH_PK(false, true, 1, 256);
    else
#endif
    if (nact == 3) {
      if (selfc) LG_LAUNCH_PK(false, true, 12, 256);
      else if (caps) LG_LAUNCH_PK(false, true, 4, 256);
      else LG_LAUNCH_PK(false, true, 0, 256);
    } else LG_LAUNCH_PK(false, false, FEAT_ALL, 64);
#undef LG_LAUNCH_PK
#endif
}

int lg_step(lg_ctx* c, const float* actions, void* stream) {
  if (!c) return LG_ERR_INVALID;
  DeviceScope ds_(c->device);
  if (!actions) { c->err = "actions is null"; return LG_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t* ev = nullptr;
  if (c->prof_max > 0) {
    if (c->prof_n < c->prof_max && (c->prof_calls % c->prof_stride) == 0) ev = &c->ev[(size_t)4 * c->prof_n++];
    c->prof_calls++;
  }
  if (ev) (void)hipEventRecord(ev[0], st);
  const bool fuse = can_fuse(c) || can_gfuse(c);
  launch_physics(c, st, actions, nullptr, c->h.N, NDOF, fuse ? 1 : 0);
  if (ev) (void)hipEventRecord(ev[1], st);
  if (fuse) {                                            // one launch per policy step
    if (ev) { (void)hipEventRecord(ev[2], st); (void)hipEventRecord(ev[3], st); }
    HIP_TRY(c, hipGetLastError());
    return LG_OK;
  }
  return launch_post(c, st, ev, nullptr, c->h.N, 0);
}

// lg_step whose post-physics kernel also fills one transition of a rollout storage (see PostSink): what
// RolloutStorage.add_transitions / PPO.process_env_step copy and compute after env.step in the reference's runner.
int lg_step_transition(lg_ctx* c, const float* actions, float* next_observations, const float* values, float gamma, float* rewards,
                       float* dones, void* stream) {
  if (!c) return LG_ERR_INVALID;
  DeviceScope ds_(c->device);
  if (!actions || !values || !rewards || !dones) { c->err = "lg_step_transition: null row"; return LG_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  const PostSink sink{next_observations, values, rewards, dones, gamma};
  if (can_fuse(c) || can_gfuse(c)) {
    launch_physics(c, st, actions, nullptr, c->h.N, NDOF, 1, sink);
    HIP_TRY(c, hipGetLastError());
    return LG_OK;
  }
  launch_physics(c, st, actions, nullptr, c->h.N);
  return launch_post(c, st, nullptr, nullptr, c->h.N, 0, nullptr, 0, sink);
}

int lg_step_subset(lg_ctx* c, const float* actions, const int32_t* env_ids, int32_t n, int32_t rollout_mode, void* stream) {
  if (!c) return LG_ERR_INVALID;
  DeviceScope ds_(c->device);
  if (!actions || !env_ids || n <= 0 || n > c->h.N) { c->err = "bad subset step arguments"; return LG_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  if (rollout_mode && can_fuse(c)) {                     // rollout steps end inside the physics kernel as well (LG_FUSE=0: two launches)
    launch_physics(c, st, actions, env_ids, n, NDOF, 2);
    HIP_TRY(c, hipGetLastError());
    return LG_OK;
  }
  launch_physics(c, st, actions, env_ids, n);
  return launch_post(c, st, nullptr, env_ids, n, rollout_mode ? 1 : 0);
}

int lg_set_reward_terms(lg_ctx* c, int32_t num_terms, const int32_t* term_ids, const float* scales, void* stream) {
  if (!c) return LG_ERR_INVALID;
  DeviceScope ds_(c->device);
  if (num_terms < 0 || num_terms > LG_MAX_REWARD_TERMS || (num_terms > 0 && (!term_ids || !scales))) { c->err = "bad reward term list"; return LG_ERR_INVALID; }
  for (int k = 0; k < num_terms; ++k) if (term_ids[k] < 0 || term_ids[k] >= LG_REW_COUNT) { c->err = "unknown reward term id"; return LG_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  lg_config& g = c->h.cfg;
  g.num_reward_terms = num_terms; c->h.K = num_terms;
  for (int k = 0; k < LG_MAX_REWARD_TERMS; ++k) { g.reward_term_ids[k] = k < num_terms ? term_ids[k] : 0; g.reward_scales[k] = k < num_terms ? scales[k] : 0.f; }
  // the three fields are adjacent in lg_config: one stream-ordered copy of that span of the device context
  char* base = (char*)&c->h; char* lo = (char*)&g.num_reward_terms; char* hi = (char*)&g.reward_scales[LG_MAX_REWARD_TERMS];
  HIP_TRY(c, hipMemcpyAsync((char*)c->d + (lo - base), lo, (size_t)(hi - lo), hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipMemcpyAsync((char*)c->d + ((char*)&c->h.K - base), &c->h.K, sizeof(int), hipMemcpyHostToDevice, st));
  reward_meta(c->h);
  hot_config(c->h);
  HIP_TRY(c, hipMemcpyAsync((char*)c->d + ((char*)&c->h.rew_term_mask - base), &c->h.rew_term_mask, 4 * sizeof(int) + sizeof(c->h.hot), hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipMemsetAsync(c->h.ep_sums, 0, (size_t)LG_MAX_REWARD_TERMS * c->h.N * sizeof(float), st));
  return LG_OK;
}

int lg_set_async_gait(lg_ctx* c, const float weights[3], float foot_z_align, void* stream) {
  if (!c || !weights) return LG_ERR_INVALID;
  DeviceScope ds_(c->device);
  lg_config& g = c->h.cfg;
  for (int i = 0; i < 3; ++i) g.async_weights[i] = weights[i];
  g.async_foot_z_align = foot_z_align;
  char* base = (char*)&c->h; char* lo = (char*)&g.async_weights[0]; char* hi = (char*)(&g.async_foot_z_align + 1);
  HIP_TRY(c, hipMemcpyAsync((char*)c->d + (lo - base), lo, (size_t)(hi - lo), hipMemcpyHostToDevice, (hipStream_t)stream));
  return LG_OK;
}

int lg_step_subset_physics(lg_ctx* c, const float* actions, const int32_t* env_ids, int32_t n, void* stream) {
  if (!c) return LG_ERR_INVALID;
  DeviceScope ds_(c->device);
  if (!actions || !env_ids || n <= 0 || n > c->h.N) { c->err = "bad subset step arguments"; return LG_ERR_INVALID; }
  launch_physics(c, (hipStream_t)stream, actions, env_ids, n);
  HIP_TRY(c, hipGetLastError());
  return LG_OK;
}

int lg_post_physics_subset(lg_ctx* c, const int32_t* env_ids, int32_t n, int32_t rollout_mode, void* stream) {
  if (!c) return LG_ERR_INVALID;
  DeviceScope ds_(c->device);
  if (!env_ids || n <= 0 || n > c->h.N) { c->err = "bad subset step arguments"; return LG_ERR_INVALID; }
  return launch_post(c, (hipStream_t)stream, nullptr, env_ids, n, rollout_mode ? 1 : 0);
}

int lg_step_physics(lg_ctx* c, const float* actions, void* stream) {
  if (!c) return LG_ERR_INVALID;
  DeviceScope ds_(c->device);
  if (!actions) { c->err = "actions is null"; return LG_ERR_INVALID; }
  launch_physics(c, (hipStream_t)stream, actions, nullptr, c->h.N);
  HIP_TRY(c, hipGetLastError());
  return LG_OK;
}

// RobotBatchRollout._sync_main_to_rollout (robot_batch_rollout.py:1447-1535): env i*(1+R) is main i, the next R envs are
// its rollouts; one lane per (rollout env, float) copies the 12 state tensors the reference copies.
static __global__ __launch_bounds__(256) void sync_kernel(const DevCtx* __restrict__ C, int R, float drift, uint32_t seed_lo, uint32_t call) {
  const int B = C->B;
  const bool net = C->cfg.control_type == LG_CTRL_ACTUATOR_NET;
  const int per = 13 + 2 * NDOF + NDOF + NDOF + NDOF + 6 + 3 + 3 + 3 + NLEG + NLEG + 1 + B * 16 + (net ? NDOF + 4 * (NDOF * 8) : 0);   // floats (+ one slot for the contact bytes)
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)C->N * per;
  if (gid >= total) return;
  const int e = (int)(gid / per), k = (int)(gid - (int64_t)e * per);
  const int src = e - e % (1 + R);
  if (src == e) return;
  int o = k;
  if (o < 13) {
    float v = C->root[(size_t)src * 13 + o];
    if (drift > 0.f && o < 3) {   // domain_rand.rollout_envs_sync_pos_drift (:1493-1497)
      uint32_t r4[4]; philox4((uint32_t)e, call, (uint32_t)o, 7u, seed_lo, 0x5a5au, r4);
      v += (u01(r4[0]) - 0.5f) * drift;
    }
    C->root[(size_t)e * 13 + o] = v; return;
  }
  o -= 13; if (o < 2 * NDOF) { C->dof[(size_t)e * (2 * NDOF) + o] = C->dof[(size_t)src * (2 * NDOF) + o]; return; }
  o -= 2 * NDOF; if (o < NDOF) { C->actions[(size_t)e * NDOF + o] = C->actions[(size_t)src * NDOF + o]; return; }
  o -= NDOF; if (o < NDOF) { C->last_actions[(size_t)e * NDOF + o] = C->last_actions[(size_t)src * NDOF + o]; return; }
  o -= NDOF; if (o < NDOF) { C->last_dof_vel[(size_t)e * NDOF + o] = C->last_dof_vel[(size_t)src * NDOF + o]; return; }
  o -= NDOF; if (o < 6) { C->last_root_vel[(size_t)e * 6 + o] = C->last_root_vel[(size_t)src * 6 + o]; return; }
  o -= 6; if (o < 3) { C->base_lin_vel[(size_t)e * 3 + o] = C->base_lin_vel[(size_t)src * 3 + o]; return; }
  o -= 3; if (o < 3) { C->base_ang_vel[(size_t)e * 3 + o] = C->base_ang_vel[(size_t)src * 3 + o]; return; }
  o -= 3; if (o < 3) { C->proj_grav[(size_t)e * 3 + o] = C->proj_grav[(size_t)src * 3 + o]; return; }
  o -= 3; if (o < NLEG) { C->feet_air[(size_t)e * NLEG + o] = C->feet_air[(size_t)src * NLEG + o]; return; }
  o -= NLEG; if (o < NLEG) { C->feet_ctime[(size_t)e * NLEG + o] = C->feet_ctime[(size_t)src * NLEG + o]; return; }
  o -= NLEG; if (o < 1) {
    for (int f = 0; f < NLEG; ++f) C->last_contacts[(size_t)e * NLEG + f] = C->last_contacts[(size_t)src * NLEG + f];
    // termination flags: the reference's rollouts are stepped with their main (:554-594) and so raise the same contact
    // termination; their stale flags then feed `_reward_termination` of the rollout steps (:806-809)
    C->reset_buf[e] = C->reset_buf[src]; C->time_out[e] = C->time_out[src];
    return;
  }
  // body states and contact forces: what the rollouts would hold had they been stepped along with their main (the
  // reference steps every env with the main's action, :554-594); the position drift moves all bodies alike
  o -= 1; if (o < B * 13) {
    float v = C->rigid[(size_t)src * B * 13 + o];
    const int comp = o % 13;
    if (drift > 0.f && comp < 3) {
      uint32_t r4[4]; philox4((uint32_t)e, call, (uint32_t)comp, 7u, seed_lo, 0x5a5au, r4);
      v += (u01(r4[0]) - 0.5f) * drift;
    }
    C->rigid[(size_t)e * B * 13 + o] = v; return;
  }
  o -= B * 13; if (o < B * 3) { C->cforce[(size_t)e * B * 3 + o] = C->cforce[(size_t)src * B * 3 + o]; return; }
  // actuator network (control.use_actuator_network in a batch-rollout task): the reference steps the rollouts along with their
  // main, with the main's action (anymal_c_batch_rollout.py:157-182), so their LSTM state and torques equal the main's at
  // every sync; here the mains are stepped alone and the state is copied
  o -= B * 3; if (o < NDOF) { C->torques[(size_t)e * NDOF + o] = C->torques[(size_t)src * NDOF + o]; return; }
  o -= NDOF;
  {
    const size_t N12 = (size_t)C->N * NDOF;
    constexpr int PL = NDOF * 8;
    const int which = o / (2 * PL), r = o - 2 * PL * which, lay = r / PL, i = r - PL * lay;      // [h | c] x [layer 0 | layer 1] x (dof x 8)
    float* T = which == 0 ? C->sea_h : C->sea_c;
    T[(lay * N12 + (size_t)e * NDOF) * 8 + i] = T[(lay * N12 + (size_t)src * NDOF) * 8 + i];
  }
}

int lg_sync_main_to_rollout(lg_ctx* c, int32_t rollouts_per_main, float pos_drift, void* stream) {
  if (!c) return LG_ERR_INVALID;
  DeviceScope ds_(c->device);
  if (rollouts_per_main <= 0 || c->h.N % (1 + rollouts_per_main) != 0) { c->err = "num_envs is not num_main * (1 + rollouts_per_main)"; return LG_ERR_INVALID; }
  const int per = 13 + 2 * NDOF + NDOF + NDOF + NDOF + 6 + 3 + 3 + 3 + NLEG + NLEG + 1 + c->h.B * 16 + (c->h.cfg.control_type == LG_CTRL_ACTUATOR_NET ? NDOF + 4 * (NDOF * 8) : 0);
  int64_t total = (int64_t)c->h.N * per;
  hipLaunchKernelGGL(sync_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, c->d, rollouts_per_main, pos_drift,
                     (uint32_t)c->h.cfg.seed, (uint32_t)(c->sync_calls++));
  HIP_TRY(c, hipGetLastError());
  return LG_OK;
}

// rollout_batch (robot_traj_grad_sampling.py:249-280): sync, `horizon` rollout steps, sync -- enqueued by one call.  Step i
// reads its action rows in place from the (n, horizon, 12) plan and the post kernel writes column i of `rewards`.
int lg_rollout_batch(lg_ctx* c, const float* all_us, int32_t horizon, const int32_t* env_ids, int32_t n, int32_t rollouts_per_main,
                     float pos_drift, float* rewards, void* stream) {
  if (!c) return LG_ERR_INVALID;
  DeviceScope ds_(c->device);
  if (!all_us || !env_ids || !rewards || horizon <= 0 || n <= 0 || n > c->h.N) { c->err = "bad rollout_batch arguments"; return LG_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  int rc = lg_sync_main_to_rollout(c, rollouts_per_main, pos_drift, stream);
  if (rc != LG_OK) return rc;
  const bool fuse = can_fuse(c);
  // One launch for the whole horizon (round 5): the workgroup that owns an env keeps its state on chip from step to step.  Not with the clock-driven gait
  // term (lg_config.gait_enabled: the two forms are not compared with it; the env class goes step by step anyway), not on mesh terrains (no persistent
  // instance), LG_PERSIST=0 = the A/B switch.
  if (fuse && horizon > 1 && c->persist && !c->h.cfg.gait_enabled && c->h.ter.mesh_type != LG_MESH_TRIMESH) {
    PostSink sk{nullptr, nullptr, nullptr, nullptr, 0.f, rewards, horizon};
    sk.nsteps = horizon;
    launch_physics(c, st, all_us, env_ids, n, horizon * NDOF, 3, sk);
    HIP_TRY(c, hipGetLastError());
    return lg_sync_main_to_rollout(c, rollouts_per_main, pos_drift, stream);
  }
  for (int i = 0; i < horizon; ++i) {
    if (fuse) {                                          // one launch per rollout step: the reward column is written by the kernel's tail
      launch_physics(c, st, all_us + (size_t)i * NDOF, env_ids, n, horizon * NDOF, 2, PostSink{nullptr, nullptr, nullptr, nullptr, 0.f, rewards + i, horizon});
      continue;
    }
    launch_physics(c, st, all_us + (size_t)i * NDOF, env_ids, n, horizon * NDOF);
    rc = launch_post(c, st, nullptr, env_ids, n, 1, rewards + i, horizon);
    if (rc != LG_OK) return rc;
  }
  HIP_TRY(c, hipGetLastError());
  return lg_sync_main_to_rollout(c, rollouts_per_main, pos_drift, stream);
}

int lg_gather_step_rows(lg_ctx* c, const int32_t* env_ids, int32_t n, float* obs_out, float* rew_out, uint8_t* reset_out, uint8_t* time_out_out, void* stream);
int lg_step_subset_rows(lg_ctx* c, const float* actions, const int32_t* env_ids, int32_t n, int32_t rollout_mode, float* obs_out, float* rew_out, uint8_t* reset_out,
                        uint8_t* time_out_out, void* stream) {
  if (!c) return LG_ERR_INVALID;
  DeviceScope ds_(c->device);
  if (!actions || !env_ids || n <= 0 || n > c->h.N) { c->err = "bad subset step arguments"; return LG_ERR_INVALID; }
  if (!obs_out || !rew_out || !reset_out || !time_out_out) { c->err = "lg_step_subset_rows: null output row"; return LG_ERR_INVALID; }
  if (rollout_mode && can_fuse(c)) {                     // the rows leave the tail of the one launch of a rollout step
    launch_physics(c, (hipStream_t)stream, actions, env_ids, n, NDOF, 2, PostSink{obs_out, nullptr, nullptr, nullptr, 0.f, rew_out, 1, reset_out, time_out_out, 1});
    HIP_TRY(c, hipGetLastError());
    return LG_OK;
  }
  const int rc = lg_step_subset(c, actions, env_ids, n, rollout_mode, stream);
  return rc != LG_OK ? rc : lg_gather_step_rows(c, env_ids, n, obs_out, rew_out, reset_out, time_out_out, stream);
}

int lg_gather_step_rows(lg_ctx* c, const int32_t* env_ids, int32_t n, float* obs_out, float* rew_out, uint8_t* reset_out, uint8_t* time_out_out, void* stream) {
  if (!c) return LG_ERR_INVALID;
  DeviceScope ds_(c->device);
  if (n < 0 || (n > 0 && !env_ids)) { c->err = "lg_gather_step_rows: bad id list"; return LG_ERR_INVALID; }
  if (n == 0) return LG_OK;
  const int64_t total = (int64_t)n * c->h.cfg.num_obs;
  hipLaunchKernelGGL(gather_step_rows_kernel, dim3((unsigned)((std::max<int64_t>(total, n) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, c->d, env_ids, n, obs_out, rew_out,
                     reset_out, time_out_out);
  HIP_TRY(c, hipGetLastError());
  return LG_OK;
}

int lg_set_state_indexed(lg_ctx* c, const float* root_states, const float* dof_state, const int32_t* env_ids, int32_t n, void* stream) {
  if (!c) return LG_ERR_INVALID;
  DeviceScope ds_(c->device);
  if (n < 0 || (n > 0 && !env_ids)) { c->err = "lg_set_state_indexed: bad id list"; return LG_ERR_INVALID; }
  if (n == 0) return LG_OK;
  hipLaunchKernelGGL(set_state_kernel, dim3((unsigned)((n + EPW - 1) / EPW)), dim3(64), 0, (hipStream_t)stream, c->d, root_states, dof_state, env_ids, n);
  HIP_TRY(c, hipGetLastError());
  return LG_OK;
}

int lg_set_extra_obs(lg_ctx* c, const float* dptr) {
  if (!c) return LG_ERR_INVALID;
  if (c->h.cfg.num_extra_obs > 0 && !dptr) { c->err = "extra obs buffer is null"; return LG_ERR_INVALID; }
  c->h.extra_obs = (const float LG_G*)dptr;
  HIP_TRY(c, hipMemcpy(c->d, &c->h, sizeof(DevCtx), hipMemcpyHostToDevice));
  return LG_OK;
}

int lg_set_extra_termination(lg_ctx* c, const uint8_t* dptr) {
  if (!c) return LG_ERR_INVALID;
  c->h.extra_term = (const uint8_t LG_G*)dptr;
  HIP_TRY(c, hipMemcpy(c->d, &c->h, sizeof(DevCtx), hipMemcpyHostToDevice));
  return LG_OK;
}

int lg_profile_begin(lg_ctx* c, int32_t max_samples, int32_t stride) {
  if (!c || max_samples <= 0 || stride <= 0) return LG_ERR_INVALID;
  DeviceScope ds_(c->device);
  for (auto e : c->ev) (void)hipEventDestroy(e);
  c->ev.assign((size_t)4 * max_samples, nullptr);
  for (auto& e : c->ev) HIP_TRY(c, hipEventCreate(&e));
  c->prof_max = max_samples; c->prof_stride = stride; c->prof_n = 0; c->prof_calls = 0;
  return LG_OK;
}

int lg_profile_end(lg_ctx* c, float mean_ms[3], int32_t* nsamples) {
  if (!c || !mean_ms || !nsamples) return LG_ERR_INVALID;
  DeviceScope ds_(c->device);
  double acc[3] = {0, 0, 0};
  for (int i = 0; i < c->prof_n; ++i) {
    hipEvent_t* ev = &c->ev[(size_t)4 * i];
    HIP_TRY(c, hipEventSynchronize(ev[3]));
    for (int k = 0; k < 3; ++k) { float ms = 0.f; HIP_TRY(c, hipEventElapsedTime(&ms, ev[k], ev[k + 1])); acc[k] += ms; }
  }
  *nsamples = c->prof_n;
  for (int k = 0; k < 3; ++k) mean_ms[k] = c->prof_n > 0 ? (float)(acc[k] / c->prof_n) : 0.f;
  for (auto e : c->ev) (void)hipEventDestroy(e);
  c->ev.clear(); c->prof_max = 0; c->prof_n = 0;
  return LG_OK;
}

int lg_compute_torques(lg_ctx* c, const float* actions, void* stream) {
  if (!c) return LG_ERR_INVALID;
  DeviceScope ds_(c->device);
  const int nb = (c->h.N + EPB - 1) / EPB;
#if NJ != 3
  hipLaunchKernelGGL((physics_kernel_chain<2, false>), dim3(nb), dim3(64), 0, (hipStream_t)stream, c->d, actions, 0, (const int32_t*)nullptr, c->h.N, NDOF, EPW);
#else
  hipLaunchKernelGGL((physics_kernel<2, false>), dim3(nb), dim3(64), 0, (hipStream_t)stream, c->d, actions, 0, 0, (const int32_t*)nullptr, c->h.N, NDOF, 0, PostSink{nullptr, nullptr, nullptr, nullptr, 0.f});
#endif
  HIP_TRY(c, hipGetLastError());
  return LG_OK;
}

int lg_simulate(lg_ctx* c, void* stream) {
  if (!c) return LG_ERR_INVALID;
  DeviceScope ds_(c->device);
  const int nb = (c->h.N + EPB - 1) / EPB;
#if NJ != 3
  if (c->h.ter.mesh_type == LG_MESH_TRIMESH) hipLaunchKernelGGL((physics_kernel_chain<1, true>), dim3(nb), dim3(64), 0, (hipStream_t)stream, c->d, (const float*)nullptr, 1, (const int32_t*)nullptr, c->h.N, NDOF, EPW);
  else hipLaunchKernelGGL((physics_kernel_chain<1, false>), dim3(nb), dim3(64), 0, (hipStream_t)stream, c->d, (const float*)nullptr, 1, (const int32_t*)nullptr, c->h.N, NDOF, EPW);
#else
  if (c->h.ter.mesh_type == LG_MESH_TRIMESH)
    hipLaunchKernelGGL((physics_kernel<1, true, false, FEAT_ALL>), dim3(nb), dim3(64), 0, (hipStream_t)stream, c->d, (const float*)nullptr, 1, 0, (const int32_t*)nullptr, c->h.N, NDOF, 0, PostSink{nullptr, nullptr, nullptr, nullptr, 0.f});
  else
    hipLaunchKernelGGL((physics_kernel<1, false, false, FEAT_ALL>), dim3(nb), dim3(64), 0, (hipStream_t)stream, c->d, (const float*)nullptr, 1, 0, (const int32_t*)nullptr, c->h.N, NDOF, 0, PostSink{nullptr, nullptr, nullptr, nullptr, 0.f});
#endif
  HIP_TRY(c, hipGetLastError());
  return LG_OK;
}

int lg_post_physics_step(lg_ctx* c, void* stream) {
  if (!c) return LG_ERR_INVALID;
  DeviceScope ds_(c->device);
  return launch_post(c, (hipStream_t)stream, nullptr, nullptr, c->h.N, 0);
}

int lg_reset_idx(lg_ctx* c, const int32_t* env_ids, int32_t n, int32_t update_curriculum, void* stream) {
  if (!c) return LG_ERR_INVALID;
  DeviceScope ds_(c->device);
  if (n < 0 || (n > 0 && !env_ids)) { c->err = "bad env id list"; return LG_ERR_INVALID; }
  if (n == 0) return LG_OK;                         // LR:172-173
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(reset_idx_kernel, dim3(1), dim3(256), 0, st, c->d, env_ids, n, update_curriculum);
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, st, c->d, 1, 0, 0);
  HIP_TRY(c, hipGetLastError());
  return LG_OK;
}

}  // extern "C"
}  // namespace LG_NS
