// lg_physics.h — articulated-body dynamics + contact for one environment, spread over a lane GROUP: a DPP quad (4 lanes) for the
// four-legged robots, eight lanes (half a DPP row, two of them idle) for the six-legged one.  Lane l of the group owns leg l
// (3 revolute joints, its links, its collision spheres); the floating base is replicated on every lane of the group and every
// cross-leg coupling goes through the 6x6 base Schur complement, reduced with DPP adds (grp_sum).
// The group width is a compile-time constant of the translation unit (LG_LEGS -> GRP, lg_step.hip compiles one kernel instance per
// supported leg count); this header is included inside that instance's namespace.
//
//   M = [ Mbb  Mb1 Mb2 Mb3 Mb4 ]      legs couple only through the base, so with  Y_k = Mkk^-1 Mbk^T
//       [ Mb1' M11             ]        S   = Mbb - sum_k Mbk Y_k                      (6x6, replicated)
//       [ ...        ...       ]        a_b = S^-1 (f_b - sum_k Mbk Mkk^-1 f_k)
//       [ Mb4'             M44 ]        a_k = Mkk^-1 f_k - Y_k a_b
//
// Replaces gym.simulate (reference legged_robot.py:100; PhysX, closed): same tensor contract (:564-584), own model:
// composite-rigid-body mass matrix + recursive Newton-Euler bias (spatial quantities about the base origin P, world
// axes), sphere-vs-heightfield contacts solved at velocity level (every leg walks the list of its active contacts; the four
// legs' j-th entries relax simultaneously) by the solver sim.physx.solver_type names (legged_robot_config.py:256-267):
//   LG_SOLVER_TGS (the reference's setting): temporal Gauss-Seidel -- `iters` sub-intervals of h = dt / iters, each ONE pass over
//     the rows with the separation re-evaluated from the displacement accumulated so far (sep = gap + J_n . dq, bias over h),
//     then dq += h v; the pose advances once, by dq, at the end of the step;
//   LG_SOLVER_PGS: `iters` sweeps at the frozen pose (bias over dt), then dq = dt v.
// Friction rows: PhysX's two scalar rows along the tangents, each clamped to +-mu f_n (LG_FRICTION_PYRAMID), or the exact 2x2
// tangential block projected on the disc (LG_FRICTION_CONE).  Semi-implicit Euler.
#pragma once
#ifndef LG_AB
#define LG_AB 0
#endif
#ifndef LG_PGS_REG
#define LG_PGS_REG 1     // slot records a lane keeps in registers through the Gauss-Seidel sweeps (0: all from LDS)
#endif
#include "lg_device.h"
#include "lg_bvh.h"
#ifndef LG_LEGS
#define LG_LEGS 4
#endif
#if LG_LEGS == 4
#define GRP 4            // lanes per env
#elif LG_LEGS == 6
#define GRP 8
#elif LG_LEGS == 2
#define GRP 2
#else
#error "kernel instances exist for 4, 6 and 2 legs"
#endif
#ifndef LG_JOINTS
#define LG_JOINTS 3
#endif
#define NLEG LG_LEGS
#define NJ LG_JOINTS     // joints per leg: 3 (the tuned kernels below), 6 (lg_chain.h: the two-legged instance)
#define NDOF (NJ * NLEG)
static_assert(NJ <= LG_MAX_JOINTS_PER_LEG && NDOF <= LG_MAX_DOF, "joint count of this instance");
#define EPW (64 / GRP)   // envs per wave
static_assert(NLEG <= LG_MAX_LEGS && NLEG <= GRP, "leg count of this instance");

// ---- all-reduce over the lanes of an env's group.  Quad: two v_add_f32_dpp quad_perm.  Eight lanes: lanes NLEG..7 carry no leg -- their
// model is a copy of leg 0's (finite arithmetic everywhere) and they are selected out of every sum here --, then row_half_mirror + the two
// quad permutes: three DPP adds, no LDS.
LG_DEV int lane_in_group() { return (int)(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) & (GRP - 1)); }
#if GRP == 4
LG_DEV float grp_sum(float x) { return quad_sum(x); }
#elif GRP == 2
LG_DEV float grp_sum(float x) { return x + dpp_xor1(x); }
#else
LG_DEV float dpp_half_mirror(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true)); }
LG_DEV float grp_sum(float x) { x = lane_in_group() < NLEG ? x : 0.f; x += dpp_half_mirror(x); x += dpp_xor1(x); x += dpp_xor2(x); return x; }
#endif
LG_DEV V3 grp_sum(V3 a) { return v3(grp_sum(a.x), grp_sum(a.y), grp_sum(a.z)); }
// minimum over ALL lanes of the group (the self-collision pass deals its pair tests to every lane, the idle ones of the six-legged instance included)
LG_DEV int dpp_xor1_i(int x) { return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true); }
LG_DEV int dpp_xor2_i(int x) { return __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true); }
#if GRP == 4
LG_DEV float grp_min_all(float x) { x = fminf(x, dpp_xor1(x)); x = fminf(x, dpp_xor2(x)); return x; }
LG_DEV int grp_min_all(int x) { x = min(x, dpp_xor1_i(x)); x = min(x, dpp_xor2_i(x)); return x; }
#elif GRP == 2
LG_DEV float grp_min_all(float x) { return fminf(x, dpp_xor1(x)); }
LG_DEV int grp_min_all(int x) { return min(x, dpp_xor1_i(x)); }
#else
LG_DEV int dpp_half_mirror_i(int x) { return __builtin_amdgcn_mov_dpp(x, 0x141, 0xF, 0xF, true); }
LG_DEV float grp_min_all(float x) { x = fminf(x, dpp_half_mirror(x)); x = fminf(x, dpp_xor1(x)); x = fminf(x, dpp_xor2(x)); return x; }
LG_DEV int grp_min_all(int x) { x = min(x, dpp_half_mirror_i(x)); x = min(x, dpp_xor1_i(x)); x = min(x, dpp_xor2_i(x)); return x; }
#endif
LG_DEV S3 grp_sum(S3 a) { return S3{grp_sum(a.xx), grp_sum(a.xy), grp_sum(a.xz), grp_sum(a.yy), grp_sum(a.yz), grp_sum(a.zz)}; }

// Per-leg model constants staged in LDS as [field][lane of the group]: lane l reads field*GRP + l, i.e. a wave touches GRP consecutive
// dwords per field (broadcast, conflict-free) instead of issuing ~100 dependent global loads per substep.
enum { LM_JPOS = 0, LM_JROT = LM_JPOS + 3 * NJ, LM_JAXIS = LM_JROT + 9 * NJ, LM_MASS = LM_JAXIS + 3 * NJ, LM_COM = LM_MASS + NJ, LM_INERTIA = LM_COM + 3 * NJ,
       LM_FOOT_POS = LM_INERTIA + 6 * NJ, LM_FOOT_ROT = LM_FOOT_POS + 3, LM_VEL_LIMIT = LM_FOOT_ROT + 9, LM_TORQUE_LIMIT = LM_VEL_LIMIT + NJ,
       LM_DEFAULT_POS = LM_TORQUE_LIMIT + NJ, LM_PGAIN = LM_DEFAULT_POS + NJ, LM_DGAIN = LM_PGAIN + NJ, LM_CP_COUNT = LM_DGAIN + NJ,
       LM_CP_LINK = LM_CP_COUNT + 1, LM_CP_POS = LM_CP_LINK + LG_MAX_CP, LM_CP_RADIUS = LM_CP_POS + 3 * LG_MAX_CP, LM_LOWER = LM_CP_RADIUS + LG_MAX_CP, LM_UPPER = LM_LOWER + NJ,
       LM_SOFT_LO = LM_UPPER + NJ, LM_SOFT_HI = LM_SOFT_LO + NJ /* cfg.dof_pos_limits: the soft limits of _reward_dof_pos_limits */,
       LM_CP_SLIDE = LM_SOFT_HI + NJ /* 8 x 3: segment to the next sphere of a capsule's chain (link frame), lg_robot_model.cp_slide */, LM_FIELDS = LM_CP_SLIDE + 3 * LG_MAX_CP };
#if LG_JOINTS == 3
static_assert(LM_JROT == 9 && LM_MASS == 45 && LM_FOOT_POS == 75 && LM_CP_COUNT == 102 && LM_CP_POS == 111 && LM_LOWER == 143 && LM_CP_SLIDE == 155 && LM_FIELDS == 179, "the three-joint table is the one the tuned kernels were measured with");
#endif
struct LegModel {
  const float* t; int l;
  LG_DEV float f(int field) const { return t[field * GRP + l]; }
  LG_DEV V3 v(int field) const { return v3(f(field), f(field + 1), f(field + 2)); }
  LG_DEV int i(int field) const { return __float_as_int(f(field)); }
};
// cooperative fill by one wave (64 lanes); call before any LegModel read, followed by a barrier
// entry idx = field * 4 + leg of the per-leg model table (LM_*); evaluated once on the host (pack_leg_model, at lg_create): the kernels
// copy the packed table into LDS.  (Filled in the kernel it was ~25 divergent branches with a dependent load each, ten times over:
// 6-8 k cycles in front of the first barrier of every launch.)
__host__ __device__ inline float leg_model_entry(const lg_robot_model* m, const lg_config* g, int idx) {
  const int field = idx / GRP, lg_ = idx % GRP;
  const int l = lg_ < NLEG ? lg_ : 0;      // lanes without a leg (six legs on eight lanes): a copy of leg 0, no collision points, no joint limits
  if (lg_ >= NLEG && field == LM_CP_COUNT) return 0.f;
  if (lg_ >= NLEG && field >= LM_LOWER && field < LM_SOFT_LO) return 0.f;
  if (field < LM_JROT) return m->joint_pos[l][field / 3][field % 3];
  if (field < LM_JAXIS) { int k = field - LM_JROT; return m->joint_rot[l][k / 9][k % 9]; }
  if (field < LM_MASS) { int k = field - LM_JAXIS; return m->joint_axis[l][k / 3][k % 3]; }
  if (field < LM_COM) return m->link_mass[l][field - LM_MASS];
  if (field < LM_INERTIA) { int k = field - LM_COM; return m->link_com[l][k / 3][k % 3]; }
  if (field < LM_FOOT_POS) { int k = field - LM_INERTIA; return m->link_inertia[l][k / 6][k % 6]; }
  if (field < LM_FOOT_ROT) return m->foot_pos[l][field - LM_FOOT_POS];
  if (field < LM_VEL_LIMIT) return m->foot_rot[l][field - LM_FOOT_ROT];
  if (field < LM_TORQUE_LIMIT) return m->dof_vel_limit[NJ * l + field - LM_VEL_LIMIT];
  if (field < LM_DEFAULT_POS) return m->torque_limit[NJ * l + field - LM_TORQUE_LIMIT];
  if (field < LM_PGAIN) return g->default_dof_pos[NJ * l + field - LM_DEFAULT_POS];
  if (field < LM_DGAIN) return g->p_gains[NJ * l + field - LM_PGAIN];
  if (field < LM_CP_COUNT) return g->d_gains[NJ * l + field - LM_DGAIN];
  float val; int iv;
  if (field < LM_CP_LINK) { iv = m->cp_count[l]; memcpy(&val, &iv, 4); return val; }
  if (field < LM_CP_POS) { iv = m->cp_link[l][field - LM_CP_LINK]; memcpy(&val, &iv, 4); return val; }
  if (field < LM_CP_RADIUS) { int k = field - LM_CP_POS; return m->cp_pos[l][k / 3][k % 3]; }
  if (field < LM_LOWER) return m->cp_radius[l][field - LM_CP_RADIUS];
  if (field < LM_UPPER) return m->dof_lower[NJ * l + field - LM_LOWER];
  if (field < LM_SOFT_LO) return m->dof_upper[NJ * l + field - LM_UPPER];
  if (field < LM_SOFT_HI) return g->dof_pos_limits[NJ * l + field - LM_SOFT_LO][0];
  if (field < LM_CP_SLIDE) return g->dof_pos_limits[NJ * l + field - LM_SOFT_HI][1];
  { int k = field - LM_CP_SLIDE; return lg_ >= NLEG ? 0.f : m->cp_slide[l][k / 3][k % 3]; }
}
inline void pack_leg_model(float* t, const lg_robot_model* m, const lg_config* g) {
  for (int idx = 0; idx < LM_FIELDS * GRP; ++idx) t[idx] = leg_model_entry(m, g, idx);
}
// LDS copy of the packed table by `nthreads` threads (tid 0 .. nthreads-1); the caller holds the barrier
LG_DEV void fill_leg_model(float* t, const float* __restrict__ packed, int tid, int nthreads) {
  for (int idx = tid; idx < LM_FIELDS * GRP; idx += nthreads) t[idx] = packed[idx];
}

struct LegKin {
  M3 R[NJ];
  V3 O[NJ], ax[NJ], com[NJ], w[NJ], vO[NJ];
  S3 Ic[NJ];
};

LG_DEV void leg_kinematics(const LegModel& lm_, const M3& Rb, V3 pb, V3 vb, V3 wb,
                           const float q[NJ], const float qd[NJ], LegKin& k) {
  M3 Rp = Rb; V3 Op = pb, wp = wb, vp = vb;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    M3 fix;
#pragma unroll
    for (int i = 0; i < 9; ++i) fix.m[i] = lm_.f(LM_JROT + 9 * j + i);
    V3 a = lm_.v(LM_JAXIS + 3 * j);
    k.O[j] = Op + mul(Rp, lm_.v(LM_JPOS + 3 * j));
    M3 R0 = mul(Rp, fix);
    k.ax[j] = mul(R0, a);
    k.R[j] = mul(R0, axis_angle(a, q[j]));
    k.com[j] = k.O[j] + mul(k.R[j], lm_.v(LM_COM + 3 * j));
    k.vO[j] = vp + cross(wp, k.O[j] - Op);
    k.w[j] = wp + qd[j] * k.ax[j];
    float I6[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) I6[i] = lm_.f(LM_INERTIA + 6 * j + i);
    k.Ic[j] = rotate_inertia(k.R[j], I6);
    Rp = k.R[j]; Op = k.O[j]; wp = k.w[j]; vp = k.vO[j];
  }
}

#define LT(i, j) ((i) * ((i) + 1) / 2 + (j))   // packed lower-triangular index

LG_DEV void sym3_inverse(const float a[6], float o[6]) {  // a: 00 01 02 11 12 22
  float c00 = a[3] * a[5] - a[4] * a[4], c01 = a[2] * a[4] - a[1] * a[5], c02 = a[1] * a[4] - a[2] * a[3];
  float id = frcp(a[0] * c00 + a[1] * c01 + a[2] * c02);
  o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
  o[3] = (a[0] * a[5] - a[2] * a[2]) * id; o[4] = (a[1] * a[2] - a[0] * a[4]) * id; o[5] = (a[0] * a[3] - a[1] * a[1]) * id;
}
LG_DEV void sym3_mul(const float a[6], const float x[3], float y[3]) {
  y[0] = a[0] * x[0] + a[1] * x[1] + a[2] * x[2];
  y[1] = a[1] * x[0] + a[3] * x[1] + a[4] * x[2];
  y[2] = a[2] * x[0] + a[4] * x[1] + a[5] * x[2];
}
// in-place Cholesky of a packed-lower SPD 6x6; the diagonal is stored INVERTED
LG_DEV void chol6(float* A) {
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float d = A[LT(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= A[LT(j, k)] * A[LT(j, k)];
    float inv = __builtin_amdgcn_rsqf(fmaxf(d, 1e-20f));
    A[LT(j, j)] = inv;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      float s = A[LT(i, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= A[LT(i, k)] * A[LT(j, k)];
      A[LT(i, j)] = s * inv;
    }
  }
}
LG_DEV void solve6(const float* L, float* b) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float s = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= L[LT(i, k)] * b[k];
    b[i] = s * L[LT(i, i)];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    float s = b[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) s -= L[LT(k, i)] * b[k];
    b[i] = s * L[LT(i, i)];
  }
}

// explicit inverse of an SPD 6x6 from its Cholesky factor (packed lower, inverted diagonal): Linv, then Sinv = Linv^T Linv.
// Applying S^-1 as a dense symmetric mat-vec has no dependent chain, unlike forward/back substitution.
LG_DEV void spd6_inverse_from_chol(const float* L, float* Si) {
  float Li[21];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    Li[LT(j, j)] = L[LT(j, j)];
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      float sacc = 0.f;
#pragma unroll
      for (int k = j; k < i; ++k) sacc += L[LT(i, k)] * Li[LT(k, j)];
      Li[LT(i, j)] = -sacc * L[LT(i, i)];
    }
  }
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      float sacc = 0.f;
#pragma unroll
      for (int k = a; k < 6; ++k) sacc += Li[LT(k, a)] * Li[LT(k, b)];
      Si[LT(a, b)] = sacc;
    }
}
// Inverse of an SPD 6 x 6 (packed lower) by 3 x 3 blocks: S = [A B^T; B D] (A: rows 0..2, D: rows 3..5, B = rows 3..5 x cols 0..2),
//   A^-1 and the Schur complement's inverse C^-1 = (D - B A^-1 B^T)^-1 in closed form (cofactors, one reciprocal each),
//   S^-1 = [A^-1 + E^T C^-1 E, -E^T C^-1; -C^-1 E, C^-1] with E = B A^-1.
// Against the Cholesky route (six dependent rsqrt / row chains, then a triangular inverse) this is two short chains of independent products: what a wave
// that is alone on its SIMD is short of is not arithmetic but dependent latency.
LG_DEV void spd6_inverse_blocks(const float* S, float* Si) {
  const float A[6] = {S[LT(0, 0)], S[LT(1, 0)], S[LT(2, 0)], S[LT(1, 1)], S[LT(2, 1)], S[LT(2, 2)]};      // 00 01 02 11 12 22
  const float D[6] = {S[LT(3, 3)], S[LT(4, 3)], S[LT(5, 3)], S[LT(4, 4)], S[LT(5, 4)], S[LT(5, 5)]};
  float B[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) B[i][j] = S[LT(3 + i, j)];
  float Ai[6]; sym3_inverse(A, Ai);
  float E[3][3];                      // E = B A^-1
#pragma unroll
  for (int i = 0; i < 3; ++i) { const float bi[3] = {B[i][0], B[i][1], B[i][2]}; sym3_mul(Ai, bi, E[i]); }
  float C[6];                         // D - E B^T (symmetric)
  C[0] = D[0] - (E[0][0] * B[0][0] + E[0][1] * B[0][1] + E[0][2] * B[0][2]);
  C[1] = D[1] - (E[0][0] * B[1][0] + E[0][1] * B[1][1] + E[0][2] * B[1][2]);
  C[2] = D[2] - (E[0][0] * B[2][0] + E[0][1] * B[2][1] + E[0][2] * B[2][2]);
  C[3] = D[3] - (E[1][0] * B[1][0] + E[1][1] * B[1][1] + E[1][2] * B[1][2]);
  C[4] = D[4] - (E[1][0] * B[2][0] + E[1][1] * B[2][1] + E[1][2] * B[2][2]);
  C[5] = D[5] - (E[2][0] * B[2][0] + E[2][1] * B[2][1] + E[2][2] * B[2][2]);
  float Ci[6]; sym3_inverse(C, Ci);
  float G[3][3];                      // G = C^-1 E  (rows 3..5 x cols 0..2 of -S^-1)
#pragma unroll
  for (int j = 0; j < 3; ++j) { const float ej[3] = {E[0][j], E[1][j], E[2][j]}; float g[3]; sym3_mul(Ci, ej, g); G[0][j] = g[0]; G[1][j] = g[1]; G[2][j] = g[2]; }
  // lower-right block
  Si[LT(3, 3)] = Ci[0]; Si[LT(4, 3)] = Ci[1]; Si[LT(5, 3)] = Ci[2]; Si[LT(4, 4)] = Ci[3]; Si[LT(5, 4)] = Ci[4]; Si[LT(5, 5)] = Ci[5];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Si[LT(3 + i, j)] = -G[i][j];
  // upper-left block: A^-1 + E^T G
  const int ai[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) Si[LT(a, b)] = Ai[ai[a][b]] + E[0][a] * G[0][b] + E[1][a] * G[1][b] + E[2][a] * G[2][b];
}
LG_DEV void symv6(const float* Si, const float* x, float* y) {
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    float sacc = 0.f;
#pragma unroll
    for (int b = 0; b < 6; ++b) sacc += (b <= a ? Si[LT(a, b)] : Si[LT(b, a)]) * x[b];
    y[a] = sacc;
  }
}

// terrain surface under (x, y): height and unit normal of the regular-grid triangulation (diagonal v(i,j)->v(i+1,j+1)).
// Split in two so that callers can put other work between the four sample loads and their first use.
struct TerrainView { int mesh_type, rows, cols; float hscale, vscale, border; const int16_t LG_G* __restrict__ H;
                     const int16_t LG_G* __restrict__ Hmin;   /* per cell: min(H[i][j], H[i+1][j], H[i][j+1]) -- what the height scan takes (LR:929-936); built by lg_create */
                     MeshView M;
                     LatticeView L;   /* cell != null: the mesh's triangles by lattice cell (closest_point_lattice) */
                     const float* __restrict__ GV; /* non-null: grid mesh */ const float4* __restrict__ GV4; /* its vertices (rows x cols) x (x, y, z, 0), world frame */
                     const float* __restrict__ GM; int mcols; /* max vertex z per 2 x 2 block of vertices */
                     const float4* __restrict__ SEG4; /* non-null: the grid mesh's vertices as above, for the capsule segments' edge pieces (contact_detect_mesh<true>; set whichever way the sphere queries go) */ };
struct TerrainCell { float u, v; int16_t h0, h1, h2, h3; };
LG_DEV TerrainCell terrain_fetch(const TerrainView& T, float x, float y) {
  // branch-free on purpose (the plane reads its 1 x 1 dummy grid): a conditional around the loads would make the
  // compiler wait for them at the end of the branch, and the point of the split is to leave them in flight
  TerrainCell c;
  const float ihs = T.hscale > 0.f ? frcp(T.hscale) : 0.f;
  float fx = (x + T.border) * ihs, fy = (y + T.border) * ihs;
  int i = (int)floorf(fx), j = (int)floorf(fy);
  i = max(0, min(i, T.rows - 2)); j = max(0, min(j, T.cols - 2));
  c.u = fminf(fmaxf(fx - (float)i, 0.f), 1.f); c.v = fminf(fmaxf(fy - (float)j, 0.f), 1.f);
  const int dc = T.cols > 1 ? 1 : 0, dr = T.rows > 1 ? T.cols : 0;
  const int16_t* r0 = T.H + (size_t)i * T.cols + j;
  c.h0 = r0[0]; c.h1 = r0[dc]; c.h2 = r0[dr]; c.h3 = r0[dr + dc];
  return c;
}
LG_DEV void terrain_eval(const TerrainView& T, const TerrainCell& c, float* h, V3* n) {
  if (T.mesh_type == LG_MESH_PLANE) { *h = 0.f; *n = v3(0, 0, 1); return; }
  const float ihs = frcp(T.hscale);
  float h0 = T.vscale * c.h0, h1 = T.vscale * c.h1, h2 = T.vscale * c.h2, h3 = T.vscale * c.h3;
  float dhdu, dhdv;
  if (c.v >= c.u) { dhdu = h3 - h1; dhdv = h1 - h0; } else { dhdu = h2 - h0; dhdv = h3 - h2; }
  *h = h0 + c.u * dhdu + c.v * dhdv;
  V3 g = v3(-dhdu * ihs, -dhdv * ihs, 1.f);
  *n = __builtin_amdgcn_rsqf(dot(g, g)) * g;
}
LG_DEV void terrain_query(const TerrainView& T, float x, float y, float* h, V3* n) {
  const TerrainCell c = terrain_fetch(T, x, y);
  terrain_eval(T, c, h, n);
}

#if NJ == 3      // ---- the tuned three-joint kernels' own piece (lg_chain.h holds the six-joint instance's)
// per-contact-slot scratch in LDS, laid out [slot][lane][field]: one lane's record of a slot is 60 contiguous floats
// (240 B, 16-B aligned), so the sweeps fetch it with 15 ds_read_b128 instead of 59 ds_read_b32 -- a lone wave on a SIMD
// gets a fifth of the LDS rate on 4-byte reads and the full rate on 16-byte ones (MI355X_MICROARCH.md, LDS).  Rows of
// 60 dwords put the 16 lanes of every ds_read_b128 lane group on 16 distinct 4-bank sets: conflict-free.
// fields 0..11 are written by the contact detection (the multipliers then by the sweeps; field 9 by the set-up), 12..59 by the
// set-up: each block is a whole number of 16-B units
// The sweeps run on packed fp32 (v_pk_fma_f32: two lanes of a 64-bit register pair per instruction), so everything they
// combine pairwise sits at an EVEN offset of the record, next to its partner: the two tangential multipliers, the
// (t1, t2) components interleaved, (An1, An2), the rows of the inverse tangential block, the base-response columns in
// pairs of base coordinates, the joint responses of joints (0, 1) of the three contact-frame axes.
// The blocks are laid out in the order a sweep step needs them (contact frame and gap, Jacobian pieces, tangents, the
// tangential block, then the responses), because the step starts computing as soon as the first 16-byte reads land.
enum { CF_N = 0, CF_GAP = 3 /* signed gap at the start of the step */, CF_R = 4, CF_ACTIVE = 7, CF_L0 = 8, CF_ANN = 9 /* 1 / (Ann + cfm), written by the set-up */, CF_L1 = 10, CF_L2 = 11,
       CF_SETUP = 12, CF_JK0 = 12, CF_JK1 = 15, CF_JK2 = 18, CF_ZC2 = 21 /* 3: (Mkk^-1 J_k^T)[c][joint 2] */,
       CF_T12 = 24 /* t1.x t2.x t1.y t2.y t1.z t2.z */, CF_AN12 = 30 /* An1 An2 */, CF_B = 32 /* cone: inverse tangential block, rows (B11 B12) (B12 B22); pyramid: 1/A11, A12, A12, 1/A22 */,
       CF_WB = 36 /* 3 x 6: base response per unit contact-frame impulse */, CF_ZCP = 54 /* 3 x 2: (Mkk^-1 J_k^T)[c][joint 0, 1] */, CF_FIELDS = 60 };
static_assert(CF_T12 % 4 == 0 && CF_L0 == 8 && CF_ANN == 9 && CF_L1 == 10 && CF_L2 == 11 && CF_GAP == 3 && CF_ACTIVE == 7, "16-byte units of the record as CS4 reads them");
static_assert(CF_T12 % 2 == 0 && CF_AN12 % 2 == 0 && CF_B % 2 == 0 && CF_WB % 2 == 0 && CF_ZCP % 2 == 0 && CF_L1 % 2 == 0, "packed operands sit at even offsets");
typedef float pk2 __attribute__((ext_vector_type(2)));
LG_DEV pk2 pk_splat(float x) { pk2 r = {x, x}; return r; }
LG_DEV pk2 pk_fma(pk2 a, pk2 b, pk2 c) { return __builtin_elementwise_fma(a, b, c); }
#define CS(slot, f) cst[((slot) * 64 + lane) * CF_FIELDS + (f)]
// 16-byte unit u of a lane's slot record: record fields are only ever touched four at a time -- a 4-byte access at a stride of 60 dwords
// across the lanes lands on 16 of the 64 banks (4-way conflict); a ds_*_b128 of the same record does not conflict (odd number of units)
#define CS4(slot, u) (reinterpret_cast<float4*>(cst + ((slot) * 64 + lane) * CF_FIELDS)[u])
// which lanes have an active contact in a slot: one 64-bit ballot per slot behind the records, written by the wave that detects the slot
#define LG_CST_FLOATS (LG_MAX_CP * CF_FIELDS * 64 + 2 * LG_MAX_CP)
#define AMASK(slot) (reinterpret_cast<unsigned long long*>(const_cast<float*>(cst) + LG_MAX_CP * CF_FIELDS * 64)[slot])
static_assert(CF_FIELDS % 4 == 0 && (CF_FIELDS / 4) % 2 == 1, "slot rows: 16-B aligned, odd number of 16-B units (bank spread)");
// the whole record of one slot, 15 x 16 B
LG_DEV void load_slot_record(const float* cst, int slot, int lane, float rec[CF_FIELDS]) {
  const float4* p = reinterpret_cast<const float4*>(&CS(slot, 0));
#pragma unroll
  for (int i = 0; i < CF_FIELDS / 4; ++i) { float4 v = p[i]; rec[4 * i] = v.x; rec[4 * i + 1] = v.y; rec[4 * i + 2] = v.z; rec[4 * i + 3] = v.w; }
}
LG_DEV V3 lds3(const float* cst, int slot, int f, int lane) { return v3(CS(slot, f), CS(slot, f + 1), CS(slot, f + 2)); }
LG_DEV void sts3(float* cst, int slot, int f, int lane, V3 a) { CS(slot, f) = a.x; CS(slot, f + 1) = a.y; CS(slot, f + 2) = a.z; }

#endif           // NJ == 3
// Diagnostic build only (-DLG_STAMPS): lane 0 of workgroup 0 accumulates shader-clock deltas per phase.
#ifdef LG_STAMPS
#define STAMP(i) do { unsigned long long _t = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); \
                      if (stamps) { stamps[i] += _t - stamp_t; } stamp_t = _t; } while (0)
#define STAMP_DECL unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
#else
#define STAMP(i)
#define STAMP_DECL
#endif

struct PhysParams {
  float dt; V3 grav; int iters; float contact_offset, max_depen, erp, cfm, terrain_mu; int solver, fric;
  unsigned slide_mask;      // bit sl: some leg's sphere in slot sl stands for a capsule part (lg_robot_model.cp_slide), kernel-uniform
  float cache_reach;        // triangle meshes: how far beyond `range` a closest-point query looks for the distance cache (LG_MESH_CACHE_REACH; DevCtx::mesh_reach)
  unsigned slot_perm;       // nibble p: the slot at position p of the detection deal (capsule instances; contact_detect_begin_caps)
};

#define LG_MESH_CONTACT_MARGIN 0.1f      // triangle-mesh contacts: how far below a surface a sphere's centre may have sunk and still be pushed out
#define LG_MESH_CACHE_REACH 0.05f      // (round 6: 0.15 -> 0.05, config 3 0.365 -> 0.345 ms: a wave pays for the widest window among its lanes, and windows grow with the square of the reach; LG_MESH_REACH: A/B)
// self-collision candidates: pairs[i] = leg a | slot a << 8 | leg b << 16 | slot b << 24 (global memory; read for the rare pairs that pass the filter);
// tab[i] = {slot-record offsets of the two spheres, radius a, radius b, filter threshold} (sc_prefilter; the workgroup's LDS copy where there is room for
// one); mask: LDS words [3][64] through which the helper waves hand their share of the filter to the main wave (null: this wave filters every pair).
struct SelfCol { const unsigned* pairs; int n; const uint4* tab; unsigned* mask; };

struct QuadState {           // per lane: replicated base + own leg (the name is from the four-legged instance)
  float root[13];            // pos3, quat xyzw, lin vel3, ang vel3 (world)
  float q[NJ], qd[NJ];
};

#if NJ == 3      // ---- the tuned three-joint kernels' own piece (lg_chain.h holds the six-joint instance's)
// Leg part of the bias forces: recursive Newton-Euler with zero generalised acceleration, moments about the base origin.
// Outputs the three joint bias torques and the leg's total force / moment (the caller adds the base and quad-sums).
LG_DEV void leg_bias(const LegModel& lm_, const LegKin& k, V3 pb, V3 wb, const float qd[3], V3 grav, float bk[3], V3& Fs, V3& Ns) {
  const float lm[3] = {lm_.f(LM_MASS), lm_.f(LM_MASS + 1), lm_.f(LM_MASS + 2)};
  V3 wp = wb, alp = v3(0, 0, 0), aOp = v3(0, 0, 0), Op = pb;
  V3 F[3], NP[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    V3 d = k.O[j] - Op;
    V3 aO = aOp + cross(alp, d) + cross(wp, cross(wp, d));
    V3 al = alp + qd[j] * cross(wp, k.ax[j]);
    V3 w = k.w[j];
    V3 rc = k.com[j] - k.O[j];
    V3 ac = aO + cross(al, rc) + cross(w, cross(w, rc));
    F[j] = lm[j] * (ac - grav);
    NP[j] = mul(k.Ic[j], al) + cross(w, mul(k.Ic[j], w)) + cross(k.com[j] - pb, F[j]);
    wp = w; alp = al; aOp = aO; Op = k.O[j];
  }
  Fs = v3(0, 0, 0); Ns = v3(0, 0, 0);
#pragma unroll
  for (int j = 2; j >= 0; --j) {
    Fs = Fs + F[j]; Ns = Ns + NP[j];
    bk[j] = dot(k.ax[j], Ns - cross(k.O[j] - pb, Fs));
  }
}

// Contact detection for slots [S0, S1): sphere centre, terrain surface under it, gap, activation; results go to the
// LDS slot table (CF_ACTIVE, CF_N, CF_R, CF_GAP, zeroed impulses).  Unrolled and branch-free so the lookups overlap.
// Two halves: `begin` computes the sphere centres and issues the height-sample loads, `finish` consumes them — the
// helper waves run their actuator network in between, which hides the memory latency of the lookups.
template <int S0, int S1>
struct ContactProbe { V3 xs[S1 - S0]; float rads[S1 - S0]; TerrainCell cell[S1 - S0]; };
template <int S0, int S1>
LG_DEV void contact_detect_begin(const LegModel& lm_, const TerrainView& T, const LegKin& k, const M3& Rb, V3 pb, ContactProbe<S0, S1>& pr) {
#pragma unroll
  for (int sl = S0; sl < S1; ++sl) {
    const int link = lm_.i(LM_CP_LINK + sl);
    const V3 lp = lm_.v(LM_CP_POS + 3 * sl);
    pr.rads[sl - S0] = lm_.f(LM_CP_RADIUS + sl);
    const V3 xb = pb + mul(Rb, lp), x0 = k.O[0] + mul(k.R[0], lp), x1 = k.O[1] + mul(k.R[1], lp), x2 = k.O[2] + mul(k.R[2], lp);
    pr.xs[sl - S0] = link < 0 ? xb : (link == 0 ? x0 : (link == 1 ? x1 : x2));
  }
#pragma unroll
  for (int i = 0; i < S1 - S0; ++i) pr.cell[i] = terrain_fetch(T, pr.xs[i].x, pr.xs[i].y);
}
template <int S0, int S1>
LG_DEV void contact_detect_finish(const LegModel& lm_, const TerrainView& T, const PhysParams& P, V3 pb, const ContactProbe<S0, S1>& pr,
                                  float* cst, int lane) {
  const int ncp = lm_.i(LM_CP_COUNT);
#pragma unroll
  for (int sl = S0; sl < S1; ++sl) {
    float hh; V3 n;
    terrain_eval(T, pr.cell[sl - S0], &hh, &n);
    const V3 x = pr.xs[sl - S0];
    const float phi = (x.z - hh) * n.z - pr.rads[sl - S0];
    const bool active = (sl < ncp) && (phi < P.contact_offset);
    const V3 r = (x - pr.rads[sl - S0] * n) - pb;
    CS4(sl, 0) = make_float4(n.x, n.y, n.z, phi);
    CS4(sl, 1) = make_float4(r.x, r.y, r.z, active ? 1.f : 0.f);
    CS4(sl, 2) = make_float4(0.f, 0.f, 0.f, 0.f);                // multipliers (field 9, 1 / Ann, comes from the set-up)
    const unsigned long long am = __ballot(active);
    if (lane == 0) AMASK(sl) = am;
  }
}
template <int S0, int S1>
LG_DEV void contact_detect(const LegModel& lm_, const TerrainView& T, const PhysParams& P, const LegKin& k, const M3& Rb, V3 pb,
                           float* cst, int lane) {
  ContactProbe<S0, S1> pr;
  contact_detect_begin<S0, S1>(lm_, T, k, Rb, pb, pr);
  contact_detect_finish<S0, S1>(lm_, T, P, pb, pr, cst, lane);
}

#endif           // NJ == 3
// ---- capsule segments (lg_robot_model.cp_slide = the vector from a sphere of a capsule's chain to the next one, link frame; zero: none).  Against the
// piecewise-planar surface of a height grid the FACES of the terrain meet a capsule at its spheres first; what passes between two spheres is a convex
// EDGE -- and the creases of the surface lie on the grid lines.  So a slot with a segment has up to three candidates: its sphere (as ever), and the
// segment [x, x + g] against the piece of the first x line and of the first y line its ground track crosses (the two height samples on that line: an
// exact segment-segment closest-point pair, normal = the direction between the two points).  An edge candidate takes the slot when it is deeper than the
// sphere by more than 10 um.  The oracle restates the same rule.  Plane terrains have no lines: the plain instance serves them.
LG_DEV V3 sel3(bool c, V3 a, V3 b) { return v3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }   // (a conditional between two V3 OBJECTS may be compiled as a
                                                                                                  //  conditional between their addresses: scratch)
struct PackedCell { unsigned h01, h23; };
LG_DEV PackedCell terrain_fetch_packed(const TerrainView& T, float x, float y) {
  const TerrainCell c = terrain_fetch(T, x, y);
  PackedCell p;
  p.h01 = (unsigned)(unsigned short)c.h0 | ((unsigned)(unsigned short)c.h1 << 16);
  p.h23 = (unsigned)(unsigned short)c.h2 | ((unsigned)(unsigned short)c.h3 << 16);
  return p;
}
LG_DEV TerrainCell terrain_unpack(const TerrainView& T, float x, float y, const PackedCell& p) {
  TerrainCell c;
  const float ihs = T.hscale > 0.f ? frcp(T.hscale) : 0.f;
  const float fx = (x + T.border) * ihs, fy = (y + T.border) * ihs;
  int i = (int)floorf(fx), j = (int)floorf(fy);
  i = max(0, min(i, T.rows - 2)); j = max(0, min(j, T.cols - 2));
  c.u = fminf(fmaxf(fx - (float)i, 0.f), 1.f); c.v = fminf(fmaxf(fy - (float)j, 0.f), 1.f);
  c.h0 = (int16_t)(p.h01 & 0xffffu); c.h1 = (int16_t)(p.h01 >> 16); c.h2 = (int16_t)(p.h23 & 0xffffu); c.h3 = (int16_t)(p.h23 >> 16);
  return c;
}
// The piece of a grid line a segment's ground track crosses first.  axis 0: a line of constant x (index L along the rows, piece [j, j + 1] along the
// columns); axis 1: constant y.  f0 / g0: grid coordinates of the segment's start along / across the axis, df / dg: of its vector.
struct EdgePiece { bool on; int L, j; };
LG_DEV EdgePiece caps_edge_piece(float f0, float df, float g0, float dg, int nL, int nJ) {
  EdgePiece e;
  const float fl0 = floorf(f0), fl1 = floorf(f0 + df);
  e.on = fl0 != fl1;
  const float L = df > 0.f ? fl0 + 1.f : fl0;
  const float sc = e.on ? (L - f0) * frcp(df) : 0.f;      // (which piece of the line: a last-bit difference picks the neighbour piece, which shares the vertex there)
  e.L = (int)L;
  e.j = (int)floorf(fmaf(sc, dg, g0));
  e.on = e.on && e.L >= 0 && e.L <= nL - 1 && e.j >= 0 && e.j <= nJ - 2;
  e.L = max(0, min(e.L, nL - 1)); e.j = max(0, min(e.j, nJ - 2));
  return e;
}
// closest points of the segments A0 + s d1 and E0 + u d2 (s, u in [0, 1]); d2 is never degenerate here (a grid step long)
LG_DEV void seg_seg_closest(V3 A0, V3 d1, V3 E0, V3 d2, V3* A, V3* E) {
  const V3 r = A0 - E0;
  const float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r), c = dot(d1, r), b = dot(d1, d2);
  const float den = a * e - b * b;
  float sN = den > 1e-12f * a * e ? fminf(fmaxf((b * f - c * e) / den, 0.f), 1.f) : 0.f;
  float u = (b * sN + f) / e;
  const float uc = fminf(fmaxf(u, 0.f), 1.f);
  if (uc != u) sN = a > 0.f ? fminf(fmaxf((b * uc - c) / a, 0.f), 1.f) : 0.f;
  *A = A0 + sN * d1; *E = E0 + uc * d2;
}
#if NJ == 3      // ---- the tuned three-joint kernels' own piece (lg_chain.h holds the six-joint instance's)
template <int S0, int S1>
struct ContactProbeC { V3 x[S1 - S0], gv[S1 - S0]; float rads[S1 - S0]; PackedCell cell[S1 - S0]; unsigned e[S1 - S0][2]; int lj[S1 - S0][2]; /* L | j << 16, bit 31: no crossing */ };
// perm: which slot stands at position p of the deal (nibble p).  The waves take POSITIONS [S0, S1); the host orders the slots so that the ones with a
// segment (twice the work: two edge pieces each) are spread over the waves that have room -- one to the main wave, one to wave 3, whose second position
// is usually empty, none to wave 1, which has the leg bias (PhysParams::slot_perm; results do not depend on who detects a slot).
template <int S0, int S1>
LG_DEV void contact_detect_begin_caps(const LegModel& lm_, const TerrainView& T, const LegKin& k, const M3& Rb, V3 pb, unsigned slide_mask, ContactProbeC<S0, S1>& pr,
                                      unsigned perm = 0x76543210u) {
  const float ihs = T.hscale > 0.f ? frcp(T.hscale) : 0.f;
#pragma unroll
  for (int ps = S0; ps < S1; ++ps) {
    const int i = ps - S0, sl = (int)((perm >> (4 * ps)) & 7u);
    const int link = lm_.i(LM_CP_LINK + sl);
    const V3 lp = lm_.v(LM_CP_POS + 3 * sl);
    pr.rads[i] = lm_.f(LM_CP_RADIUS + sl);
    const V3 xb = pb + mul(Rb, lp), x0 = k.O[0] + mul(k.R[0], lp), x1 = k.O[1] + mul(k.R[1], lp), x2 = k.O[2] + mul(k.R[2], lp);
    const V3 x = sel3(link < 0, xb, sel3(link == 0, x0, sel3(link == 1, x1, x2)));
    pr.x[i] = x; pr.gv[i] = v3(0, 0, 0); pr.e[i][0] = 0u; pr.e[i][1] = 0u; pr.lj[i][0] = (int)0x80000000; pr.lj[i][1] = (int)0x80000000;
    pr.cell[i] = terrain_fetch_packed(T, x.x, x.y);
    if ((slide_mask >> sl) & 1u) {                      // (kernel-uniform)
      const V3 ls = lm_.v(LM_CP_SLIDE + 3 * sl);
      const V3 gv = sel3(link < 0, mul(Rb, ls), sel3(link == 0, mul(k.R[0], ls), sel3(link == 1, mul(k.R[1], ls), mul(k.R[2], ls))));
      pr.gv[i] = gv;
      const float fx0 = (x.x + T.border) * ihs, fy0 = (x.y + T.border) * ihs, dfx = gv.x * ihs, dfy = gv.y * ihs;
      const EdgePiece px = caps_edge_piece(fx0, dfx, fy0, dfy, T.rows, T.cols), py = caps_edge_piece(fy0, dfy, fx0, dfx, T.cols, T.rows);
      const int16_t* hx = T.H + (size_t)px.L * T.cols + px.j;      // (L, j), (L, j + 1)
      const int16_t* hy = T.H + (size_t)py.j * T.cols + py.L;      // (j, L), (j + 1, L)
      const int16_t a0 = hx[0], a1 = hx[1], b0 = hy[0], b1 = hy[T.cols];
      pr.e[i][0] = (unsigned)(unsigned short)a0 | ((unsigned)(unsigned short)a1 << 16);
      pr.e[i][1] = (unsigned)(unsigned short)b0 | ((unsigned)(unsigned short)b1 << 16);
      pr.lj[i][0] = px.L | (px.j << 16) | (px.on ? 0 : (int)0x80000000);
      pr.lj[i][1] = py.L | (py.j << 16) | (py.on ? 0 : (int)0x80000000);
    }
  }
}
template <int S0, int S1>
LG_DEV void contact_detect_finish_caps(const LegModel& lm_, const TerrainView& T, const PhysParams& P, V3 pb, unsigned slide_mask, const ContactProbeC<S0, S1>& pr,
                                       float* cst, int lane, unsigned perm = 0x76543210u) {
  const int ncp = lm_.i(LM_CP_COUNT);
#pragma unroll
  for (int ps = S0; ps < S1; ++ps) {
    const int i = ps - S0, sl = (int)((perm >> (4 * ps)) & 7u);
    const float rad = pr.rads[i];
    float hh; V3 n;
    V3 x = pr.x[i];
    terrain_eval(T, terrain_unpack(T, x.x, x.y, pr.cell[i]), &hh, &n);
    float phi = (x.z - hh) * n.z - rad;
    if ((slide_mask >> sl) & 1u) {
      const V3 x0 = pr.x[i], gv = pr.gv[i];
      const float zlow = fminf(x0.z, x0.z + gv.z) - rad - P.contact_offset;     // nothing of the segment's capsule is lower than this
#pragma unroll
      for (int ax = 0; ax < 2; ++ax) {
        const int lj = pr.lj[i][ax];
        const unsigned pk = pr.e[i][ax];
        const float h0 = T.vscale * (float)(int16_t)(pk & 0xffffu), h1 = T.vscale * (float)(int16_t)(pk >> 16);
        // an edge piece wholly below the capsule cannot touch it: most substeps no lane of the wave has a candidate (wave-uniform skip)
        const bool cand = lj >= 0 && zlow < fmaxf(h0, h1);
        if (!__any(cand)) continue;
        const int L = lj & 0xffff, j = (lj >> 16) & 0x7fff;
        const float cl = (float)L * T.hscale - T.border, cj = (float)j * T.hscale - T.border;
        const V3 E0 = ax == 0 ? v3(cl, cj, h0) : v3(cj, cl, h0);
        const V3 d2 = ax == 0 ? v3(0.f, T.hscale, h1 - h0) : v3(T.hscale, 0.f, h1 - h0);
        V3 A, E; seg_seg_closest(x0, gv, E0, d2, &A, &E);
        const V3 d = A - E; const float dist = norm(d);
        const float sg = d.z >= 0.f ? 1.f : -1.f;
        const V3 ne = dist > 1e-9f ? (sg * frcp(dist)) * d : v3(0, 0, 1);
        const float pe = sg * dist - rad;
        const bool better = cand && pe < phi - 1e-5f;
        phi = better ? pe : phi; n = sel3(better, ne, n); x = sel3(better, A, x);
      }
    }
    const bool active = (sl < ncp) && (phi < P.contact_offset);
    const V3 r = (x - rad * n) - pb;
    CS4(sl, 0) = make_float4(n.x, n.y, n.z, phi);
    CS4(sl, 1) = make_float4(r.x, r.y, r.z, active ? 1.f : 0.f);
    CS4(sl, 2) = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned long long am = __ballot(active);
    if (lane == 0) AMASK(sl) = am;
  }
}
// begin + finish in one place (the main wave's share, the single-wave instances)
template <int S0, int S1>
LG_DEV void contact_detect_caps(const LegModel& lm_, const TerrainView& T, const PhysParams& P, const LegKin& k, const M3& Rb, V3 pb,
                                float* cst, int lane) {
  ContactProbeC<S0, S1> pr;
  const unsigned perm = (S0 == 0 && S1 == LG_MAX_CP) ? 0x76543210u : P.slot_perm;      // (all slots: any order)
  contact_detect_begin_caps<S0, S1>(lm_, T, k, Rb, pb, P.slide_mask, pr, perm);
  contact_detect_finish_caps<S0, S1>(lm_, T, P, pb, P.slide_mask, pr, cst, lane, perm);
}

#endif           // NJ == 3
// Closest point on a GRID mesh (lg_terrain.grid_vertices): the triangles of cell (i, j) are (v0, v3, v1) and (v0, v2, v3) with
// v0 = (i, j), v1 = (i, j+1), v2 = (i+1, j), v3 = (i+1, j+1), and the slope correction moved every vertex by at most one cell in x and
// y, so a cell's triangles lie inside [(i-1) hs, (i+2) hs] x [(j-1) hs, (j+2) hs]: the cells that can hold a point within R of p are
// an index range around p, scanned row by row (two new vertices per cell) with a box test per cell in front of the two triangle
// tests.  Same per-triangle arithmetic, tolerances and tie rule as `closest_point` (lg_bvh.h), which is independent of the order
// the triangles are met in: the result is the BVH's.  ~10-30 triangle tests for a foot on the ground instead of a tree walk with
// dependent 128-byte node fetches.
// (closest_grid_triangle: lg_bvh.h)
// Every stage below is a few ROUNDS of independent loads (indices clamped, loads unconditional) followed by arithmetic: a wave pays
// one L2 latency per round whatever its lanes need, and the first version's one-cell-at-a-time loop spent 36 k cycles per pair of
// queries on ~20 dependent rounds each.
LG_DEV void closest_point_grid(const TerrainView& T, ClosestQuery& A, int* visits = nullptr, unsigned long long* tdbg = nullptr) {
#ifdef LG_STAMPS
  unsigned long long tg0 = __builtin_amdgcn_s_memtime();
#define GSTAMP(k) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (tdbg) tdbg[k] += t_ - tg0; tg0 = t_; } while (0)
#else
#define GSTAMP(k)
#endif
  if (!A.on) return;
#if LG_AB == 1      // timing probe: no query at all (nothing is ever found)
  A.found = false; A.cp = A.p; A.fn = v3(0, 0, 1); return;
#endif
  // pointers read out of a struct are generic to the compiler (flat_load: counted on both wait counters, drained with vmcnt(0) and
  // lgkmcnt(0) together); these three are device-global by construction
  typedef float f4v __attribute__((ext_vector_type(4)));
  typedef const f4v __attribute__((address_space(1)))* gf4; typedef const float __attribute__((address_space(1)))* gf1;
  const gf4 GV4 = (gf4)T.GV4; const gf1 GM = (gf1)T.GM;
  auto ld4 = [](gf4 q) -> float4 { const f4v v = *q; return make_float4(v.x, v.y, v.z, v.w); };
  float R = A.max_dist; const float ihs = frcp(T.hscale);
  float bestabs = -1.f; bool found = false;
  V3 bestp = A.p, bestn = v3(0, 0, 1);
  const V3 p = A.p;
  const float gx = (p.x + T.border) * ihs, gy = (p.y + T.border) * ihs;
  float gr = R * ihs + 1e-3f;
  int i0 = (int)floorf(gx - gr) - 1, i1 = (int)floorf(gx + gr) + 1, j0 = (int)floorf(gy - gr) - 1, j1 = (int)floorf(gy + gr) + 1;
  i0 = max(i0, 0); j0 = max(j0, 0); i1 = min(i1, T.rows - 2); j1 = min(j1, T.cols - 2);
  A.found = false; A.cp = p; A.fn = v3(0, 0, 1);
  if (i0 > i1 || j0 > j1) return;
  int margin = 1;
  // the window that matters for a sphere whose centre is ABOVE the ground: only what lies within radius + contact_offset can touch it
  const float Rs = fminf(R, A.range + 1e-3f), grs = Rs * ihs + 1e-3f;
  const int si0 = max((int)floorf(gx - grs) - 1, 0), si1 = min((int)floorf(gx + grs) + 1, T.rows - 2);
  const int sj0 = max((int)floorf(gy - grs) - 1, 0), sj1 = min((int)floorf(gy + grs) + 1, T.cols - 2);
  {  // how far the sphere is above everything around it: the highest vertex of the 2 x 2 blocks that cover the window's vertices
     // (a vertex's height is its height sample; the slope correction moves x and y only).  More than radius + contact_offset: no
     // contact is possible, the exact closest point is not needed, and the caller's distance cache gets the proven lower bound.
    const int bi0 = i0 >> 1, bi1 = (i1 + 1) >> 1, bj0 = j0 >> 1, bj1 = (j1 + 1) >> 1;
    float top = -1e30f, top_s = -1e30f;
    unsigned moved = 0u;                     // last mantissa bit of a block's height: the block holds a vertex the slope correction moved (lg_create)
    for (int bi = bi0; bi <= bi1; bi += 4)
      for (int bj = bj0; bj <= bj1; bj += 8) {
        float tv[32];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int t = 0; t < 8; ++t) tv[8 * u + t] = GM[(size_t)min(bi + u, bi1) * T.mcols + min(bj + t, bj1)];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            top = fmaxf(top, tv[8 * u + t]);
            moved |= __float_as_uint(tv[8 * u + t]);
            const int b_i = min(bi + u, bi1), b_j = min(bj + t, bj1);
            const bool near = b_i >= (si0 >> 1) && b_i <= ((si1 + 1) >> 1) && b_j >= (sj0 >> 1) && b_j <= ((sj1 + 1) >> 1);
            top_s = fmaxf(top_s, near ? tv[8 * u + t] : -1e30f);
          }
      }
    const float clear = p.z - top;
    GSTAMP(22);
    if (clear > A.range) { A.lb = fminf(R, clear); return; }
    // No moved vertex among the vertices of the window: every triangle there lies inside its own cell's square, so the cells that can hold a point within
    // the radius are the ones the radius reaches -- no ring of one cell around them (the ring exists because a moved vertex carries its triangles up to one cell
    // over).  A sphere hovering a hand above the ground then scans one round of cells instead of four, a foot on the ground 4 cells instead of 9-16.
    margin = (moved & 1u) ? 1 : 0;
    if (!margin) {
      i0 = max((int)floorf(gx - gr), 0); i1 = min((int)floorf(gx + gr), T.rows - 2); j0 = max((int)floorf(gy - gr), 0); j1 = min((int)floorf(gy + gr), T.cols - 2);
    }
    // the centre is above every vertex near it: the sphere is not sunk into the ground (the margin in the caller's range is for
    // that case), so a contact needs a triangle within radius + contact_offset -- search that far only.  Nothing there: the cache
    // gets that radius as its lower bound.
    if (p.z > top_s + 1e-4f) {
      R = Rs; A.lb = Rs;
      i0 = max((int)floorf(gx - grs) - margin, 0); i1 = min((int)floorf(gx + grs) + margin, T.rows - 2);
      j0 = max((int)floorf(gy - grs) - margin, 0); j1 = min((int)floorf(gy + grs) + margin, T.cols - 2);
    }
  }
  float best2 = R * R;
  int ci = -1, cj = -1;
  // box of a cell's four vertices against the current best distance (cheap, unrolled over a round's cells) ...
  auto box_ok = [&](float4 a0, float4 a1, float4 b0, float4 b1) -> bool {
    const float lox = fminf(fminf(a0.x, a1.x), fminf(b0.x, b1.x)), hix = fmaxf(fmaxf(a0.x, a1.x), fmaxf(b0.x, b1.x));
    const float loy = fminf(fminf(a0.y, a1.y), fminf(b0.y, b1.y)), hiy = fmaxf(fmaxf(a0.y, a1.y), fmaxf(b0.y, b1.y));
    const float loz = fminf(fminf(a0.z, a1.z), fminf(b0.z, b1.z)), hiz = fmaxf(fmaxf(a0.z, a1.z), fmaxf(b0.z, b1.z));
    const float dx = fmaxf(fmaxf(lox - p.x, 0.f), p.x - hix), dy = fmaxf(fmaxf(loy - p.y, 0.f), p.y - hiy), dz = fmaxf(fmaxf(loz - p.z, 0.f), p.z - hiz);
    return dx * dx + dy * dy + dz * dz <= best2 * (1.f + 1e-5f) + 1e-12f;
  };
  // ... and the two exact triangle tests of a cell that passed: ONE instance of this code, run in a loop over the cells a round
  // selected (their vertices are re-read, now cache hits).  Inlined at every cell of the unrolled rounds it was ~3.6 k instructions.
  auto exact = [&](int i, int j) {
    if (visits) ++*visits;
    const gf4 ra = GV4 + (size_t)i * T.cols + j; const gf4 rb = ra + T.cols;
    const float4 a0 = ld4(ra), a1 = ld4(ra + 1), b0 = ld4(rb), b1 = ld4(rb + 1);
    const V3 v0 = v3(a0.x, a0.y, a0.z), v1 = v3(a1.x, a1.y, a1.z), v2 = v3(b0.x, b0.y, b0.z), v3_ = v3(b1.x, b1.y, b1.z);
#pragma unroll 1
    for (int h = 0; h < 2; ++h) closest_grid_triangle(p, v0, h == 0 ? v3_ : v2, h == 0 ? v1 : v3_, best2, found, bestabs, bestp, bestn);
  };
  {
    // the cell under the sphere first.  On most ground its triangles are the closest ones or nearly so, and whatever distance they
    // give bounds the search: the window shrinks to the cells that can hold something closer, and the box tests of the round(s)
    // below pass fewer neighbours (each passing cell costs two exact triangle tests, ~2 k cycles of a wave whose busiest lane
    // decides; with the radius-sized initial bound a foot near a cell border passed 4-6 cells).  The cell is met again by the scan:
    // a triangle met twice changes nothing, the update rules are idempotent.
    ci = max(i0, min((int)floorf(gx), i1)); cj = max(j0, min((int)floorf(gy), j1));
    exact(ci, cj);
    GSTAMP(23);
#if LG_AB == 2      // timing probe: the cell under the sphere only
    A.found = found; A.cp = bestp; A.fn = bestn; return;
#endif
    if (found) {
      gr = sqrtf(best2) * ihs * (1.f + 1e-4f) + 1e-3f;
      i0 = max(i0, (int)floorf(gx - gr) - margin); i1 = min(i1, (int)floorf(gx + gr) + margin);
      j0 = max(j0, (int)floorf(gy - gr) - margin); j1 = min(j1, (int)floorf(gy + gr) + margin);
    }
  }
  // rounds of 4 x 4 cells = 5 x 5 vertices, twenty-five 16-byte loads in flight: the usual window (a foot on the ground, a sphere
  // a few centimetres above it) is one round
#pragma unroll 1
  for (int i = i0; i <= i1; i += 4) {
#pragma unroll 1
    for (int jb = j0; jb <= j1; jb += 4) {
      float4 v[5][5];
#ifdef LG_STAMPS
      if (visits) *visits += 1000;        // diagnostic: thousands = rounds of this lane
#endif
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const gf4 rr = GV4 + (size_t)min(i + u, T.rows - 1) * T.cols;
#pragma unroll
        for (int t = 0; t < 5; ++t) v[u][t] = ld4(rr + min(jb + t, T.cols - 1));
      }
      GSTAMP(25);
      unsigned pass = 0u;
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (i + u <= i1 && jb + t <= j1 && !(i + u == ci && jb + t == cj) && box_ok(v[u][t], v[u][t + 1], v[u + 1][t], v[u + 1][t + 1])) pass |= 1u << (4 * u + t);   // (the cell under the sphere has had its exact test)
      GSTAMP(26);
#pragma unroll 1
      while (pass) {
        const int bit = __ffs(pass) - 1; pass &= pass - 1u;
        exact(i + (bit >> 2), jb + (bit & 3));
      }
    }
  }
  GSTAMP(24);
  A.found = found; A.cp = bestp; A.fn = bestn;
}

// (closest_point_lattice, the lane-by-lane form of the lattice query, lives in lg_bvh.h: the SDF kernel of lg_mesh.hip uses it too)
#if NJ == 3      // ---- the tuned three-joint kernels' own piece (lg_chain.h holds the six-joint instance's)
// Triangle-mesh terrain (LG_MESH_TRIMESH): the surface under a sphere is the closest point of the collision mesh within
// range = radius + contact_offset + LG_MESH_CONTACT_MARGIN (the margin lets a sphere whose centre has sunk below the surface
// still find it); normal = direction from that point to the sphere centre (flipped when the centre is behind the deciding
// face), gap = signed distance - radius.  Same slot-table outputs as above.
//
// `cq` (optional, LDS, [slot][4][lane]) caches per sphere the position and the unsigned surface distance of its last
// query, which looked LG_MESH_CACHE_REACH further than `range`.  While the sphere has moved less than that distance minus
// `range` since, no triangle can be within `range` and the traversal is skipped — an exact cull, the result is what
// the query would have returned (nothing).  Most spheres of a walking robot (trunk, hips, thighs) ride on it.
#define CQ(slot, f) cq[((slot) * 4 + (f)) * 64 + lane]
// `cq` (optional, LDS, [slot][4][lane], persisted per env between steps) caches per sphere the position and the unsigned
// surface distance of its last query (distance < 0: no entry).  Two exact uses of it:
//   * cull: the query looked LG_MESH_CACHE_REACH further than `range`; while the sphere has moved less than that distance
//     minus `range` since, no triangle can be within `range` and the traversal is skipped;
//   * bound: the surface cannot be farther than the cached distance plus the distance travelled, so the search starts
//     with that radius instead of the full reach - a foot in stance tests a handful of triangles, not the ~60 within
//     reach.  A reset teleports the sphere; the travel term then exceeds the reach and the bound falls back to it.
// The BVH traversal stack of a slot pair (sp0, sp0 + 1) lives in the SET-UP blocks of the two slots' records (dwords 12..59 of this lane's
// record: 48 >= BVH_STACK): dead while the slots are being detected -- the previous substep's sweeps are over, the set-up of this one
// comes after the rendezvous.  Keys in the even slot's block, node indices in the odd one's.
static_assert(CF_FIELDS - CF_SETUP >= BVH_STACK && (LG_MAX_CP % 2) == 0, "a slot record's set-up block holds one traversal stack");
LG_DEV float* mesh_stack_k(float* cst, int sp0, int lane) { return cst + ((sp0 * 64 + lane) * CF_FIELDS + CF_SETUP); }
LG_DEV int* mesh_stack_i(float* cst, int sp0, int lane) { return reinterpret_cast<int*>(cst + (((sp0 + 1) * 64 + lane) * CF_FIELDS + CF_SETUP)); }
// ---- the same queries, a PAIR per lane, with the faces of the whole wave's queries dealt over its 64 lanes.  Lane by lane (closest_point_lattice) the wave
// waits for its busiest lane: measured on config 3, 13 cells / 18 face fetches / 37 exact tests against a mean of 3.6 / 4.2 / 8.2.  Here every lane
//   1. lists the faces of the groups its queries reach in a table in LDS (space claimed with one LDS atomic add per cell) -- in the first round(s) the cell
//      under each sphere, whose distance then bounds the window, afterwards the window's cells, eight at a time,
//   2. tests the table's faces t = lane, lane + 64, ...: squared distance -> 64-bit atomic min per query (distance bits | face), then, among the faces within the
//      tolerance band of that minimum, the deciding normal -> atomic max (|plane distance| bits | face): `closest_point`'s tie rule, which is order-free,
//   3. and recomputes point and normal of its own queries' winners.
// The table lives in the set-up blocks of the wave's two slot records (dwords 12..59 of [slot][lane][CF_FIELDS], where the tree walk keeps its stacks): per
// lane and slot 8 dwords of query record (centre xyz, acceptance limit, min key, max key) and 13 table entries of 3 dwords (query | face, distance, |plane distance|).
// A table that fills up is tested and refilled (the lanes keep their place).
#define LATP_INVALID 0xffffffffu
#define LATP_BLOCK 8
struct LatQ { V3 p; float best2, prev; bool on; int i0, i1, j0, j1, ci, cj; float fx, fy; };
LG_DEV void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xC07F);                        // lgkmcnt(0); one wave's LDS operations complete in order
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
LG_DEV void closest_point_lattice_pair(const LatticeView& L, ClosestQuery& QA, ClosestQuery& QB, float* cst, int sp0, int lane, unsigned long long* dbg = nullptr) {
#ifdef LG_LATVIS
  unsigned long long tl0 = __builtin_amdgcn_s_memtime();
#define LSTAMP(k) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (dbg && lane == 0) dbg[k] += t_ - tl0; tl0 = t_; } while (0)
#else
#define LSTAMP(k)
#endif
  static_assert(CF_FIELDS - CF_SETUP >= 48, "a slot record's set-up block holds a query record and LATP_PER table entries");
  typedef unsigned u2v __attribute__((ext_vector_type(2))); typedef float f4v __attribute__((ext_vector_type(4)));
  typedef const u2v __attribute__((address_space(1)))* gu2; typedef const f4v __attribute__((address_space(1)))* gf4;
  const gf4 CELL = (gf4)L.cell; const gu2 RUN = (gu2)L.run; const gf4 TRI = (gf4)L.tris;
  const float ihx = frcp(L.hx), ihy = frcp(L.hy);
  auto frag = [&](int hh, int ln) -> float* { return cst + (((sp0 + hh) * 64 + ln) * CF_FIELDS + CF_SETUP); };
  auto qrec = [&](int qid) -> float* { return frag(qid & 1, qid >> 1); };
  auto ent = [&](int t) -> float* { const int fr = t / LATP_PER, w = t - LATP_PER * fr; return frag(fr >> 6, fr & 63) + 8 + 3 * w; };
  unsigned* const ctr = reinterpret_cast<unsigned*>(frag(0, 0) + 47);
  auto cell_ok = [&](V3 p, float best2, int i, int j, f4v z, u2v r) -> unsigned {
    const float xa = L.x0 + ((float)i - LATTICE_TOL) * L.hx, xb = L.x0 + ((float)(i + 1) + LATTICE_TOL) * L.hx;
    const float ya = L.y0 + ((float)j - LATTICE_TOL) * L.hy, yb = L.y0 + ((float)(j + 1) + LATTICE_TOL) * L.hy;
    const float dx = fmaxf(fmaxf(xa - p.x, 0.f), p.x - xb), dy = fmaxf(fmaxf(ya - p.y, 0.f), p.y - yb);
    const float dz0 = fmaxf(fmaxf(z.x - p.z, 0.f), p.z - z.y), dz1 = fmaxf(fmaxf(z.z - p.z, 0.f), p.z - z.w);
    const float dxy = dx * dx + dy * dy, lim = best2 * (1.f + 1e-5f) + 1e-12f;
    return (((r.y & 0xffffu) != 0u && dxy + dz0 * dz0 <= lim) ? 1u : 0u) | (((r.y >> 16) != 0u && dxy + dz1 * dz1 <= lim) ? 2u : 0u);
  };
  // the faces of a cell's passing group(s) are ONE run of the face list: claimed with one atomic add, written as (query | face) entries
  auto claim = [&](unsigned ok, u2v r, unsigned tag) -> bool {
    const int n0 = (int)(r.y & 0xffffu), n1 = (int)(r.y >> 16);
    const int first = (int)r.x + ((ok & 1u) ? 0 : n0), n = ((ok & 1u) ? n0 : 0) + ((ok & 2u) ? n1 : 0);
    const int off = (int)__hip_atomic_fetch_add(ctr, (unsigned)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (off + n > L.cap) {
      for (int t = off; t < L.cap; ++t) reinterpret_cast<unsigned*>(ent(t))[0] = LATP_INVALID;
      return false;
    }
    for (int kf = 0; kf < n; ++kf) reinterpret_cast<unsigned*>(ent(off + kf))[0] = tag | (unsigned)(first + kf);
    return true;
  };
  auto open = [&](ClosestQuery& A, LatQ& s) {
    s.on = A.on; s.p = A.p; s.best2 = A.max_dist * A.max_dist; s.prev = __builtin_inff();
    A.found = false; A.cp = A.p; A.fn = v3(0, 0, 1);
    s.fx = (s.p.x - L.x0) * ihx; s.fy = (s.p.y - L.y0) * ihy;
    const float grx = A.max_dist * ihx + 2.f * LATTICE_TOL, gry = A.max_dist * ihy + 2.f * LATTICE_TOL;
    s.i0 = max((int)floorf(s.fx - grx), 0); s.i1 = min((int)floorf(s.fx + grx), L.nx - 1); s.j0 = max((int)floorf(s.fy - gry), 0); s.j1 = min((int)floorf(s.fy + gry), L.ny - 1);
    if (s.i0 > s.i1 || s.j0 > s.j1) s.on = false;
    s.ci = min(max(s.i0, min((int)floorf(s.fx), s.i1)), L.nx - 1); s.cj = min(max(s.j0, min((int)floorf(s.fy), s.j1)), L.ny - 1);     // (inside the table also for a query that is off)
  };
  LatQ q0, q1;
  open(QA, q0); open(QB, q1);
  if (__ballot(q0.on || q1.on) == 0ull) return;
  {
    float* r0 = frag(0, lane); float* r1 = frag(1, lane);
    r0[0] = q0.p.x; r0[1] = q0.p.y; r0[2] = q0.p.z; r0[3] = q0.best2 * (1.f + 1e-5f) + 1e-12f;
    r1[0] = q1.p.x; r1[1] = q1.p.y; r1[2] = q1.p.z; r1[3] = q1.best2 * (1.f + 1e-5f) + 1e-12f;
    reinterpret_cast<unsigned long long*>(r0 + 4)[0] = 0x7f800000ull << 32; reinterpret_cast<unsigned long long*>(r0 + 4)[1] = 0ull;
    reinterpret_cast<unsigned long long*>(r1 + 4)[0] = 0x7f800000ull << 32; reinterpret_cast<unsigned long long*>(r1 + 4)[1] = 0ull;
  }
  LSTAMP(16);
  // round(s) A: the cells under the spheres (cpend: bit h = query h's centre cell still to list); rounds B: the windows, LATP_BLOCK cells of the row-major
  // window at a time (kb = first of them, pm = two bits per cell still to list)
  unsigned cpend = (q0.on ? 1u : 0u) | (q1.on ? 2u : 0u);
  bool centres = true;                       // wave-uniform
  int h = -1, kb = 0, ww = 1, wn = 0, wi0 = 0, wj0 = 0, wci = 0, wcj = 0; V3 wp = q0.p; float wbest = 0.f, wrw = 1.f; unsigned pm = 0u;
  bool lane_more = false;
#pragma unroll 1
  for (;;) {
    if (lane == 0) *ctr = 0u;
    wave_lds_sync();
    // ---- 1. list faces
    if (centres) {
      if (cpend) {
        const size_t c0 = (size_t)q0.cj * L.nx + q0.ci, c1 = (size_t)q1.cj * L.nx + q1.ci;
        const f4v z0 = CELL[c0], z1 = CELL[c1]; const u2v r0 = RUN[c0], r1 = RUN[c1];
        bool full = false;
        if (cpend & 1u) {
          const unsigned ok = cell_ok(q0.p, q0.best2, q0.ci, q0.cj, z0, r0);
          if (ok == 0u || claim(ok, r0, (unsigned)(2 * lane) << 25)) cpend &= ~1u; else full = true;
        }
        if ((cpend & 2u) && !full) {
          const unsigned ok = cell_ok(q1.p, q1.best2, q1.ci, q1.cj, z1, r1);
          if (ok == 0u || claim(ok, r1, (unsigned)(2 * lane + 1) << 25)) cpend &= ~2u;
        }
      }
    } else {
      bool full = false;
#pragma unroll 1
      while (lane_more && !full) {
        const bool fresh = pm == 0u;
        if (fresh) {
          bool have = false;
          if (h >= 0) { kb += LATP_BLOCK; have = kb < wn; }
          if (!have) {
            ++h; if (h == 0 && !q0.on) ++h; if (h == 1 && !q1.on) ++h;
            if (h >= 2) { lane_more = false; break; }
            const bool s1 = h == 1;
            wp = s1 ? q1.p : q0.p; wbest = s1 ? q1.best2 : q0.best2; wi0 = s1 ? q1.i0 : q0.i0; wj0 = s1 ? q1.j0 : q0.j0;
            ww = (s1 ? q1.i1 : q0.i1) - wi0 + 1; wn = ww * ((s1 ? q1.j1 : q0.j1) - wj0 + 1); wrw = 1.f / (float)ww;
            wci = s1 ? q1.ci : q0.ci; wcj = s1 ? q1.cj : q0.cj;
            kb = 0;
          }
        }
        // (a lane that found the table full comes back to its cells with pm = the cells still to list, and fetches the records again.  LATP_BLOCK = 8:
        //  with sixteen records in registers the instance spilled 57 VGPRs -- in every phase of the kernel, not only here)
        f4v z[LATP_BLOCK]; u2v r[LATP_BLOCK];
#pragma unroll
        for (int c = 0; c < LATP_BLOCK; ++c) {
          const int k = min(kb + c, wn - 1), jj = (int)(((float)k + 0.5f) * wrw), ii = k - jj * ww;
          const size_t cidx = (size_t)(wj0 + jj) * L.nx + wi0 + ii;
          z[c] = CELL[cidx]; r[c] = RUN[cidx];
        }
        if (fresh) {
#pragma unroll
          for (int c = 0; c < LATP_BLOCK; ++c) {
            const int k = kb + c, jj = (int)(((float)k + 0.5f) * wrw), ci_ = wi0 + k - jj * ww, cj_ = wj0 + jj;
            if (k < wn && !(ci_ == wci && cj_ == wcj)) pm |= cell_ok(wp, wbest, ci_, cj_, z[c], r[c]) << (2 * c);
          }
        }
        const unsigned tag = (unsigned)(2 * lane + h) << 25;
#pragma unroll
        for (int c = 0; c < LATP_BLOCK; ++c) {
          const unsigned ok = (pm >> (2 * c)) & 3u;
          if (ok == 0u || full) continue;
          if (claim(ok, r[c], tag)) pm &= ~(3u << (2 * c)); else full = true;
        }
      }
    }
    wave_lds_sync();
    const int T = (int)min(*ctr, (unsigned)L.cap);
    LSTAMP(17);
#ifdef LG_LATVIS
    if (dbg && lane == 0) { dbg[30] += T; dbg[31] += 1; }
#endif
    // ---- 2a. distances
#pragma unroll 1
    for (int t = lane; t < T; t += 64) {
      float* e = ent(t);
      const unsigned code = reinterpret_cast<unsigned*>(e)[0];
      float d2o = -1.f, abo = 0.f;
      if (code != LATP_INVALID) {
        const int qid = (int)(code >> 25), f = (int)(code & 0x1ffffffu);
        float* qr = qrec(qid);
        const gf4 Tp = TRI + (size_t)f * 3;
        const f4v ta = Tp[0], tb = Tp[1], tc = Tp[2];
        const float4 pr = *reinterpret_cast<const float4*>(qr);
        const float run = __uint_as_float(reinterpret_cast<const unsigned*>(qr)[5]);          // the query's best so far (high word of the min key)
        const float lim = fminf(pr.w, run * (1.f + 1e-5f) + 1e-12f);
        const V3 p = v3(pr.x, pr.y, pr.z), a = v3(ta.x, ta.y, ta.z), b = v3(tb.x, tb.y, tb.z), cc = v3(tc.x, tc.y, tc.z);
        if (tri_box_dist2(p, a, b, cc) <= lim) {
          const V3 fn = cross(b - a, cc - a); const float fl = norm(fn);
          if (fl > 1e-10f) {
            const V3 qp = closest_on_triangle(p, a, b, cc);
            const V3 dq = p - qp; const float d2 = dot(dq, dq);
            if (d2 <= lim) {
              const V3 nh = (1.f / fl) * fn;
              const float sd = dot(dq, nh);
              abo = fabsf(sd) * (sd > 0.f ? 1.001f : 1.f); d2o = d2;
              __hip_atomic_fetch_min(reinterpret_cast<unsigned long long*>(qr + 4), ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)(f + 1),
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
          }
        }
      }
      e[1] = d2o; e[2] = abo;
    }
    wave_lds_sync();
    LSTAMP(18);
    // the owner: a strictly smaller distance than before this table drops the normals chosen so far
    const bool last_centres = centres && __ballot(cpend != 0u) == 0ull;
    {
      const float n0d = __uint_as_float(reinterpret_cast<const unsigned*>(frag(0, lane))[5]), n1d = __uint_as_float(reinterpret_cast<const unsigned*>(frag(1, lane))[5]);
      if (n0d < q0.prev * (1.f - 1e-5f) - 1e-12f) reinterpret_cast<unsigned long long*>(frag(0, lane) + 4)[1] = 0ull;
      if (n1d < q1.prev * (1.f - 1e-5f) - 1e-12f) reinterpret_cast<unsigned long long*>(frag(1, lane) + 4)[1] = 0ull;
      q0.prev = n0d; q1.prev = n1d;
      if (last_centres) {
        // every centre cell has been tested: its distance bounds the window (a window of one cell is that cell: done)
        auto shrink = [&](LatQ& s, float d) {
          if (!s.on) return;
          if (d < __builtin_inff()) {
            s.best2 = d;
            const float rr = sqrtf(d) * (1.f + 1e-4f), grx = rr * ihx + 2.f * LATTICE_TOL, gry = rr * ihy + 2.f * LATTICE_TOL;
            s.i0 = max(s.i0, (int)floorf(s.fx - grx)); s.i1 = min(s.i1, (int)floorf(s.fx + grx)); s.j0 = max(s.j0, (int)floorf(s.fy - gry)); s.j1 = min(s.j1, (int)floorf(s.fy + gry));
          }
          if (s.i0 == s.i1 && s.j0 == s.j1) s.on = false;
        };
        shrink(q0, n0d); shrink(q1, n1d);
        lane_more = q0.on || q1.on;
      }
    }
    wave_lds_sync();
    // ---- 2b. the deciding normal among the faces within the band of the minimum
#pragma unroll 1
    for (int t = lane; t < T; t += 64) {
      float* e = ent(t);
      const unsigned code = reinterpret_cast<unsigned*>(e)[0];
      const float d2 = e[1];
      if (code == LATP_INVALID || !(d2 >= 0.f)) continue;
      float* qr = qrec((int)(code >> 25));
      const float dmin = __uint_as_float(reinterpret_cast<const unsigned*>(qr)[5]);
      if (d2 <= dmin * (1.f + 1e-5f) + 1e-12f)
        __hip_atomic_fetch_max(reinterpret_cast<unsigned long long*>(qr + 6), ((unsigned long long)__float_as_uint(e[2]) << 32) | (0xffffffffu - ((code & 0x1ffffffu) + 1u)),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    wave_lds_sync();
    LSTAMP(28);
    if (last_centres) centres = false;
    if (!centres && __ballot(lane_more) == 0ull) break;
  }
  // ---- 3. the winners of this lane's queries
  auto close = [&](ClosestQuery& A, const LatQ& s, const float* qr) {
    if (!A.on) return;
    const unsigned long long kmin = reinterpret_cast<const unsigned long long*>(qr + 4)[0], kab = reinterpret_cast<const unsigned long long*>(qr + 4)[1];
    const unsigned fmin = (unsigned)kmin, fab = kab ? 0xffffffffu - (unsigned)kab : 0u;     // face + 1; 0: none
    if (fmin) {
      const gf4 Tp = TRI + (size_t)(fmin - 1u) * 3;
      const f4v ta = Tp[0], tb = Tp[1], tc = Tp[2];
      A.cp = closest_on_triangle(s.p, v3(ta.x, ta.y, ta.z), v3(tb.x, tb.y, tb.z), v3(tc.x, tc.y, tc.z));
      A.found = true;
    }
    if (fab) {
      const gf4 Tp = TRI + (size_t)(fab - 1u) * 3;
      const f4v ta = Tp[0], tb = Tp[1], tc = Tp[2];
      const V3 a = v3(ta.x, ta.y, ta.z), fn = cross(v3(tb.x, tb.y, tb.z) - a, v3(tc.x, tc.y, tc.z) - a);
      A.fn = (1.f / norm(fn)) * fn;
    }
  };
  close(QA, q0, frag(0, lane)); close(QB, q1, frag(1, lane));
  wave_lds_sync();                                                     // (the records are the caller's again)
}

template <bool MCAPS = false>
LG_DEV void contact_detect_mesh(int s0, int s1, const LegModel& lm_, const TerrainView& T, const PhysParams& P, const LegKin& k,
                                const M3& Rb, V3 pb, float* cst, int lane, float* cq = nullptr, unsigned long long* dbg = nullptr) {
  const int ncp = lm_.i(LM_CP_COUNT);
  // slots in pairs: one BVH traversal serves two neighbouring spheres (closest_point_pair)
#pragma unroll 1
  for (int sp0 = s0; sp0 < s1; sp0 += 2) {
    V3 xs[2]; float rads[2], ranges[2], reaches[2]; ClosestQuery Q[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // POSITION sp0 + h of the deal holds slot `sl` (P.slot_perm: which spheres share a wave is the host's choice, lg_create; the scratch records and the
      // distance cache are indexed by position, everything the model and the solver see by slot)
      const int ps = sp0 + h, sl = (int)((P.slot_perm >> (4 * ps)) & 7u);
      xs[h] = pb; rads[h] = 0.f; ranges[h] = 0.f; reaches[h] = 0.f;
      Q[h].p = pb; Q[h].max_dist = 0.f; Q[h].on = false; Q[h].found = false; Q[h].cp = pb; Q[h].fn = v3(0, 0, 1); Q[h].range = 0.f; Q[h].lb = 0.f;
      if (ps < s1 && sl < ncp) {
        const int link = lm_.i(LM_CP_LINK + sl);
        const V3 lp = lm_.v(LM_CP_POS + 3 * sl);
        rads[h] = lm_.f(LM_CP_RADIUS + sl);
        const V3 x = link < 0 ? pb + mul(Rb, lp) : (link == 0 ? k.O[0] + mul(k.R[0], lp) : (link == 1 ? k.O[1] + mul(k.R[1], lp) : k.O[2] + mul(k.R[2], lp)));
        xs[h] = x;
        // (capsule segments, lg_robot_model.cp_slide: against the edges of a GRID mesh below (MCAPS); on other meshes the spheres stand alone -- a closest-point
        //  query of a segment is not built, and sliding the sphere to the segment point nearest ONE mesh point loses the contact of its own end)
        const float range = rads[h] + P.contact_offset + LG_MESH_CONTACT_MARGIN;
        const float reach = cq ? range + P.cache_reach : range;
        ranges[h] = range; reaches[h] = reach;
        bool query = true; float bound = reach;
        if (cq) {
          const float dq = CQ(ps, 3);
          if (dq >= 0.f) {
            const float travel = norm(x - v3(CQ(ps, 0), CQ(ps, 1), CQ(ps, 2)));
            query = !(travel < dq - range);
            bound = fminf(reach, dq + travel * 1.0001f + 1e-4f);
          }
        }
        // grid meshes: a sphere higher above everything around it than radius + contact_offset cannot touch (the margin in `range`
        // is for centres that have sunk BELOW the surface)
        Q[h].p = x; Q[h].max_dist = bound; Q[h].on = query; Q[h].range = rads[h] + P.contact_offset; Q[h].lb = bound;
      }
    }
#ifdef LG_STAMPS
    {
      int visits = 0;
      const unsigned long long tq0 = __builtin_amdgcn_s_memtime();
      if (T.GV) { closest_point_grid(T, Q[0], &visits, (dbg && lane == 0) ? dbg : nullptr); closest_point_grid(T, Q[1], &visits, (dbg && lane == 0) ? dbg : nullptr); }
#if 1
      else if (T.L.cell) closest_point_lattice_pair(T.L, Q[0], Q[1], cst, sp0, lane, dbg);
#else
      else if (T.L.cell) {
        unsigned long long v64 = 0ull;
        closest_point_lattice(T.L, Q[0], &v64); closest_point_lattice(T.L, Q[1], &v64);
        int f[3] = {(int)(v64 & 0x1fffffull), (int)((v64 >> 21) & 0x1fffffull), (int)(v64 >> 42)}, fs[3], fm[3];
        for (int q = 0; q < 3; ++q) { fs[q] = fm[q] = f[q]; for (int off = 32; off > 0; off >>= 1) { fm[q] = max(fm[q], __shfl_xor(fm[q], off)); fs[q] += __shfl_xor(fs[q], off); } }
        if (dbg && lane == 0) { dbg[16] += fs[0]; dbg[17] += fs[1]; dbg[18] += fs[2]; dbg[28] += fm[0]; dbg[30] += fm[1]; dbg[31] += fm[2]; }
      }
#endif
      else closest_point_pair_t<true>(T.M, Q[0], Q[1], mesh_stack_k(cst, sp0, lane), mesh_stack_i(cst, sp0, lane), &visits);
      __builtin_amdgcn_s_waitcnt(0);
      if (dbg && lane == 0) dbg[53] += __builtin_amdgcn_s_memtime() - tq0;
      // diagnostic: queries issued / traversal steps (sum and max over the wave) of workgroup 0, wave 2
      int vmax = visits, vsum = visits;
      for (int off = 32; off > 0; off >>= 1) { vmax = max(vmax, __shfl_xor(vmax, off)); vsum += __shfl_xor(vsum, off); }
      const int nq = __popcll(__ballot(Q[0].on)) + __popcll(__ballot(Q[1].on));
      if (dbg && lane == 0) { dbg[54] += nq; dbg[55] += vsum; dbg[56] += vmax; dbg[57] += 1; }
    }
#else
    if (T.GV) { closest_point_grid(T, Q[0]); closest_point_grid(T, Q[1]); }
    else if (T.L.cell) closest_point_lattice_pair(T.L, Q[0], Q[1], cst, sp0, lane);
    else closest_point_pair_t<true>(T.M, Q[0], Q[1], mesh_stack_k(cst, sp0, lane), mesh_stack_i(cst, sp0, lane), nullptr);
#endif
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ps = sp0 + h, sl = (int)((P.slot_perm >> (4 * ps)) & 7u);
      if (ps >= s1) continue;
      bool active = false; V3 n = v3(0, 0, 1); float phi = 1.f;
      const V3 x = xs[h]; const float rad = rads[h];
      if (Q[h].on) {
        // nothing found: nothing lies within the radius that was searched (= `reach` whenever the cached distance was exact; a grid
        // mesh may return a smaller proven bound)
        const V3 diff = x - Q[h].cp; const float dist = Q[h].found ? norm(diff) : (T.GV ? Q[h].lb : reaches[h]);
        if (cq) { CQ(ps, 0) = x.x; CQ(ps, 1) = x.y; CQ(ps, 2) = x.z; CQ(ps, 3) = dist; }
        if (Q[h].found && dist <= ranges[h]) {
          const float sign = dot(diff, Q[h].fn) < 0.f ? -1.f : 1.f;
          n = dist > 1e-6f ? (sign / dist) * diff : Q[h].fn;
          phi = sign * dist - rad;
          active = phi < P.contact_offset;
        }
      }
      V3 xc = x;
      if (MCAPS && sl < ncp && ((P.slide_mask >> sl) & 1u)) {                     // (kernel-uniform)
        // Capsule segments on a GRID mesh (round 6; T.SEG4 = lg_terrain.grid_vertices): the height grid's rule (contact_detect_*_caps) with the mesh's own
        // edges -- the segment [x, x + gv] against the mesh edge (L, j)-(L, j + 1) of the first lattice line of each axis its ground track crosses; the edge's
        // end points are the mesh's vertices, wherever the slope correction put them.  The candidate takes the slot when it is
        // deeper than what the sphere found within its range -- nothing: phi = 1 -- by more than 10 um.  Evaluated whatever the sphere's distance cache said: the
        // cache speaks for the sphere's centre, not for the far end of its segment.
        typedef float f4v __attribute__((ext_vector_type(4))); typedef const f4v __attribute__((address_space(1)))* gf4;
        const gf4 SG = (gf4)T.SEG4;
        const int link = lm_.i(LM_CP_LINK + sl);
        const V3 ls = lm_.v(LM_CP_SLIDE + 3 * sl);
        const V3 gv = sel3(link < 0, mul(Rb, ls), sel3(link == 0, mul(k.R[0], ls), sel3(link == 1, mul(k.R[1], ls), mul(k.R[2], ls))));
        const float ihs = T.hscale > 0.f ? frcp(T.hscale) : 0.f;
        const float fx0 = (x.x + T.border) * ihs, fy0 = (x.y + T.border) * ihs, dfx = gv.x * ihs, dfy = gv.y * ihs;
        const EdgePiece px = caps_edge_piece(fx0, dfx, fy0, dfy, T.rows, T.cols), py = caps_edge_piece(fy0, dfy, fx0, dfx, T.cols, T.rows);
        const gf4 ex = SG + (size_t)px.L * T.cols + px.j;      // (L, j), (L, j + 1)
        const gf4 ey = SG + (size_t)py.j * T.cols + py.L;      // (j, L), (j + 1, L)
        const f4v a0 = ex[0], a1 = ex[1], b0 = ey[0], b1 = ey[T.cols];
        const float zlow = fminf(x.z, x.z + gv.z) - rad - P.contact_offset;       // nothing of the segment's capsule is lower than this
#pragma unroll
        for (int ax = 0; ax < 2; ++ax) {
          const f4v e0 = ax == 0 ? a0 : b0, e1 = ax == 0 ? a1 : b1;
          const V3 E0 = v3(e0.x, e0.y, e0.z), d2 = v3(e1.x - e0.x, e1.y - e0.y, e1.z - e0.z);
          const bool cand = (ax == 0 ? px.on : py.on) && zlow < fmaxf(e0.z, e1.z) && d2.x * d2.x + d2.y * d2.y > 0.25f * T.hscale * T.hscale;   // (an edge the correction stood upright -- the seam of two wall faces -- or collapsed is no crease)
          if (!__any(cand)) continue;                                             // most substeps no lane of the wave has a candidate
          V3 A, E; seg_seg_closest(x, gv, E0, cand ? d2 : v3(0.f, T.hscale, 0.f), &A, &E);
          const V3 d = A - E; const float dist = norm(d);
          const float sg = d.z >= 0.f ? 1.f : -1.f;
          const V3 ne = dist > 1e-9f ? (sg * frcp(dist)) * d : v3(0, 0, 1);
          const float pe = sg * dist - rad;
          const bool better = cand && pe < phi - 1e-5f && (sg > 0.f || dist <= rad);      // (an axis below the edge by more than the radius is the spheres' business: on a mesh with walls "below the edge" need not mean "under the surface")
          phi = better ? pe : phi; n = sel3(better, ne, n); xc = sel3(better, A, xc);
        }
        active = phi < P.contact_offset;
      }
      const V3 r = (xc - rad * n) - pb;
      CS4(sl, 0) = make_float4(n.x, n.y, n.z, phi);
      CS4(sl, 1) = make_float4(r.x, r.y, r.z, active ? 1.f : 0.f);
      CS4(sl, 2) = make_float4(0.f, 0.f, 0.f, 0.f);
      const unsigned long long am = __ballot(active);
      if (lane == 0) AMASK(sl) = am;
    }
  }
}

// Solver data of contact slot `sl` (contact frame, Jacobian pieces, A = J M^-1 J^T, the M^-1 J^T columns), from the
// detection results in the slot table and this leg's share of the factorised mass matrix.  Any wave of the workgroup
// that holds the leg kinematics and (Mi, Mbk, Y, Si) can run it: the slots of one substep are dealt to all four waves.
LG_DEV void contact_setup_slot(int sl, const LegModel& lm_, const LegKin& k, V3 pb, const float Mi[6], const float Mbk[6][3],
                               const float Y[3][6], const float Si[21], float cfm, int fric, float* cst, int lane) {
  const int ncp = lm_.i(LM_CP_COUNT);
  int lk = -1;
  if (sl < ncp) { int link = lm_.i(LM_CP_LINK + sl); lk = link < 0 ? -1 : (link > 2 ? 2 : link); }
  const float4 n4 = CS4(sl, 0), r4 = CS4(sl, 1);
  const V3 n = v3(n4.x, n4.y, n4.z), r = v3(r4.x, r4.y, r4.z);
  const V3 p = r + pb;
  float out[CF_FIELDS - CF_SETUP];   // the set-up block of the slot record, stored below as 12 x 16 B
#define OUT(f) out[(f) - CF_SETUP]
  // contact frame and Jacobian pieces (computed on every lane of the wave; inactive lanes carry harmless values)
  V3 a0 = fabsf(n.x) < 0.57735f ? v3(1, 0, 0) : v3(0, 1, 0);
  V3 t1 = cross(a0, n); t1 = __builtin_amdgcn_rsqf(dot(t1, t1)) * t1;
  V3 t2 = cross(n, t1);
  V3 jk[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) jk[j] = j <= lk ? cross(k.ax[j], p - k.O[j]) : v3(0, 0, 0);
  // A = J M^-1 J^T in the contact frame (rows n, t1, t2)
  V3 dirs[3] = {n, t1, t2};
  float Wb[3][6], Wk[3][3];   // M^-1 J^T columns: base part and own-leg part
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    V3 d = dirs[c];
    V3 rd = cross(r, d);
    float jkv[3] = {dot(jk[0], d), dot(jk[1], d), dot(jk[2], d)}, z[3];
    sym3_mul(Mi, jkv, z);
    float g[6] = {d.x, d.y, d.z, rd.x, rd.y, rd.z};
#pragma unroll
    for (int a = 0; a < 6; ++a) g[a] -= Mbk[a][0] * z[0] + Mbk[a][1] * z[1] + Mbk[a][2] * z[2];
    symv6(Si, g, Wb[c]);
#pragma unroll
    for (int a = 0; a < 6; ++a) OUT(CF_WB + 6 * c + a) = Wb[c][a];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float w = z[j];
#pragma unroll
      for (int a = 0; a < 6; ++a) w -= Y[j][a] * Wb[c][a];
      Wk[c][j] = w;
    }
    OUT(CF_ZCP + 2 * c) = z[0]; OUT(CF_ZCP + 2 * c + 1) = z[1]; OUT(CF_ZC2 + c) = z[2];
  }
  float A[3][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    V3 d = dirs[b]; V3 rd = cross(r, d);
    float jb[6] = {d.x, d.y, d.z, rd.x, rd.y, rd.z};
    float jkv[3] = {dot(jk[0], d), dot(jk[1], d), dot(jk[2], d)};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float sacc = jkv[0] * Wk[c][0] + jkv[1] * Wk[c][1] + jkv[2] * Wk[c][2];
#pragma unroll
      for (int a = 0; a < 6; ++a) sacc += jb[a] * Wb[c][a];
      A[b][c] = sacc;
    }
  }
  OUT(CF_T12) = t1.x; OUT(CF_T12 + 1) = t2.x; OUT(CF_T12 + 2) = t1.y; OUT(CF_T12 + 3) = t2.y; OUT(CF_T12 + 4) = t1.z; OUT(CF_T12 + 5) = t2.z;
#pragma unroll
  for (int j = 0; j < 3; ++j) { OUT(CF_JK0 + 3 * j) = jk[j].x; OUT(CF_JK0 + 3 * j + 1) = jk[j].y; OUT(CF_JK0 + 3 * j + 2) = jk[j].z; }
  // the sweeps only ever divide by these: store the reciprocal of the normal row and the inverse of the 2x2 tangential
  // block (computed once here, by whichever wave sets the slot up, instead of in each of the four sweeps)
  const float a11 = A[1][1] + cfm, a12 = A[1][2], a22 = A[2][2] + cfm;
  const float idet = frcp(a11 * a22 - a12 * a12);
  CS4(sl, 2) = make_float4(0.f, frcp(A[0][0] + cfm), 0.f, 0.f);       // (L0, 1 / Ann, L1, L2): the multipliers are still zero here
  OUT(CF_AN12) = A[1][0]; OUT(CF_AN12 + 1) = A[2][0];
  const bool pyr = fric != LG_FRICTION_CONE;
  OUT(CF_B) = pyr ? frcp(a11) : a22 * idet; OUT(CF_B + 1) = pyr ? a12 : -a12 * idet; OUT(CF_B + 2) = pyr ? a12 : -a12 * idet; OUT(CF_B + 3) = pyr ? frcp(a22) : a11 * idet;
#undef OUT
  float4* dst = reinterpret_cast<float4*>(&CS(sl, CF_SETUP));
#pragma unroll
  for (int i = 0; i < (CF_FIELDS - CF_SETUP) / 4; ++i) dst[i] = make_float4(out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]);
}

// per lane: its active slots packed four bits each, lowest first; *count = how many.  The sweeps walk these lists, so a
// sweep takes max-over-lanes(count) steps instead of one step per slot that is active anywhere in the wave.
LG_DEV unsigned active_slot_list(const float* cst, int lane, int* count) {
  unsigned list = 0; int n = 0;
#pragma unroll
  for (int sl = 0; sl < LG_MAX_CP; ++sl) {
    const bool a = ((AMASK(sl) >> lane) & 1ull) != 0ull;
    list |= a ? (unsigned)sl << (4 * n) : 0u;
    n += a ? 1 : 0;
  }
  *count = n;
  return list;
}

// wave-uniform: slots with at least one active contact in this wave
LG_DEV unsigned active_slot_mask(const float* cst, int lane) {
  unsigned slot_mask = 0;
#pragma unroll
  for (int sl = 0; sl < LG_MAX_CP; ++sl)
    if (__builtin_amdgcn_readfirstlane(AMASK(sl) != 0ull ? 1 : 0)) slot_mask |= 1u << sl;
  return slot_mask;
}

// How the contact set-up of one substep is shared between the waves of a workgroup: this wave takes every n-th active slot.
struct SlotShare { int n, me; bool late; };   // late: the rendezvous after the set-up is held right in front of the sweeps
// (Mi 6 | Mbk 18 | Y 18 | Si 21) of every lane, published by the main wave for the helper waves: [field][lane]
#define XS_FIELDS 63
#define XS_STRIDE 68   // dwords per lane: 16-B aligned rows, 17 (odd) 16-B units -> conflict-free ds_read/write_b128
LG_DEV void publish_mass_factors(float* xs, int lane, const float Mi[6], const float Mbk[6][3], const float Y[3][6], const float Si[21]) {
  float rec[64];
#pragma unroll
  for (int i = 0; i < 6; ++i) rec[i] = Mi[i];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int j = 0; j < 3; ++j) { rec[6 + a * 3 + j] = Mbk[a][j]; rec[24 + j * 6 + a] = Y[j][a]; }
#pragma unroll
  for (int i = 0; i < 21; ++i) rec[42 + i] = Si[i];
  rec[63] = 0.f;
  float4* p = reinterpret_cast<float4*>(xs + lane * XS_STRIDE);
#pragma unroll
  for (int i = 0; i < 16; ++i) p[i] = make_float4(rec[4 * i], rec[4 * i + 1], rec[4 * i + 2], rec[4 * i + 3]);
}
LG_DEV void fetch_mass_factors(const float* xs, int lane, float Mi[6], float Mbk[6][3], float Y[3][6], float Si[21]) {
  float rec[64];
  const float4* p = reinterpret_cast<const float4*>(xs + lane * XS_STRIDE);
#pragma unroll
  for (int i = 0; i < 16; ++i) { float4 v = p[i]; rec[4 * i] = v.x; rec[4 * i + 1] = v.y; rec[4 * i + 2] = v.z; rec[4 * i + 3] = v.w; }
#pragma unroll
  for (int i = 0; i < 6; ++i) Mi[i] = rec[i];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int j = 0; j < 3; ++j) { Mbk[a][j] = rec[6 + a * 3 + j]; Y[j][a] = rec[24 + j * 6 + a]; }
#pragma unroll
  for (int i = 0; i < 21; ++i) Si[i] = rec[42 + i];
}

// Layout of a lane's record (floats): everything the helper waves' slot set-up multiplies PAIRWISE sits in (even, odd) pairs, so that a
// ds_read_b128 delivers two operands of a v_pk_fma_f32 each:
//   0..5  Mi (3 x 3 symmetric, packed) | 6, 7 unused
//   8 + 2 (3 j + p)   (Mbk[2p][j], Mbk[2p+1][j])           j = 0..2 joints, p = 0..2 pairs of base coordinates
//   26 + 2 a          (Y[0][a], Y[1][a])                    a = 0..5
//   38 + a            Y[2][a]
//   44 + 2 (3 b + p)  (S^-1[2p][b], S^-1[2p+1][b])          the full symmetric 6 x 6, column b in three pairs
// (heightfield / plane instances; the triangle-mesh instances keep the scalar record below: their LDS has no room for the 16 extra dwords per lane)
#define XS_FIELDS_PK 80
#define XS_STRIDE_PK 84   // dwords per lane: 16-B aligned rows, 21 (odd) 16-B units -> conflict-free ds_read/write_b128
struct MassFactorsP { float Mi[6]; pk2 Mbk[3][3]; pk2 Y01[6]; float Y2[6]; pk2 Si[6][3]; };
LG_DEV void publish_mass_factors_pk(float* xs, int lane, const float Mi[6], const float Mbk[6][3], const float Y[3][6], const float Si[21]) {
  float rec[XS_FIELDS_PK];
#pragma unroll
  for (int i = 0; i < 6; ++i) rec[i] = Mi[i];
  rec[6] = 0.f; rec[7] = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int pp = 0; pp < 3; ++pp) { rec[8 + 2 * (3 * j + pp)] = Mbk[2 * pp][j]; rec[8 + 2 * (3 * j + pp) + 1] = Mbk[2 * pp + 1][j]; }
#pragma unroll
  for (int a = 0; a < 6; ++a) { rec[26 + 2 * a] = Y[0][a]; rec[26 + 2 * a + 1] = Y[1][a]; rec[38 + a] = Y[2][a]; }
#pragma unroll
  for (int b = 0; b < 6; ++b)
#pragma unroll
    for (int pp = 0; pp < 3; ++pp) {
      const int r0 = 2 * pp, r1 = 2 * pp + 1;
      rec[44 + 2 * (3 * b + pp)] = r0 >= b ? Si[LT(r0, b)] : Si[LT(b, r0)];
      rec[44 + 2 * (3 * b + pp) + 1] = r1 >= b ? Si[LT(r1, b)] : Si[LT(b, r1)];
    }
  float4* p = reinterpret_cast<float4*>(xs + lane * XS_STRIDE_PK);
#pragma unroll
  for (int i = 0; i < XS_FIELDS_PK / 4; ++i) p[i] = make_float4(rec[4 * i], rec[4 * i + 1], rec[4 * i + 2], rec[4 * i + 3]);
}
LG_DEV void fetch_mass_factors_pk(const float* xs, int lane, MassFactorsP& F) {
  float rec[XS_FIELDS_PK];
  const float4* p = reinterpret_cast<const float4*>(xs + lane * XS_STRIDE_PK);
#pragma unroll
  for (int i = 0; i < XS_FIELDS_PK / 4; ++i) { float4 v = p[i]; rec[4 * i] = v.x; rec[4 * i + 1] = v.y; rec[4 * i + 2] = v.z; rec[4 * i + 3] = v.w; }
#pragma unroll
  for (int i = 0; i < 6; ++i) F.Mi[i] = rec[i];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int pp = 0; pp < 3; ++pp) { F.Mbk[j][pp].x = rec[8 + 2 * (3 * j + pp)]; F.Mbk[j][pp].y = rec[8 + 2 * (3 * j + pp) + 1]; }
#pragma unroll
  for (int a = 0; a < 6; ++a) { F.Y01[a].x = rec[26 + 2 * a]; F.Y01[a].y = rec[26 + 2 * a + 1]; F.Y2[a] = rec[38 + a]; }
#pragma unroll
  for (int b = 0; b < 6; ++b)
#pragma unroll
    for (int pp = 0; pp < 3; ++pp) { F.Si[b][pp].x = rec[44 + 2 * (3 * b + pp)]; F.Si[b][pp].y = rec[44 + 2 * (3 * b + pp) + 1]; }
}

// contact_setup_slot on packed fp32 (the helper waves' form: their factors come from LDS already paired).  Same quantities and the same
// record as the scalar form above (which stays for the main wave, whose factors live in scalar registers); the sums are formed in another
// order, so the two agree to rounding (1e-7 relative), not bit for bit.  ~290 instead of ~450 instructions per slot.
LG_DEV void contact_setup_slot_pk(int sl, const LegModel& lm_, const LegKin& k, V3 pb, const MassFactorsP& F, float cfm, int fric, float* cst, int lane) {
  const int ncp = lm_.i(LM_CP_COUNT);
  int lk = -1;
  if (sl < ncp) { int link = lm_.i(LM_CP_LINK + sl); lk = link < 0 ? -1 : (link > 2 ? 2 : link); }
  const float4 n4 = CS4(sl, 0), r4 = CS4(sl, 1);
  const V3 n = v3(n4.x, n4.y, n4.z), r = v3(r4.x, r4.y, r4.z);
  const V3 p = r + pb;
  float out[CF_FIELDS - CF_SETUP];
#define OUT(f) out[(f) - CF_SETUP]
  V3 a0 = fabsf(n.x) < 0.57735f ? v3(1, 0, 0) : v3(0, 1, 0);
  V3 t1 = cross(a0, n); t1 = __builtin_amdgcn_rsqf(dot(t1, t1)) * t1;
  V3 t2 = cross(n, t1);
  V3 jk[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) jk[j] = j <= lk ? cross(k.ax[j], p - k.O[j]) : v3(0, 0, 0);
  const V3 dirs[3] = {n, t1, t2};
  pk2 J[3][3], jk01[3], Wb[3][3], Wk01[3]; float jk2[3], Wk2[3];       // per direction: the base row (d, r x d) in pairs, the joint row, the responses
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const V3 d = dirs[c], rd = cross(r, d);
    J[c][0].x = d.x; J[c][0].y = d.y; J[c][1].x = d.z; J[c][1].y = rd.x; J[c][2].x = rd.y; J[c][2].y = rd.z;
    const float jkv[3] = {dot(jk[0], d), dot(jk[1], d), dot(jk[2], d)};
    jk01[c].x = jkv[0]; jk01[c].y = jkv[1]; jk2[c] = jkv[2];
    float z[3];
    sym3_mul(F.Mi, jkv, z);
    pk2 g[3] = {J[c][0], J[c][1], J[c][2]};
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int pp = 0; pp < 3; ++pp) g[pp] = pk_fma(-F.Mbk[j][pp], pk_splat(z[j]), g[pp]);
    pk2 w[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const float gb = (b & 1) ? g[b >> 1].y : g[b >> 1].x;
#pragma unroll
      for (int pp = 0; pp < 3; ++pp) w[pp] = pk_fma(F.Si[b][pp], pk_splat(gb), w[pp]);
    }
    pk2 wk = {z[0], z[1]}; float wk2 = z[2];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const float wa = (a & 1) ? w[a >> 1].y : w[a >> 1].x;
      wk = pk_fma(-F.Y01[a], pk_splat(wa), wk);
      wk2 -= F.Y2[a] * wa;
    }
#pragma unroll
    for (int pp = 0; pp < 3; ++pp) { Wb[c][pp] = w[pp]; OUT(CF_WB + 6 * c + 2 * pp) = w[pp].x; OUT(CF_WB + 6 * c + 2 * pp + 1) = w[pp].y; }
    Wk01[c] = wk; Wk2[c] = wk2;
    OUT(CF_ZCP + 2 * c) = z[0]; OUT(CF_ZCP + 2 * c + 1) = z[1]; OUT(CF_ZC2 + c) = z[2];
  }
  // A[b][c] = J_b . W_c over the base coordinates and the leg's joints
  auto Aij = [&](int b, int c) {
    pk2 acc = J[b][0] * Wb[c][0];
    acc = pk_fma(J[b][1], Wb[c][1], acc); acc = pk_fma(J[b][2], Wb[c][2], acc); acc = pk_fma(jk01[b], Wk01[c], acc);
    return acc.x + acc.y + jk2[b] * Wk2[c];
  };
  const float A00 = Aij(0, 0), A10 = Aij(1, 0), A20 = Aij(2, 0), A11 = Aij(1, 1), A12 = Aij(1, 2), A22 = Aij(2, 2);
  OUT(CF_T12) = t1.x; OUT(CF_T12 + 1) = t2.x; OUT(CF_T12 + 2) = t1.y; OUT(CF_T12 + 3) = t2.y; OUT(CF_T12 + 4) = t1.z; OUT(CF_T12 + 5) = t2.z;
#pragma unroll
  for (int j = 0; j < 3; ++j) { OUT(CF_JK0 + 3 * j) = jk[j].x; OUT(CF_JK0 + 3 * j + 1) = jk[j].y; OUT(CF_JK0 + 3 * j + 2) = jk[j].z; }
  const float a11 = A11 + cfm, a12 = A12, a22 = A22 + cfm;
  const float idet = frcp(a11 * a22 - a12 * a12);
  CS4(sl, 2) = make_float4(0.f, frcp(A00 + cfm), 0.f, 0.f);
  OUT(CF_AN12) = A10; OUT(CF_AN12 + 1) = A20;
  const bool pyr = fric != LG_FRICTION_CONE;
  OUT(CF_B) = pyr ? frcp(a11) : a22 * idet; OUT(CF_B + 1) = pyr ? a12 : -a12 * idet; OUT(CF_B + 2) = pyr ? a12 : -a12 * idet; OUT(CF_B + 3) = pyr ? frcp(a22) : a11 * idet;
#undef OUT
  float4* dst = reinterpret_cast<float4*>(&CS(sl, CF_SETUP));
#pragma unroll
  for (int i = 0; i < (CF_FIELDS - CF_SETUP) / 4; ++i) dst[i] = make_float4(out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]);
}

// ---- self-collision (lg_config.self_collisions; oracle: the PairRow block of simulate_env).  Candidate sphere pairs (leg a, slot a, leg b, slot b), one
// packed word each; every substep the lanes of a group test a share each against the sphere centres the contact detection left in the slot records
// (centre = base + r + radius * n), the two deepest pairs closer than contact_offset become frictionless unilateral rows between the two bodies.  Both
// sides of such a row act on the same point, so its base part cancels: the row has joint entries only, like a joint-limit row -- z = Mkk^-1 f per leg,
// base response S^-1 (-sum Mbk z), joint response z - Y W_b -- and is relaxed behind the terrain contacts of every pass.
struct SelfRow { bool on; float phi, iA, lam; V3 n; float f[3], Wb[6], Wk[3]; int slot_a, slot_b; /* fbody index of each side on THIS lane, -1: not mine */ };
// The filter of the self-collision pass: bit k of the result = "pair lgi + GRP k of this lane's share may be closer than contact_offset".  A lane's pairs
// are the indices lgi, lgi + GRP, ...; of their ordinals k this call takes share, share + nshare, ... (the four waves of a workgroup a quarter each, between
// rendezvous (A2) and (A3)).  Per pair: the two spheres' records (2 x 2 ds_read_b128; centre - base = r + radius n, the base cancels in the difference) and a
// squared distance against the entry's threshold (radius a + radius b + contact_offset)^2 (1 + 1e-4) -- no root, conservative; three pairs per round, loads
// batched.  Robots walk with their links decimetres apart: the mask is zero nearly always, and the exact pass behind it then does not run.
LG_DEV unsigned sc_prefilter(const float* cst, const uint4* tab, int n, int lane, int share, int nshare) {
#if LG_AB == 41      // (timing probe: no filter work at all; the rows' code stays)
  return 0u;
#endif
  const int gb = lane & ~(GRP - 1), lgi = lane & (GRP - 1);
  const float4* rec = reinterpret_cast<const float4*>(cst);
  const unsigned gb4 = (unsigned)gb * (CF_FIELDS / 4);
  unsigned mask = 0u;
  for (int k = share; k * GRP < n; k += 3 * nshare) {
    uint4 en[3]; float4 na[3], pa[3], nb[3], pq[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) { const int i = lgi + GRP * (k + u * nshare); en[u] = tab[i < n ? i : 0]; }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const unsigned oa = gb4 + (en[u].x & 0xFFFFu) * (CF_FIELDS / 4), ob = gb4 + (en[u].x >> 16) * (CF_FIELDS / 4);
      na[u] = rec[oa]; pa[u] = rec[oa + 1]; nb[u] = rec[ob]; pq[u] = rec[ob + 1];
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int kk = k + u * nshare;
      const float ra = __uint_as_float(en[u].y), rb = __uint_as_float(en[u].z);
      const float dx = (pa[u].x - pq[u].x) + (ra * na[u].x - rb * nb[u].x), dy = (pa[u].y - pq[u].y) + (ra * na[u].y - rb * nb[u].y),
                  dz = (pa[u].z - pq[u].z) + (ra * na[u].z - rb * nb[u].z);
      const bool near = lgi + GRP * kk < n && dx * dx + dy * dy + dz * dz < __uint_as_float(en[u].w);
      mask |= near ? 1u << kk : 0u;
    }
  }
  return mask;
}
LG_DEV V3 sc_sphere(const float* cst, const LegModel& lm_, int gb, int leg, int slot, V3 pb, float* rad) {
  const float4 n4 = reinterpret_cast<const float4*>(cst + ((slot) * 64 + gb + leg) * CF_FIELDS)[0];
  const float4 r4 = reinterpret_cast<const float4*>(cst + ((slot) * 64 + gb + leg) * CF_FIELDS)[1];
  const float r = lm_.t[(LM_CP_RADIUS + slot) * GRP + leg];
  *rad = r;
  return v3(pb.x + r4.x + r * n4.x, pb.y + r4.y + r * n4.y, pb.z + r4.z + r * n4.z);
}

// One physics step of length P.dt for the env this quad owns.  tau_fn(tau[3]) delivers this leg's joint torques; it is
// called after everything that does not depend on them (kinematics, bias, mass matrix, contact set-up).
// fbody[5] (optional) receives the net contact force on {base (already quad-summed), link0, link1, link2, foot}.
// prep_fn(bk, Fs, Ns) returns true when helper waves have produced the leg bias and the contact detection (it then
// holds the rendezvous and fills the three outputs); false means this wave computes them itself.
// With helper waves on a heightfield the contact detection is dealt two slots per wave; this wave takes slots
// [0, MAIN_DETECT) before the rendezvous (0: the helpers, or the inline path, detect everything).
// SPEC = 1: the instance of the reference's own solver settings (sim.physx.solver_type = 1: TGS, PhysX's pyramid friction rows) with both
// choices fixed at compile time; SPEC = 0 reads them from the parameters (the unified step evaluates both friction forms and selects).
// FEAT: bit 0 = capsule parts (sliding spheres, contact_detect_*_caps), bit 1 = the self-collision pass -- compile-time, so that the instance without
// them is the kernel it was before they existed.
template <bool TMESH, int MAIN_DETECT, bool ALLOW_INLINE, int SPEC = 0, int FEAT = 0, class TauFn, class PrepFn, class ShareFn>
LG_DEV void physics_substep(const lg_robot_model* __restrict__ m, const LegModel& lm_, const TerrainView& T, const PhysParams& P,
                            int lane, float* cst, QuadState& s, TauFn tau_fn, PrepFn prep_fn, ShareFn share_fn, SlotShare share,
                            float* xs, float mu_robot, float madd, V3* fbody, unsigned long long* stamps = nullptr,
                            float* cq = nullptr, SelfCol scol = SelfCol{nullptr, 0, nullptr, nullptr}) {
  STAMP_DECL
  const float dt = P.dt;
  const V3 pb = v3(s.root[0], s.root[1], s.root[2]);
  const V3 vb = v3(s.root[7], s.root[8], s.root[9]), wb = v3(s.root[10], s.root[11], s.root[12]);
  const M3 Rb = quat_to_mat(s.root + 3);
  LegKin k;
  leg_kinematics(lm_, Rb, pb, vb, wb, s.q, s.qd, k);

  STAMP(1);
  // mesh terrains: this wave's share of the contact detection goes HERE, where only the state and the kinematics are live; after the
  // mass matrix and its factors it sat in the middle of ~500 live registers and the inlined closest-point scan spilled them
  // (~800 scratch loads in the substep of the main wave)
#if defined(LG_STAMPS) && defined(LG_STAMP_MAIN_MESH)
  if (TMESH && MAIN_DETECT > 0 && share.n > 1) contact_detect_mesh<(FEAT & 1) != 0>(MAIN_DETECT - 100, MAIN_DETECT - 98, lm_, T, P, k, Rb, pb, cst, lane, cq, stamps);   // diagnostic: the query counters watch this wave
#else
  if (TMESH && MAIN_DETECT > 0 && share.n > 1) contact_detect_mesh<(FEAT & 1) != 0>(MAIN_DETECT - 100, MAIN_DETECT - 98, lm_, T, P, k, Rb, pb, cst, lane, cq);   // (triangle-mesh instances: MAIN_DETECT = 100 + the first slot of this wave's pair)
#endif
  // ---------------------------------------------------------------- bias forces (RNEA, zero generalised acceleration)
  const float m0 = m->base_mass + madd, iscale = m0 * frcp(m->base_mass);
  const V3 rc0 = mul(Rb, ld3(m->base_com));
  S3 I0 = rotate_inertia(Rb, m->base_inertia);
  I0.xx *= iscale; I0.xy *= iscale; I0.xz *= iscale; I0.yy *= iscale; I0.yz *= iscale; I0.zz *= iscale;
  float lm[3] = {lm_.f(LM_MASS), lm_.f(LM_MASS + 1), lm_.f(LM_MASS + 2)};
  STAMP(2);
  // ---------------------------------------------------------------- joint-space inertia (CRBA) and its factorisation
  float Mkk[6];            // 00 01 02 11 12 22
  float Mbk[6][3];
  float mc = 0; V3 hc = v3(0, 0, 0); S3 Icp = S3{0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 2; i >= 0; --i) {
    V3 r = k.com[i] - pb;
    mc += lm[i]; hc = hc + lm[i] * r; Icp = Icp + inertia_about(k.Ic[i], lm[i], r);
    V3 w = k.ax[i], o = k.O[i] - pb;
    V3 F = cross(w, hc - mc * o);
    V3 Nn = mul(Icp, w) - cross(hc, cross(w, o));
    Mbk[0][i] = F.x; Mbk[1][i] = F.y; Mbk[2][i] = F.z; Mbk[3][i] = Nn.x; Mbk[4][i] = Nn.y; Mbk[5][i] = Nn.z;
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      float v = dot(k.ax[j], Nn - cross(k.O[j] - pb, F));
      Mkk[j == 0 ? i : (j == 1 ? 2 + i : 5)] = v;     // (0,i) -> i ; (1,i) -> 3 + (i-1) ; (2,2) -> 5
    }
  }
  float Mi[6]; sym3_inverse(Mkk, Mi);
  float Y[3][6];
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    float col[3] = {Mbk[a][0], Mbk[a][1], Mbk[a][2]}, y[3];
    sym3_mul(Mi, col, y);
    Y[0][a] = y[0]; Y[1][a] = y[1]; Y[2][a] = y[2];
  }
  float L[21];
  {
    const float mt = m0 + grp_sum(mc);
    const V3 ht = m0 * rc0 + grp_sum(hc);
    const S3 It = inertia_about(I0, m0, rc0) + grp_sum(Icp);
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) {
        float sk = Mbk[a][0] * Y[0][b] + Mbk[a][1] * Y[1][b] + Mbk[a][2] * Y[2][b];
        L[LT(a, b)] = -grp_sum(sk);
      }
    L[LT(0, 0)] += mt; L[LT(1, 1)] += mt; L[LT(2, 2)] += mt;
    // rows 3..5, cols 0..2 : [h]x
    L[LT(3, 1)] += -ht.z; L[LT(3, 2)] += ht.y;
    L[LT(4, 0)] += ht.z;  L[LT(4, 2)] += -ht.x;
    L[LT(5, 0)] += -ht.y; L[LT(5, 1)] += ht.x;
    L[LT(3, 3)] += It.xx; L[LT(4, 3)] += It.xy; L[LT(5, 3)] += It.xz;
    L[LT(4, 4)] += It.yy; L[LT(5, 4)] += It.yz; L[LT(5, 5)] += It.zz;
#if LG_AB != 32
    chol6(L);
#endif
  }
  float Si[21];
#if LG_AB == 32      // measured and dropped (round 5, four same-session pairs): the inverse by 3 x 3 blocks -- two short chains instead of six dependent
                     // rsqrt rows -- leaves the step where it was (0.0806 vs 0.0808 ms; 0.0752 vs 0.0752 on the plain instance): the main wave has slack in front
                     // of rendezvous (A2), the helper waves arrive last
  spd6_inverse_blocks(L, Si);
#else
  spd6_inverse_from_chol(L, Si);
#endif
  if (share.n > 1) { if (TMESH) publish_mass_factors(xs, lane, Mi, Mbk, Y, Si); else publish_mass_factors_pk(xs, lane, Mi, Mbk, Y, Si); }

  STAMP(3);
  // ---------------------------------------------------------------- leg bias + contact detection: helper waves or inline
  float bk[3]; V3 Fs, Ns;
  if constexpr (!TMESH && MAIN_DETECT > 0) {
    if (share.n > 1) {
      if (FEAT & 1) contact_detect_caps<0, MAIN_DETECT>(lm_, T, P, k, Rb, pb, cst, lane);
      else contact_detect<0, MAIN_DETECT>(lm_, T, P, k, Rb, pb, cst, lane);
    }
  }
  STAMP(29);   // (diagnostic: this wave's own detection ends here; what follows in stamp 5 is the wait at the rendezvous)
  if (!prep_fn(bk, Fs, Ns)) {
    // (ALLOW_INLINE = false: a launch with helper waves never gets here; the inlined fallback, unreachable, still costs the main wave
    //  registers and code)
    if (ALLOW_INLINE) {
      leg_bias(lm_, k, pb, wb, s.qd, P.grav, bk, Fs, Ns);
      if (TMESH) contact_detect_mesh<(FEAT & 1) != 0>(0, LG_MAX_CP, lm_, T, P, k, Rb, pb, cst, lane);
      else if (FEAT & 1) contact_detect_caps<0, LG_MAX_CP>(lm_, T, P, k, Rb, pb, cst, lane);
      else contact_detect<0, LG_MAX_CP>(lm_, T, P, k, Rb, pb, cst, lane);
    }
  }
  float bb[6];
  {
    V3 ac = cross(wb, cross(wb, rc0));
    V3 Fb = m0 * (ac - P.grav);
    V3 Nb = cross(wb, mul(I0, wb)) + cross(rc0, Fb);
    V3 Ft = Fb + grp_sum(Fs), Nt = Nb + grp_sum(Ns);
    bb[0] = Ft.x; bb[1] = Ft.y; bb[2] = Ft.z; bb[3] = Nt.x; bb[4] = Nt.y; bb[5] = Nt.z;
  }
  const float mu = 0.5f * (mu_robot + P.terrain_mu);   // PhysX default friction combine mode: average
  const float idt_ = frcp(dt);
  const unsigned slot_mask = active_slot_mask(cst, lane);
#ifdef LG_STAMPS
  {  // ballots by the whole wave, accumulation by the stamping lane
    unsigned long long am = 0, g4 = 0;
    for (int sl = 0; sl < LG_MAX_CP; ++sl) {
      const unsigned long long b = AMASK(sl);
      am += __popcll(b);
      for (int w = 0; w < 4; ++w) g4 += ((b >> (16 * w)) & 0xffffull) ? 1 : 0;
    }
#ifndef LG_LATVIS
    if (stamps) { stamps[16] += __popc(slot_mask); stamps[17] += 1; stamps[18] += am; stamps[28] += g4;
                  stamps[30] += __popc(slot_mask) >= 4 ? 1 : 0; stamps[31] += __popc(slot_mask) >= 5 ? 1 : 0; }
#endif   // (heightfield runs: 30 / 31 are free)
  }
#endif
  STAMP(5);
  // pass B: per-contact solver data, only for slots some lane of the wave needs.  With helper waves the active slots are
  // dealt round-robin to the `share.n` waves of the workgroup (this wave is share.me); share_fn is the rendezvous after it.
  {
    int seen = 0;
#pragma unroll 1
    for (int sl = 0; sl < LG_MAX_CP; ++sl) {
      if (!((slot_mask >> sl) & 1u)) continue;
      if ((seen++ % share.n) != share.me) continue;
      contact_setup_slot(sl, lm_, k, pb, Mi, Mbk, Y, Si, P.cfm, P.fric, cst, lane);
    }
  }
  if (!share.late) share_fn();

  STAMP(6);
  // ---------------------------------------------------------------- unconstrained velocity v* = v + dt M^-1 (tau - c)
  float tau[3];
  tau_fn(tau);            // multi-wave builds: rendezvous with the actuator waves happens here
  STAMP(4);
  float vB[6] = {vb.x, vb.y, vb.z, wb.x, wb.y, wb.z};
  float vK[3] = {s.qd[0], s.qd[1], s.qd[2]};
  {
    float rk[3] = {tau[0] - bk[0], tau[1] - bk[1], tau[2] - bk[2]}, y[3];
    sym3_mul(Mi, rk, y);
    float g[6], g0[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) g0[a] = -bb[a] - grp_sum(Mbk[a][0] * y[0] + Mbk[a][1] * y[1] + Mbk[a][2] * y[2]);
    symv6(Si, g0, g);
#pragma unroll
    for (int a = 0; a < 6; ++a) vB[a] += dt * g[a];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float yk = y[j];
#pragma unroll
      for (int a = 0; a < 6; ++a) yk -= Y[j][a] * g[a];
      vK[j] += dt * yk;
    }
  }

  // ---------------------------------------------------------------- joint position limits as unilateral rows on qd
  // (URDF lower/upper; lower >= upper = unlimited).  Whole block is skipped unless some lane of the wave is near a limit.
  float jl_sgn[3], jl_gap[3], jl_iA[3], jl_lam[3] = {0.f, 0.f, 0.f}, jl_Wb[3][6], jl_y[3][3];
  bool jl_act[3];
  bool jl_any = false;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float lo = lm_.f(LM_LOWER + j), hi = lm_.f(LM_UPPER + j);
    const float glo = s.q[j] - lo, ghi = hi - s.q[j];
    const float gap = glo <= ghi ? glo : ghi;
    jl_sgn[j] = glo <= ghi ? 1.f : -1.f;
    jl_act[j] = (lo < hi) && (gap + fminf(0.f, dt * jl_sgn[j] * vK[j]) < 0.05f);   // near the limit, or about to cross it this step
    jl_gap[j] = gap;
    jl_any |= jl_act[j];
  }
  const bool jl_wave = __ballot(jl_any) != 0ull;
  // joint j's row is set up and relaxed only when some lane of the wave is near that joint's limit (wave-uniform): a robot whose hip-flexion and knee
  // joints are continuous (ANYmal) never pays for two of the three rows -- ~60 instructions each per pass of the sweeps
  bool jl_jw[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) jl_jw[j] = __ballot(jl_act[j]) != 0ull;
  if (jl_wave) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (!jl_jw[j]) {
        jl_iA[j] = 0.f;
#pragma unroll
        for (int a = 0; a < 6; ++a) jl_Wb[j][a] = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) jl_y[j][a] = 0.f;
        continue;
      }
      float ej[3] = {j == 0 ? jl_sgn[j] : 0.f, j == 1 ? jl_sgn[j] : 0.f, j == 2 ? jl_sgn[j] : 0.f};
      sym3_mul(Mi, ej, jl_y[j]);
      float gvec[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) gvec[a] = -(Mbk[a][0] * jl_y[j][0] + Mbk[a][1] * jl_y[j][1] + Mbk[a][2] * jl_y[j][2]);
      symv6(Si, gvec, jl_Wb[j]);
      float wj = jl_y[j][j];
#pragma unroll
      for (int a = 0; a < 6; ++a) wj -= Y[j][a] * jl_Wb[j][a];
      jl_iA[j] = frcp(jl_sgn[j] * wj + P.cfm);
    }
  }

  // ---------------------------------------------------------------- self-collision, part 1 (FEAT bit 1): this wave's share of the pair filter
  unsigned scm = 0u;
  if (FEAT & 2) scm = sc_prefilter(cst, scol.tab, scol.n, lane, 0, scol.mask ? 4 : 1);

  // ---------------------------------------------------------------- contact / limit rows: TGS sub-intervals or PGS sweeps
  int my_count; const unsigned my_list = active_slot_list(cst, lane, &my_count);
  int my_steps = 0;                                   // wave-uniform: the longest list
#pragma unroll
  for (int j = 0; j < LG_MAX_CP; ++j) my_steps += __ballot(my_count > j) != 0ull ? 1 : 0;
  // With helper waves this wave is the last to be dealt a slot of the set-up (share.late): the unconstrained velocity, the
  // joint-limit rows and the slot lists above need nothing from the slot records, so they ran while the other waves set
  // their slots up; the rendezvous that closes the set-up comes only now.
  if (share.late) share_fn();
  // ---------------------------------------------------------------- self-collision, part 2: the rows (behind (A3): the helper waves' shares of the filter are in)
  SelfRow sc[2];
  bool sc_wave = false;
  if (FEAT & 2) {
    const int gb = lane & ~(GRP - 1), lgi = lane & (GRP - 1);
    const int BIG = 0x7fffffff;
    if (scol.mask) scm |= scol.mask[lane] | scol.mask[64 + lane] | scol.mask[128 + lane];
    float p0 = P.contact_offset, p1 = P.contact_offset; int i0 = BIG, i1 = BIG;     // this lane's two deepest of the pairs the filter let through
    // (every lane walks its own set bits, lowest first: as many rounds as the busiest lane has candidates -- one or two --, not one per ordinal flagged
    //  anywhere in the wave; everything from the LDS table: a global load per round was most of this pass under flailing random actions)
    for (unsigned rest = scm; __ballot(rest != 0u) != 0ull; rest &= rest - 1u) {
      const bool mine = rest != 0u;
      const int kk = mine ? __ffs((int)rest) - 1 : 0, i = lgi + GRP * kk;
      const uint4 en = scol.tab[mine ? i : 0];
      const unsigned ia = en.x & 0xFFFFu, ib = en.x >> 16;                          // record index = slot * 64 + leg
      float ra, rb;
      const V3 ca = sc_sphere(cst, lm_, gb, ia & 63u, ia >> 6, pb, &ra), cb = sc_sphere(cst, lm_, gb, ib & 63u, ib >> 6, pb, &rb);
      const float ph = mine ? norm(ca - cb) - ra - rb : P.contact_offset;
      if (ph < p0) { p1 = p0; i1 = i0; p0 = ph; i0 = i; } else if (ph < p1) { p1 = ph; i1 = i; }
    }
    // the group's two deepest, lowest index first among equals (the order the oracle meets them in)
    const float m1 = grp_min_all(p0); const int w1 = grp_min_all(p0 == m1 ? i0 : BIG);
    const float c2 = i0 == w1 ? p1 : p0; const int ci2 = i0 == w1 ? i1 : i0;
    const float m2 = grp_min_all(c2); const int w2 = grp_min_all(c2 == m2 ? ci2 : BIG);
    const int win[2] = {w1, w2};
    sc_wave = __ballot(w1 != BIG) != 0ull;
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
      SelfRow& R = sc[q2];
      R.on = false; R.lam = 0.f; R.phi = 0.f; R.iA = 0.f; R.n = v3(0, 0, 1); R.slot_a = -1; R.slot_b = -1;
#pragma unroll
      for (int j = 0; j < 3; ++j) { R.f[j] = 0.f; R.Wk[j] = 0.f; }
#pragma unroll
      for (int a = 0; a < 6; ++a) R.Wb[a] = 0.f;
      if (__ballot(win[q2] != BIG) == 0ull) continue;                               // (wave-uniform)
      const bool on = win[q2] != BIG;
      const unsigned ix = scol.tab[on ? win[q2] : 0].x;
      const int la = ix & 63u, sa = (ix & 0xFFFFu) >> 6, lb = (ix >> 16) & 63u, sb = ix >> 22;
      float ra, rb;
      const V3 ca = sc_sphere(cst, lm_, gb, la, sa, pb, &ra), cb = sc_sphere(cst, lm_, gb, lb, sb, pb, &rb);
      const V3 d = ca - cb; const float dist = norm(d);
      const bool ok = on && dist > 1e-9f;
      const V3 n = sel3(ok, frcp(dist) * d, v3(0, 0, 1));
      const float phi = dist - ra - rb;
      const V3 pc = cb + (rb + 0.5f * phi) * n;
      const int link_a = __float_as_int(lm_.t[(LM_CP_LINK + sa) * GRP + la]), link_b = __float_as_int(lm_.t[(LM_CP_LINK + sb) * GRP + lb]);
      const int ka = link_a < 0 ? -1 : (link_a > 2 ? 2 : link_a), kb = link_b < 0 ? -1 : (link_b > 2 ? 2 : link_b);
      const bool mine_a = ok && lgi == la, mine_b = ok && lgi == lb;
      float f[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float jn = dot(n, cross(k.ax[j], pc - k.O[j]));
        f[j] = (mine_a && j <= ka ? jn : 0.f) - (mine_b && j <= kb ? jn : 0.f);
      }
      float z[3]; sym3_mul(Mi, f, z);
      float gvec[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) gvec[a] = -grp_sum(Mbk[a][0] * z[0] + Mbk[a][1] * z[1] + Mbk[a][2] * z[2]);
      symv6(Si, gvec, R.Wb);
      float A = 0.f;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float w = z[j];
#pragma unroll
        for (int a = 0; a < 6; ++a) w -= Y[j][a] * R.Wb[a];
        R.Wk[j] = w; R.f[j] = f[j]; A += f[j] * w;
      }
      A = grp_sum(A) + P.cfm;
      R.on = ok; R.phi = phi; R.n = n; R.iA = ok ? frcp(A) : 0.f;
      R.slot_a = mine_a ? (link_a < 0 ? 0 : (link_a > 3 ? 4 : link_a + 1)) : -1;
      R.slot_b = mine_b ? (link_b < 0 ? 0 : (link_b > 3 ? 4 : link_b + 1)) : -1;
    }
  }

  // the step's generalised displacement: dq = sum over the sub-intervals of h * v (TGS), dt * v of the last sweep (PGS)
  pk2 dqB[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}}; pk2 dqK01 = {0.f, 0.f}; float dqK2 = 0.f;
  const bool tgs = SPEC == 1 ? true : P.solver == LG_SOLVER_TGS;
  const int iters = P.iters > 0 ? P.iters : 1;
  const float h = tgs ? dt * frcp((float)iters) : dt, ih = frcp(h);
  const float tgsf = tgs ? 1.f : 0.f;
  const float vlim[3] = {lm_.f(LM_VEL_LIMIT), lm_.f(LM_VEL_LIMIT + 1), lm_.f(LM_VEL_LIMIT + 2)};
  if (slot_mask || jl_wave || ((FEAT & 2) && sc_wave)) {
    // packed state of the sweeps: base velocity in three pairs, joints (0, 1) as a pair and joint 2 alone, Y by joint pair
    pk2 vBp[3] = {{vB[0], vB[1]}, {vB[2], vB[3]}, {vB[4], vB[5]}};
    pk2 vK01 = {vK[0], vK[1]}; float vK2 = vK[2];
    pk2 Y01[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) { Y01[a].x = Y[0][a]; Y01[a].y = Y[1][a]; }
    const bool pyr = SPEC == 1 ? true : P.fric != LG_FRICTION_CONE;
    const float erp_ih = P.erp * ih;
    // A lane with nothing to relax at a step still reads a record (and multiplies it by zero impulses): it must be one the
    // set-up has written in THIS launch for every lane -- any slot of the wave's mask -- not a slot nobody uses, whose LDS
    // may hold another kernel's bits (0 x NaN would poison the base velocity of the quad).
    const int idle_sl = slot_mask ? __builtin_ctz(slot_mask) : 0;
    // One relaxation of a slot record held in registers.  The records of a lane's first PGS_REG active slots are fetched
    // once (15 ds_read_b128 each) in front of the sweeps and stay in VGPRs through all `iters` passes, multipliers included:
    // the lone main wave pays an LDS round trip per record per pass otherwise (the passes are a dependent chain, nothing
    // hides it).  Steps beyond PGS_REG fetch their record per pass.  Measured (A/B in one session, physics kernel at 4096 envs):
    // PGS_REG 1: -1.0 us; 2: +-0; 4: +2.5 us -- a wave has 256 architectural VGPRs, what lives beyond them sits in AGPRs and
    // costs a v_accvgpr_read per use, so only the first record (most legs have one contact, the foot) pays.
    constexpr int PGS_REG = LG_PGS_REG;
    float regs[PGS_REG > 0 ? PGS_REG : 1][CF_FIELDS];
#pragma unroll
    for (int st_ = 0; st_ < PGS_REG; ++st_) {
      if (st_ < my_steps) {                              // wave-uniform
        const int sl = st_ < my_count ? (int)((my_list >> (4 * st_)) & 0xfu) : idle_sl;
        load_slot_record(cst, sl, lane, regs[st_]);
      }
    }
    auto relax = [&](float* rec, const bool active) {
      const V3 n = v3(rec[CF_N], rec[CF_N + 1], rec[CF_N + 2]), r = v3(rec[CF_R], rec[CF_R + 1], rec[CF_R + 2]);
      const V3 jk0 = v3(rec[CF_JK0], rec[CF_JK0 + 1], rec[CF_JK0 + 2]), jk1 = v3(rec[CF_JK1], rec[CF_JK1 + 1], rec[CF_JK1 + 2]);
      const V3 jk2 = v3(rec[CF_JK2], rec[CF_JK2 + 1], rec[CF_JK2 + 2]);
      const float l0 = rec[CF_L0], iAnn = rec[CF_ANN];
      const pk2 l12 = {rec[CF_L1], rec[CF_L2]}, an12 = {rec[CF_AN12], rec[CF_AN12 + 1]};
      const pk2 b_r0 = {rec[CF_B], rec[CF_B + 1]}, b_r1 = {rec[CF_B + 2], rec[CF_B + 3]};
      // separation now: the gap of the start of the step plus the displacement of the contact point along the normal so far
      // (TGS; PGS keeps the gap) and the bias velocity over the (sub-)interval
      const V3 dp = v3(dqB[0].x, dqB[0].y, dqB[1].x) + cross(v3(dqB[1].y, dqB[2].x, dqB[2].y), r) + dqK01.x * jk0 + dqK01.y * jk1 + dqK2 * jk2;
      const float sep = fmaf(tgsf, dot(n, dp), rec[CF_GAP]);
      const float bn = sep >= 0.f ? -sep * ih : fminf(-sep * erp_ih, P.max_depen);
      V3 vp = v3(vBp[0].x, vBp[0].y, vBp[1].x) + cross(v3(vBp[1].y, vBp[2].x, vBp[2].y), r) + vK01.x * jk0 + vK01.y * jk1 + vK2 * jk2;
      const float u0 = dot(n, vp);
      pk2 u12 = pk_splat(vp.x) * (pk2){rec[CF_T12], rec[CF_T12 + 1]};
      u12 = pk_fma(pk_splat(vp.y), (pk2){rec[CF_T12 + 2], rec[CF_T12 + 3]}, u12);
      u12 = pk_fma(pk_splat(vp.z), (pk2){rec[CF_T12 + 4], rec[CF_T12 + 5]}, u12);
      const float ln = fmaxf(l0 - (u0 - bn) * iAnn, 0.f);
      const float dn = ln - l0;
      const pk2 w12 = pk_fma(an12, pk_splat(dn), u12);
      const float lim = mu * ln;
      // cone: exact 2x2 tangential block, projected on the disc
      pk2 c12 = {0.f, 0.f};
      if (SPEC != 1) {
        pk2 t12 = b_r0 * pk_splat(w12.x);
        t12 = pk_fma(b_r1, pk_splat(w12.y), t12);
        c12 = l12 - t12;
        const float m2 = c12.x * c12.x + c12.y * c12.y;
        if (m2 > lim * lim) { const float sc = m2 > 0.f ? lim * __builtin_amdgcn_rsqf(m2) : 0.f; c12 = c12 * pk_splat(sc); }
      }
      // pyramid: two scalar rows, each clamped on its own (b_r0 = (1/A11, A12), b_r1 = (A12, 1/A22))
      const float p1 = fminf(fmaxf(l12.x - w12.x * b_r0.x, -lim), lim);
      const float p2 = fminf(fmaxf(l12.y - fmaf(b_r0.y, p1 - l12.x, w12.y) * b_r1.y, -lim), lim);
      pk2 n12; n12.x = pyr ? p1 : c12.x; n12.y = pyr ? p2 : c12.y;
      const float d0 = active ? dn : 0.f;
      pk2 d12 = n12 - l12;
      d12.x = active ? d12.x : 0.f; d12.y = active ? d12.y : 0.f;
      rec[CF_L0] = active ? ln : l0; rec[CF_L1] = active ? n12.x : l12.x; rec[CF_L2] = active ? n12.y : l12.y;
      pk2 g[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        pk2 acc = pk_splat(d0) * (pk2){rec[CF_WB + 2 * p], rec[CF_WB + 2 * p + 1]};
        acc = pk_fma(pk_splat(d12.x), (pk2){rec[CF_WB + 6 + 2 * p], rec[CF_WB + 6 + 2 * p + 1]}, acc);
        acc = pk_fma(pk_splat(d12.y), (pk2){rec[CF_WB + 12 + 2 * p], rec[CF_WB + 12 + 2 * p + 1]}, acc);
        g[p].x = grp_sum(acc.x); g[p].y = grp_sum(acc.y);
      }
#pragma unroll
      for (int p = 0; p < 3; ++p) vBp[p] = vBp[p] + g[p];
      pk2 w01 = pk_splat(d0) * (pk2){rec[CF_ZCP], rec[CF_ZCP + 1]};
      w01 = pk_fma(pk_splat(d12.x), (pk2){rec[CF_ZCP + 2], rec[CF_ZCP + 3]}, w01);
      w01 = pk_fma(pk_splat(d12.y), (pk2){rec[CF_ZCP + 4], rec[CF_ZCP + 5]}, w01);
      float w2 = d0 * rec[CF_ZC2] + d12.x * rec[CF_ZC2 + 1] + d12.y * rec[CF_ZC2 + 2];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const float ga = (a & 1) ? g[a >> 1].y : g[a >> 1].x;
        w01 = pk_fma(-Y01[a], pk_splat(ga), w01);
        w2 -= Y[2][a] * ga;
      }
      vK01 = vK01 + w01; vK2 += w2;
    };
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
      // every lane walks the list of its own active slots; step j relaxes the j-th active contact of each of the four
      // legs together (Jacobi across the quad, Gauss-Seidel along the lists).  A pass takes max-over-lanes(list length)
      // steps -- measured 2.3 on the headline workload against 3.0 slots that are active somewhere in the wave.
#pragma unroll
      for (int st_ = 0; st_ < PGS_REG; ++st_)
        if (st_ < my_steps) relax(regs[st_], st_ < my_count);
#pragma unroll 1
      for (int step = PGS_REG; step < my_steps; ++step) {
        const bool active = step < my_count;
        const int sl = active ? (int)((my_list >> (4 * step)) & 0xfu) : idle_sl;   // idle lanes read a slot every lane has a record for and apply nothing
        float rec[CF_FIELDS];
        // (Measured and dropped, A/B in one session: reading the head of the record first and skipping the step when no lane of the wave
        //  can take an impulse -- multipliers zero, contact point separating at least as fast as its bias asks: exactly a no-op -- runs at
        //  0.0791 ms per step against 0.0783: such steps are too rare wave-wide to pay for the ballot and the branch.)
        load_slot_record(cst, sl, lane, rec);
        relax(rec, active);
        // multipliers back to the record, without a branch: an idle lane rewrites what it has just read
        *reinterpret_cast<float4*>(&CS(sl, CF_L0)) = make_float4(rec[CF_L0], rec[CF_ANN], rec[CF_L1], rec[CF_L2]);
      }
      if ((FEAT & 2) && sc_wave) {
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {        // the self-collision rows, one after the other
          SelfRow& R = sc[q2];
          const float u = grp_sum(R.f[0] * vK01.x + R.f[1] * vK01.y + R.f[2] * vK2);
          const float dsep = grp_sum(R.f[0] * dqK01.x + R.f[1] * dqK01.y + R.f[2] * dqK2);
          const float sep = fmaf(tgsf, dsep, R.phi);
          const float bn = sep >= 0.f ? -sep * ih : fminf(-sep * erp_ih, P.max_depen);
          const float ln = fmaxf(R.lam - (u - bn) * R.iA, 0.f);
          const float dl = R.on ? ln - R.lam : 0.f;
          R.lam = R.on ? ln : R.lam;
#pragma unroll
          for (int p = 0; p < 3; ++p) { vBp[p].x = fmaf(dl, R.Wb[2 * p], vBp[p].x); vBp[p].y = fmaf(dl, R.Wb[2 * p + 1], vBp[p].y); }
          vK01.x = fmaf(dl, R.Wk[0], vK01.x); vK01.y = fmaf(dl, R.Wk[1], vK01.y); vK2 = fmaf(dl, R.Wk[2], vK2);
        }
      }
      if (jl_wave) {
        float vBs[6] = {vBp[0].x, vBp[0].y, vBp[1].x, vBp[1].y, vBp[2].x, vBp[2].y}, vKs[3] = {vK01.x, vK01.y, vK2};
        const float dqs[3] = {dqK01.x, dqK01.y, dqK2};
#pragma unroll
        for (int j = 0; j < 3; ++j) {           // joint j of the four legs together, like a contact slot
          if (!jl_jw[j]) continue;
          const float u = jl_sgn[j] * vKs[j];
          const float gap = fmaf(tgsf * jl_sgn[j], dqs[j], jl_gap[j]);
          const float bn = gap >= 0.f ? -gap * ih : fminf(-gap * erp_ih, 10.f);
          const float ln = fmaxf(jl_lam[j] - (u - bn) * jl_iA[j], 0.f);
          const float dl = jl_act[j] ? ln - jl_lam[j] : 0.f;
          if (jl_act[j]) jl_lam[j] = ln;
          float gq[6];
#pragma unroll
          for (int a = 0; a < 6; ++a) gq[a] = grp_sum(dl * jl_Wb[j][a]);
#pragma unroll
          for (int a = 0; a < 6; ++a) vBs[a] += gq[a];
#pragma unroll
          for (int jj = 0; jj < 3; ++jj) {
            float w = dl * jl_y[j][jj];
#pragma unroll
            for (int a = 0; a < 6; ++a) w -= Y[jj][a] * gq[a];
            vKs[jj] += w;
          }
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) { vBp[p].x = vBs[2 * p]; vBp[p].y = vBs[2 * p + 1]; }
        vK01.x = vKs[0]; vK01.y = vKs[1]; vK2 = vKs[2];
      }
      // end of the sub-interval (TGS) / of the last sweep (PGS): joint speed limit (URDF <limit velocity>, a hard cap on |qd|),
      // then the displacement of the interval
      const bool close = tgs || it == iters - 1;
      {
        const float c0 = vlim[0] > 0.f ? fminf(fmaxf(vK01.x, -vlim[0]), vlim[0]) : vK01.x;
        const float c1 = vlim[1] > 0.f ? fminf(fmaxf(vK01.y, -vlim[1]), vlim[1]) : vK01.y;
        const float c2 = vlim[2] > 0.f ? fminf(fmaxf(vK2, -vlim[2]), vlim[2]) : vK2;
        vK01.x = close ? c0 : vK01.x; vK01.y = close ? c1 : vK01.y; vK2 = close ? c2 : vK2;
      }
      const pk2 hh = pk_splat(close ? h : 0.f);
#pragma unroll
      for (int p = 0; p < 3; ++p) dqB[p] = pk_fma(hh, vBp[p], dqB[p]);
      dqK01 = pk_fma(hh, vK01, dqK01); dqK2 = fmaf(hh.x, vK2, dqK2);
    }
#pragma unroll
    for (int st_ = 0; st_ < PGS_REG; ++st_) {
      if (st_ < my_steps && st_ < my_count) {            // (an idle lane has nothing to store: its record was only read)
        const int sl = (int)((my_list >> (4 * st_)) & 0xfu);
        *reinterpret_cast<float4*>(&CS(sl, CF_L0)) = make_float4(regs[st_][CF_L0], regs[st_][CF_ANN], regs[st_][CF_L1], regs[st_][CF_L2]);
      }
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) { vB[2 * p] = vBp[p].x; vB[2 * p + 1] = vBp[p].y; }
    vK[0] = vK01.x; vK[1] = vK01.y; vK[2] = vK2;
  } else {
    // no row anywhere in the wave: the velocity is the unconstrained one over the whole step
#pragma unroll
    for (int j = 0; j < 3; ++j) if (vlim[j] > 0.f) vK[j] = fminf(fmaxf(vK[j], -vlim[j]), vlim[j]);
#pragma unroll
    for (int p = 0; p < 3; ++p) { dqB[p].x = dt * vB[2 * p]; dqB[p].y = dt * vB[2 * p + 1]; }
    dqK01.x = dt * vK[0]; dqK01.y = dt * vK[1]; dqK2 = dt * vK[2];
  }

  STAMP(7);
  // ---------------------------------------------------------------- net contact force per body (world frame): the step's impulse / dt
  if (fbody) {
    V3 fb[5] = {v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0)};
    const float idt = idt_;
#pragma unroll 1
    for (int sl = 0; sl < LG_MAX_CP; ++sl) {
      if (!((slot_mask >> sl) & 1u)) continue;
      const bool on = ((AMASK(sl) >> lane) & 1ull) != 0ull;       // selected, not branched on: a lane without this contact adds zero
      const float4 n4 = CS4(sl, 0), lam = CS4(sl, 2), ta = CS4(sl, CF_T12 / 4), tb = CS4(sl, CF_T12 / 4 + 1);   // T12 = t1.x t2.x t1.y t2.y | t1.z t2.z ..
      const V3 t1 = v3(ta.x, ta.z, tb.x), t2 = v3(ta.y, ta.w, tb.y);
      V3 f = idt * (lam.x * v3(n4.x, n4.y, n4.z) + lam.z * t1 + lam.w * t2);
      f = v3(on ? f.x : 0.f, on ? f.y : 0.f, on ? f.z : 0.f);
      int link = lm_.i(LM_CP_LINK + sl);
      int slotb = link < 0 ? 0 : (link > 3 ? 4 : link + 1);
#pragma unroll
      for (int b = 0; b < 5; ++b) if (b == slotb) fb[b] = fb[b] + f;
    }
    if ((FEAT & 2) && sc_wave) {                // a self-contact loads both of its bodies: each side by the lane that owns it
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const V3 f = (idt * sc[q2].lam) * sc[q2].n;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          if (b == sc[q2].slot_a) fb[b] = fb[b] + f;
          if (b == sc[q2].slot_b) fb[b] = fb[b] - f;
        }
      }
    }
    fbody[0] = grp_sum(fb[0]);
#pragma unroll
    for (int b = 1; b < 5; ++b) fbody[b] = fb[b];
  }

  // ---------------------------------------------------------------- pose advance by the step's displacement (semi-implicit Euler)
  s.root[7] = vB[0]; s.root[8] = vB[1]; s.root[9] = vB[2]; s.root[10] = vB[3]; s.root[11] = vB[4]; s.root[12] = vB[5];
  s.root[0] += dqB[0].x; s.root[1] += dqB[0].y; s.root[2] += dqB[1].x;
  {
    V3 w = v3(dqB[1].y, dqB[2].x, dqB[2].y); float ang = norm(w);
    float sh, ch; sincos_fast(0.5f * ang, &sh, &ch);
    sh = ang > 1e-12f ? sh * frcp(ang) : 0.5f;
    float dq0 = sh * w.x, dq1 = sh * w.y, dq2 = sh * w.z, dq3 = ch;
    float* qq = s.root + 3;
    float x = dq3 * qq[0] + dq0 * qq[3] + dq1 * qq[2] - dq2 * qq[1];
    float y = dq3 * qq[1] - dq0 * qq[2] + dq1 * qq[3] + dq2 * qq[0];
    float z = dq3 * qq[2] + dq0 * qq[1] - dq1 * qq[0] + dq2 * qq[3];
    float w4 = dq3 * qq[3] - dq0 * qq[0] - dq1 * qq[1] - dq2 * qq[2];
    float inv = __builtin_amdgcn_rsqf(x * x + y * y + z * z + w4 * w4);
    qq[0] = x * inv; qq[1] = y * inv; qq[2] = z * inv; qq[3] = w4 * inv;
  }
  s.qd[0] = vK[0]; s.qd[1] = vK[1]; s.qd[2] = vK[2];
  s.q[0] += dqK01.x; s.q[1] += dqK01.y; s.q[2] += dqK2;
  STAMP(8);
}
#endif           // NJ == 3
