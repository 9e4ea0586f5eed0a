// lg_physics.h — articulated-body dynamics + contact for one environment, spread over a DPP quad (4 lanes):
// lane l owns leg l (3 revolute joints, its links, its collision spheres); the floating base is replicated on all
// four lanes and every cross-leg coupling goes through the 6x6 base Schur complement, reduced with quad DPP adds.
//
//   M = [ Mbb  Mb1 Mb2 Mb3 Mb4 ]      legs couple only through the base, so with  Y_k = Mkk^-1 Mbk^T
//       [ Mb1' M11             ]        S   = Mbb - sum_k Mbk Y_k                      (6x6, replicated)
//       [ ...        ...       ]        a_b = S^-1 (f_b - sum_k Mbk Mkk^-1 f_k)
//       [ Mb4'             M44 ]        a_k = Mkk^-1 f_k - Y_k a_b
//
// Replaces gym.simulate (reference legged_robot.py:100; PhysX, closed): same tensor contract (:564-584), own model:
// composite-rigid-body mass matrix + recursive Newton-Euler bias (spatial quantities about the base origin P, world
// axes), sphere-vs-heightfield contacts solved at velocity level by projected Gauss-Seidel (the four lanes' contacts of
// one slot relax simultaneously, slots sequentially), Coulomb friction disc, semi-implicit Euler.
#pragma once
#include "lg_device.h"

struct LegKin {
  M3 R[3];
  V3 O[3], ax[3], com[3], w[3], vO[3];
  S3 Ic[3];
};

LG_DEV void leg_kinematics(const lg_robot_model* __restrict__ m, int l, const M3& Rb, V3 pb, V3 vb, V3 wb,
                           const float q[3], const float qd[3], LegKin& k) {
  M3 Rp = Rb; V3 Op = pb, wp = wb, vp = vb;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    M3 fix = ldm3(m->joint_rot[l][j]);
    V3 a = ld3(m->joint_axis[l][j]);
    k.O[j] = Op + mul(Rp, ld3(m->joint_pos[l][j]));
    M3 R0 = mul(Rp, fix);
    k.ax[j] = mul(R0, a);
    k.R[j] = mul(R0, axis_angle(a, q[j]));
    k.com[j] = k.O[j] + mul(k.R[j], ld3(m->link_com[l][j]));
    k.vO[j] = vp + cross(wp, k.O[j] - Op);
    k.w[j] = wp + qd[j] * k.ax[j];
    k.Ic[j] = rotate_inertia(k.R[j], m->link_inertia[l][j]);
    Rp = k.R[j]; Op = k.O[j]; wp = k.w[j]; vp = k.vO[j];
  }
}

#define LT(i, j) ((i) * ((i) + 1) / 2 + (j))   // packed lower-triangular index

LG_DEV void sym3_inverse(const float a[6], float o[6]) {  // a: 00 01 02 11 12 22
  float c00 = a[3] * a[5] - a[4] * a[4], c01 = a[2] * a[4] - a[1] * a[5], c02 = a[1] * a[4] - a[2] * a[3];
  float id = 1.0f / (a[0] * c00 + a[1] * c01 + a[2] * c02);
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
    float inv = 1.0f / sqrtf(fmaxf(d, 1e-20f));
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

// terrain surface under (x, y): height and unit normal of the regular-grid triangulation (diagonal v(i,j)->v(i+1,j+1))
struct TerrainView { int mesh_type, rows, cols; float hscale, vscale, border; const int16_t* __restrict__ H; };
LG_DEV void terrain_query(const TerrainView& T, float x, float y, float* h, V3* n) {
  if (T.mesh_type == LG_MESH_PLANE) { *h = 0.f; *n = v3(0, 0, 1); return; }
  float fx = (x + T.border) / T.hscale, fy = (y + T.border) / T.hscale;
  int i = (int)floorf(fx), j = (int)floorf(fy);
  i = max(0, min(i, T.rows - 2)); j = max(0, min(j, T.cols - 2));
  float u = fminf(fmaxf(fx - (float)i, 0.f), 1.f), v = fminf(fmaxf(fy - (float)j, 0.f), 1.f);
  const int16_t* r0 = T.H + (size_t)i * T.cols + j;
  float h0 = T.vscale * r0[0], h1 = T.vscale * r0[1], h2 = T.vscale * r0[T.cols], h3 = T.vscale * r0[T.cols + 1];
  float dhdu, dhdv;
  if (v >= u) { dhdu = h3 - h1; dhdv = h1 - h0; } else { dhdu = h2 - h0; dhdv = h3 - h2; }
  *h = h0 + u * dhdu + v * dhdv;
  V3 g = v3(-dhdu / T.hscale, -dhdv / T.hscale, 1.f);
  *n = (1.f / norm(g)) * g;
}

// per-contact-slot scratch in LDS, laid out [slot][field][lane] (lane-contiguous: conflict-free ds_read_b32)
enum { CF_N = 0, CF_T1 = 3, CF_T2 = 6, CF_R = 9, CF_JK0 = 12, CF_JK1 = 15, CF_JK2 = 18,
       CF_ANN = 21, CF_AN1, CF_AN2, CF_A11, CF_A12, CF_A22, CF_BN, CF_L0, CF_L1, CF_L2, CF_ACTIVE, CF_FIELDS = 32 };
#define CS(slot, f) cst[((slot) * CF_FIELDS + (f)) * 64 + lane]
LG_DEV V3 lds3(const float* cst, int slot, int f, int lane) { return v3(CS(slot, f), CS(slot, f + 1), CS(slot, f + 2)); }
LG_DEV void sts3(float* cst, int slot, int f, int lane, V3 a) { CS(slot, f) = a.x; CS(slot, f + 1) = a.y; CS(slot, f + 2) = a.z; }

struct PhysParams {
  float dt; V3 grav; int iters; float contact_offset, max_depen, erp, cfm, terrain_mu;
};

struct QuadState {           // per lane: replicated base + own leg
  float root[13];            // pos3, quat xyzw, lin vel3, ang vel3 (world)
  float q[3], qd[3];
};

// One physics step of length P.dt for the env this quad owns.  tau[3] = this leg's joint torques.
// fbody[5] (optional) receives the net contact force on {base (already quad-summed), link0, link1, link2, foot}.
LG_DEV void physics_substep(const lg_robot_model* __restrict__ m, const TerrainView& T, const PhysParams& P, int l, int lane,
                            float* cst, QuadState& s, const float tau[3], float mu_robot, float madd, V3* fbody) {
  const float dt = P.dt;
  const V3 pb = v3(s.root[0], s.root[1], s.root[2]);
  const V3 vb = v3(s.root[7], s.root[8], s.root[9]), wb = v3(s.root[10], s.root[11], s.root[12]);
  const M3 Rb = quat_to_mat(s.root + 3);
  LegKin k;
  leg_kinematics(m, l, Rb, pb, vb, wb, s.q, s.qd, k);

  // ---------------------------------------------------------------- bias forces (RNEA, zero generalised acceleration)
  const float m0 = m->base_mass + madd, iscale = m0 / m->base_mass;
  const V3 rc0 = mul(Rb, ld3(m->base_com));
  S3 I0 = rotate_inertia(Rb, m->base_inertia);
  I0.xx *= iscale; I0.xy *= iscale; I0.xz *= iscale; I0.yy *= iscale; I0.yz *= iscale; I0.zz *= iscale;
  float lm[3] = {m->link_mass[l][0], m->link_mass[l][1], m->link_mass[l][2]};
  float bk[3]; V3 Fs = v3(0, 0, 0), Ns = v3(0, 0, 0);
  {
    V3 wp = wb, alp = v3(0, 0, 0), aOp = v3(0, 0, 0), Op = pb;
    V3 F[3], NP[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      V3 d = k.O[j] - Op;
      V3 aO = aOp + cross(alp, d) + cross(wp, cross(wp, d));
      V3 al = alp + s.qd[j] * cross(wp, k.ax[j]);
      V3 w = k.w[j];
      V3 rc = k.com[j] - k.O[j];
      V3 ac = aO + cross(al, rc) + cross(w, cross(w, rc));
      F[j] = lm[j] * (ac - P.grav);
      NP[j] = mul(k.Ic[j], al) + cross(w, mul(k.Ic[j], w)) + cross(k.com[j] - pb, F[j]);
      wp = w; alp = al; aOp = aO; Op = k.O[j];
    }
#pragma unroll
    for (int j = 2; j >= 0; --j) {
      Fs = Fs + F[j]; Ns = Ns + NP[j];
      bk[j] = dot(k.ax[j], Ns - cross(k.O[j] - pb, Fs));
    }
  }
  float bb[6];
  {
    V3 ac = cross(wb, cross(wb, rc0));
    V3 Fb = m0 * (ac - P.grav);
    V3 Nb = cross(wb, mul(I0, wb)) + cross(rc0, Fb);
    V3 Ft = Fb + quad_sum(Fs), Nt = Nb + quad_sum(Ns);
    bb[0] = Ft.x; bb[1] = Ft.y; bb[2] = Ft.z; bb[3] = Nt.x; bb[4] = Nt.y; bb[5] = Nt.z;
  }

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
    const float mt = m0 + quad_sum(mc);
    const V3 ht = m0 * rc0 + quad_sum(hc);
    const S3 It = inertia_about(I0, m0, rc0) + quad_sum(Icp);
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) {
        float sk = Mbk[a][0] * Y[0][b] + Mbk[a][1] * Y[1][b] + Mbk[a][2] * Y[2][b];
        L[LT(a, b)] = -quad_sum(sk);
      }
    L[LT(0, 0)] += mt; L[LT(1, 1)] += mt; L[LT(2, 2)] += mt;
    // rows 3..5, cols 0..2 : [h]x
    L[LT(3, 1)] += -ht.z; L[LT(3, 2)] += ht.y;
    L[LT(4, 0)] += ht.z;  L[LT(4, 2)] += -ht.x;
    L[LT(5, 0)] += -ht.y; L[LT(5, 1)] += ht.x;
    L[LT(3, 3)] += It.xx; L[LT(4, 3)] += It.xy; L[LT(5, 3)] += It.xz;
    L[LT(4, 4)] += It.yy; L[LT(5, 4)] += It.yz; L[LT(5, 5)] += It.zz;
    chol6(L);
  }

  // ---------------------------------------------------------------- unconstrained velocity v* = v + dt M^-1 (tau - c)
  float vB[6] = {vb.x, vb.y, vb.z, wb.x, wb.y, wb.z};
  float vK[3] = {s.qd[0], s.qd[1], s.qd[2]};
  {
    float rk[3] = {tau[0] - bk[0], tau[1] - bk[1], tau[2] - bk[2]}, y[3];
    sym3_mul(Mi, rk, y);
    float g[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) g[a] = -bb[a] - quad_sum(Mbk[a][0] * y[0] + Mbk[a][1] * y[1] + Mbk[a][2] * y[2]);
    solve6(L, g);
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

  // ---------------------------------------------------------------- contact detection + per-contact setup
  const float mu = 0.5f * (mu_robot + P.terrain_mu);   // PhysX default friction combine mode: average
  const int ncp = m->cp_count[l];
  unsigned slot_mask = 0;    // wave-uniform: slots with at least one active contact in this wave
#pragma unroll 1
  for (int sl = 0; sl < LG_MAX_CP; ++sl) {
    bool active = false;
    V3 n = v3(0, 0, 1), x = pb; float phi = 0.f, rad = 0.f; int lk = -1;
    if (sl < ncp) {
      int link = m->cp_link[l][sl];
      lk = link < 0 ? -1 : (link > 2 ? 2 : link);
      V3 lp = ld3(m->cp_pos[l][sl]);
      rad = m->cp_radius[l][sl];
      if (lk < 0) x = pb + mul(Rb, lp);
      else if (lk == 0) x = k.O[0] + mul(k.R[0], lp);
      else if (lk == 1) x = k.O[1] + mul(k.R[1], lp);
      else x = k.O[2] + mul(k.R[2], lp);
      float h; terrain_query(T, x.x, x.y, &h, &n);
      phi = (x.z - h) * n.z - rad;
      active = phi < P.contact_offset;
    }
    CS(sl, CF_ACTIVE) = active ? 1.f : 0.f;
    CS(sl, CF_L0) = 0.f; CS(sl, CF_L1) = 0.f; CS(sl, CF_L2) = 0.f;
    if (__ballot(active) == 0ull) continue;
    slot_mask |= 1u << sl;
    // contact frame and Jacobian pieces (computed on every lane of the wave; inactive lanes carry harmless values)
    V3 p = x - rad * n, r = p - pb;
    V3 a0 = fabsf(n.x) < 0.57735f ? v3(1, 0, 0) : v3(0, 1, 0);
    V3 t1 = cross(a0, n); t1 = (1.f / norm(t1)) * t1;
    V3 t2 = cross(n, t1);
    V3 jk[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) jk[j] = j <= lk ? cross(k.ax[j], p - k.O[j]) : v3(0, 0, 0);
    float bn = phi >= 0.f ? -phi / dt : fminf(-phi * P.erp / dt, P.max_depen);
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
      solve6(L, g);
#pragma unroll
      for (int a = 0; a < 6; ++a) Wb[c][a] = g[a];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float w = z[j];
#pragma unroll
        for (int a = 0; a < 6; ++a) w -= Y[j][a] * g[a];
        Wk[c][j] = w;
      }
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
    sts3(cst, sl, CF_N, lane, n); sts3(cst, sl, CF_T1, lane, t1); sts3(cst, sl, CF_T2, lane, t2); sts3(cst, sl, CF_R, lane, r);
    sts3(cst, sl, CF_JK0, lane, jk[0]); sts3(cst, sl, CF_JK1, lane, jk[1]); sts3(cst, sl, CF_JK2, lane, jk[2]);
    CS(sl, CF_ANN) = A[0][0] + P.cfm; CS(sl, CF_AN1) = A[1][0]; CS(sl, CF_AN2) = A[2][0];
    CS(sl, CF_A11) = A[1][1] + P.cfm; CS(sl, CF_A12) = A[1][2]; CS(sl, CF_A22) = A[2][2] + P.cfm;
    CS(sl, CF_BN) = bn;
  }

  // ---------------------------------------------------------------- projected Gauss-Seidel
  if (slot_mask) {
#pragma unroll 1
    for (int it = 0; it < P.iters; ++it) {
#pragma unroll 1
      for (int sl = 0; sl < LG_MAX_CP; ++sl) {
        if (!((slot_mask >> sl) & 1u)) continue;
        const bool active = CS(sl, CF_ACTIVE) != 0.f;
        V3 n = lds3(cst, sl, CF_N, lane), t1 = lds3(cst, sl, CF_T1, lane), t2 = lds3(cst, sl, CF_T2, lane);
        V3 r = lds3(cst, sl, CF_R, lane);
        V3 jk0 = lds3(cst, sl, CF_JK0, lane), jk1 = lds3(cst, sl, CF_JK1, lane), jk2 = lds3(cst, sl, CF_JK2, lane);
        // velocity of the contact point
        V3 vp = v3(vB[0], vB[1], vB[2]) + cross(v3(vB[3], vB[4], vB[5]), r) + vK[0] * jk0 + vK[1] * jk1 + vK[2] * jk2;
        float u0 = dot(n, vp), u1 = dot(t1, vp), u2 = dot(t2, vp);
        float l0 = CS(sl, CF_L0), l1 = CS(sl, CF_L1), l2 = CS(sl, CF_L2);
        float Ann = CS(sl, CF_ANN), An1 = CS(sl, CF_AN1), An2 = CS(sl, CF_AN2);
        float A11 = CS(sl, CF_A11), A12 = CS(sl, CF_A12), A22 = CS(sl, CF_A22);
        float ln = fmaxf(l0 - (u0 - CS(sl, CF_BN)) / Ann, 0.f);
        float dn = ln - l0;
        float w1 = u1 + An1 * dn, w2 = u2 + An2 * dn;
        float det = A11 * A22 - A12 * A12;
        float n1 = l1 - (A22 * w1 - A12 * w2) / det;
        float n2 = l2 - (-A12 * w1 + A11 * w2) / det;
        float lim = mu * ln, mag = sqrtf(n1 * n1 + n2 * n2);
        if (mag > lim) { float sc = mag > 0.f ? lim / mag : 0.f; n1 *= sc; n2 *= sc; }
        float d0 = active ? dn : 0.f, d1 = active ? n1 - l1 : 0.f, d2 = active ? n2 - l2 : 0.f;
        if (active) { CS(sl, CF_L0) = ln; CS(sl, CF_L1) = n1; CS(sl, CF_L2) = n2; }
        // apply: world impulse f at the contact point
        V3 f = d0 * n + d1 * t1 + d2 * t2;
        float jkf[3] = {dot(jk0, f), dot(jk1, f), dot(jk2, f)}, z[3];
        sym3_mul(Mi, jkf, z);
        V3 rf = cross(r, f);
        float g[6] = {f.x, f.y, f.z, rf.x, rf.y, rf.z};
#pragma unroll
        for (int a = 0; a < 6; ++a) g[a] = quad_sum(g[a] - (Mbk[a][0] * z[0] + Mbk[a][1] * z[1] + Mbk[a][2] * z[2]));
        solve6(L, g);
#pragma unroll
        for (int a = 0; a < 6; ++a) vB[a] += g[a];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          float w = z[j];
#pragma unroll
          for (int a = 0; a < 6; ++a) w -= Y[j][a] * g[a];
          vK[j] += w;
        }
      }
    }
  }

  // ---------------------------------------------------------------- joint speed limit (URDF <limit velocity>)
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float lim = m->dof_vel_limit[3 * l + j];
    if (lim > 0.f) vK[j] = fminf(fmaxf(vK[j], -lim), lim);
  }

  // ---------------------------------------------------------------- net contact force per body (world frame)
  if (fbody) {
    V3 fb[5] = {v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0)};
    const float idt = 1.f / dt;
#pragma unroll 1
    for (int sl = 0; sl < LG_MAX_CP; ++sl) {
      if (!((slot_mask >> sl) & 1u)) continue;
      if (CS(sl, CF_ACTIVE) == 0.f) continue;
      V3 f = idt * (CS(sl, CF_L0) * lds3(cst, sl, CF_N, lane) + CS(sl, CF_L1) * lds3(cst, sl, CF_T1, lane) + CS(sl, CF_L2) * lds3(cst, sl, CF_T2, lane));
      int link = m->cp_link[l][sl];
      int slotb = link < 0 ? 0 : (link > 3 ? 4 : link + 1);
#pragma unroll
      for (int b = 0; b < 5; ++b) if (b == slotb) fb[b] = fb[b] + f;
    }
    fbody[0] = quad_sum(fb[0]);
#pragma unroll
    for (int b = 1; b < 5; ++b) fbody[b] = fb[b];
  }

  // ---------------------------------------------------------------- semi-implicit Euler
  s.root[7] = vB[0]; s.root[8] = vB[1]; s.root[9] = vB[2]; s.root[10] = vB[3]; s.root[11] = vB[4]; s.root[12] = vB[5];
  s.root[0] += dt * vB[0]; s.root[1] += dt * vB[1]; s.root[2] += dt * vB[2];
  {
    V3 w = v3(vB[3], vB[4], vB[5]); float wn = norm(w), ang = wn * dt;
    float sh, ch; sincosf(0.5f * ang, &sh, &ch);
    sh = wn > 1e-9f ? sh / wn : 0.5f * dt;
    float dq0 = sh * w.x, dq1 = sh * w.y, dq2 = sh * w.z, dq3 = ch;
    float* qq = s.root + 3;
    float x = dq3 * qq[0] + dq0 * qq[3] + dq1 * qq[2] - dq2 * qq[1];
    float y = dq3 * qq[1] - dq0 * qq[2] + dq1 * qq[3] + dq2 * qq[0];
    float z = dq3 * qq[2] + dq0 * qq[1] - dq1 * qq[0] + dq2 * qq[3];
    float w4 = dq3 * qq[3] - dq0 * qq[0] - dq1 * qq[1] - dq2 * qq[2];
    float inv = 1.f / sqrtf(x * x + y * y + z * z + w4 * w4);
    qq[0] = x * inv; qq[1] = y * inv; qq[2] = z * inv; qq[3] = w4 * inv;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) { s.qd[j] = vK[j]; s.q[j] += dt * vK[j]; }
}
