// lg_chain.h — the physics of the instances whose legs are chains of NJ != 3 joints: today the two-legged, six-joints-per-leg one (Cassie:
// reference envs/cassie/cassie.py, cassie_config.py, resources/robots/cassie/urdf/cassie.urdf:315-416 -- an OPEN chain: the knee-spring joints that would
// close a loop are commented out in the reference's file).  Included by lg_step.hip inside the instance's namespace, behind lg_physics.h.
//
// Same model and the same solver as the tuned three-joint kernels of lg_physics.h -- floating base replicated on the lanes of an env's group, one leg per
// lane, legs coupled through the 6x6 base Schur complement reduced with DPP adds; composite-rigid-body mass matrix + recursive Newton-Euler bias; sphere
// (+ capsule-segment edge) contacts against plane / height grid / grid mesh; TGS or PGS over contact, self-collision and joint-limit rows with pyramid or
// cone friction -- written once over NJ in plain loops: a leg's joint block Mkk is NJ x NJ (packed, inverted through its Cholesky factor), every
// "three" of lg_physics.h is a loop bound.  One wave per workgroup, no helper waves, no fused tail: this instance is about coverage, its
// step is physics_kernel_chain + post_kernel.  The oracle (run-time joint count) is the checker, as for the others.
#pragma once
static_assert(NJ == 6, "lg_chain.h inverts the leg block with the 6 x 6 routines of lg_physics.h");

#define CH_NCP 4                       // contact slots per leg of this instance (Cassie: two toe spheres + the pelvis sphere on one leg); LDS: 4 x 64 x 84 floats
// slot record in LDS, [slot][lane][field]
enum { CH_N = 0, CH_GAP = 3, CH_R = 4, CH_ACTIVE = 7, CH_L0 = 8, CH_IANN = 9, CH_L1 = 10, CH_L2 = 11, CH_T1 = 12, CH_T2 = 15, CH_AN = 18 /* An1 An2 */, CH_B = 20 /* 4 */,
       CH_WB = 24 /* 3 x 6 */, CH_JK = 42 /* NJ x 3 */, CH_ZC = CH_JK + 3 * NJ /* 3 x NJ: (Mkk^-1 J^T)[c][j] */, CH_USED = CH_ZC + 3 * NJ, CH_FIELDS = (CH_USED + 3) / 4 * 4 + ((((CH_USED + 3) / 4) % 2) ? 0 : 4) };
#define CH_CST_FLOATS (CH_NCP * CH_FIELDS * 64)
#define CHS(slot, f) cst[((slot) * 64 + lane) * CH_FIELDS + (f)]
#define PKI(a, b) ((a) >= (b) ? LT(a, b) : LT(b, a))     // packed-lower index of a symmetric matrix entry

// frame of link `link` of this lane's leg (-1: the base; >= NJ: the last link, which carries the foot body)
LG_DEV void ch_link_frame(const LegKin& k, const M3& Rb, V3 pb, int link, M3* R, V3* O) {
  M3 Rr = Rb; V3 Or = pb;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const bool me = link == j || (j == NJ - 1 && link > j);
#pragma unroll
    for (int i = 0; i < 9; ++i) Rr.m[i] = me ? k.R[j].m[i] : Rr.m[i];
    Or = sel3(me, k.O[j], Or);
  }
  *R = Rr; *O = Or;
}

// torques of this lane's NJ joints (LR:425-448): PD / velocity / torque mode (no actuator network on this instance)
LG_DEV void ch_leg_torques(const lg_config& g, const LegModel& lm_, const float act[NJ], const float q[NJ], const float qd[NJ], const float last_qd[NJ], float tau[NJ]) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const float a = act[j] * g.action_scale, kp = lm_.f(LM_PGAIN + j), kd = lm_.f(LM_DGAIN + j);
    float t;
    if (g.control_type == LG_CTRL_P) t = kp * (a + lm_.f(LM_DEFAULT_POS + j) - q[j]) - kd * qd[j];
    else if (g.control_type == LG_CTRL_V) t = kp * (a - qd[j]) - kd * (qd[j] - last_qd[j]) / g.sim_dt;
    else t = a;
    const float lim = lm_.f(LM_TORQUE_LIMIT + j);
    tau[j] = fminf(fmaxf(t, -lim), lim);
  }
}

LG_DEV void ch_wave_sync() {          // LDS writes of one half of the wave visible to the other (one wave's LDS operations complete in order)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// contact detection of slot sl: plane / height grid (sphere + the capsule segment's edge candidates) or grid mesh (closest point + the segment against the mesh's own edges)
template <bool TMESH>
LG_DEV void ch_detect_slot(int sl, const LegModel& lm_, const TerrainView& T, const PhysParams& P, const LegKin& k, const M3& Rb, V3 pb, float* cst, int lane) {
  const int ncp = lm_.i(LM_CP_COUNT);
  const int link = lm_.i(LM_CP_LINK + sl);
  M3 Rl; V3 Ol; ch_link_frame(k, Rb, pb, link, &Rl, &Ol);
  const float rad = lm_.f(LM_CP_RADIUS + sl);
  const V3 x0 = Ol + mul(Rl, lm_.v(LM_CP_POS + 3 * sl));
  const V3 gv = mul(Rl, lm_.v(LM_CP_SLIDE + 3 * sl));
  const bool seg = gv.x != 0.f || gv.y != 0.f || gv.z != 0.f;
  V3 x = x0, n = v3(0, 0, 1); float phi = 1.f; bool hit = true;
  if (TMESH) {                                         // (grid meshes only: validate())
    ClosestQuery Q;
    Q.p = x; Q.max_dist = rad + P.contact_offset + LG_MESH_CONTACT_MARGIN; Q.on = sl < ncp; Q.found = false; Q.cp = x; Q.fn = v3(0, 0, 1);
    Q.range = rad + P.contact_offset; Q.lb = Q.max_dist;
    closest_point_grid(T, Q);
    hit = false;
    if (Q.on && Q.found) {
      const V3 diff = x - Q.cp; const float dist = norm(diff);
      if (dist <= rad + P.contact_offset + LG_MESH_CONTACT_MARGIN) {
        const float sign = dot(diff, Q.fn) < 0.f ? -1.f : 1.f;
        n = dist > 1e-6f ? (sign / dist) * diff : Q.fn;
        phi = sign * dist - rad; hit = true;
      }
    }
    if (T.SEG4 && __any(seg && sl < ncp)) {            // capsule segments against the grid mesh's own edges (round 6: contact_detect_mesh<true> of lg_physics.h, the same rule)
      typedef float f4v __attribute__((ext_vector_type(4))); typedef const f4v __attribute__((address_space(1)))* gf4;
      const gf4 SG = (gf4)T.SEG4;
      const float ihs = frcp(T.hscale);
      const float fx0 = (x0.x + T.border) * ihs, fy0 = (x0.y + T.border) * ihs, dfx = gv.x * ihs, dfy = gv.y * ihs;
      const float zlow = fminf(x0.z, x0.z + gv.z) - rad - P.contact_offset;
#pragma unroll
      for (int ax = 0; ax < 2; ++ax) {
        const EdgePiece pc = ax == 0 ? caps_edge_piece(fx0, dfx, fy0, dfy, T.rows, T.cols) : caps_edge_piece(fy0, dfy, fx0, dfx, T.cols, T.rows);
        const gf4 ep = ax == 0 ? SG + (size_t)pc.L * T.cols + pc.j : SG + (size_t)pc.j * T.cols + pc.L;
        const f4v e0 = ep[0], e1 = ep[ax == 0 ? 1 : T.cols];
        const V3 E0 = v3(e0.x, e0.y, e0.z), d2 = v3(e1.x - e0.x, e1.y - e0.y, e1.z - e0.z);
        const bool cand = seg && sl < ncp && pc.on && zlow < fmaxf(e0.z, e1.z) && d2.x * d2.x + d2.y * d2.y > 0.25f * T.hscale * T.hscale;
        if (!__any(cand)) continue;
        V3 A, E; seg_seg_closest(x0, gv, E0, cand ? d2 : v3(0.f, T.hscale, 0.f), &A, &E);
        const V3 d = A - E; const float dist = norm(d);
        const float sg = d.z >= 0.f ? 1.f : -1.f;
        const V3 ne = dist > 1e-9f ? (sg * frcp(dist)) * d : v3(0, 0, 1);
        const float pe = sg * dist - rad;
        const bool better = cand && pe < phi - 1e-5f && (sg > 0.f || dist <= rad);
        phi = better ? pe : phi; n = sel3(better, ne, n); x = sel3(better, A, x); hit = hit || better;
      }
    }
  } else {
    float hh; terrain_query(T, x.x, x.y, &hh, &n);
    phi = (x.z - hh) * n.z - rad;
    if (T.mesh_type == LG_MESH_HEIGHTFIELD && __any(seg)) {
      const float ihs = frcp(T.hscale);
      const float fx0 = (x0.x + T.border) * ihs, fy0 = (x0.y + T.border) * ihs, dfx = gv.x * ihs, dfy = gv.y * ihs;
      const float zlow = fminf(x0.z, x0.z + gv.z) - rad - P.contact_offset;
#pragma unroll
      for (int ax = 0; ax < 2; ++ax) {
        const EdgePiece pc = ax == 0 ? caps_edge_piece(fx0, dfx, fy0, dfy, T.rows, T.cols) : caps_edge_piece(fy0, dfy, fx0, dfx, T.cols, T.rows);
        const int16_t* hp = ax == 0 ? T.H + (size_t)pc.L * T.cols + pc.j : T.H + (size_t)pc.j * T.cols + pc.L;
        const float h0 = T.vscale * (float)hp[0], h1 = T.vscale * (float)hp[ax == 0 ? 1 : T.cols];
        const bool cand = seg && pc.on && zlow < fmaxf(h0, h1);
        const float cl = (float)pc.L * T.hscale - T.border, cj = (float)pc.j * T.hscale - T.border;
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
  }
  const bool active = hit && (sl < ncp) && (phi < P.contact_offset);
  const V3 r = (x - rad * n) - pb;
  CHS(sl, CH_N) = n.x; CHS(sl, CH_N + 1) = n.y; CHS(sl, CH_N + 2) = n.z; CHS(sl, CH_GAP) = phi;
  CHS(sl, CH_R) = r.x; CHS(sl, CH_R + 1) = r.y; CHS(sl, CH_R + 2) = r.z; CHS(sl, CH_ACTIVE) = active ? 1.f : 0.f;
  CHS(sl, CH_L0) = 0.f; CHS(sl, CH_L1) = 0.f; CHS(sl, CH_L2) = 0.f; CHS(sl, CH_IANN) = 0.f;
}

// sphere centre of (leg, slot) of this lane's env, from the slot records (the self-collision pass)
LG_DEV V3 ch_sc_sphere(const float* cst, const LegModel& lm_, int gb, int leg, int slot, V3 pb, float* rad) {
  const float* rec = cst + ((slot) * 64 + gb + leg) * CH_FIELDS;
  const float r = lm_.t[(LM_CP_RADIUS + slot) * GRP + leg];
  *rad = r;
  return v3(pb.x + rec[CH_R] + r * rec[CH_N], pb.y + rec[CH_R + 1] + r * rec[CH_N + 1], pb.z + rec[CH_R + 2] + r * rec[CH_N + 2]);
}
struct ChSelfRow { bool on; float phi, iA, lam; V3 n; float f[NJ], Wb[6], Wk[NJ]; int slot_a, slot_b; };

// One physics step of length P.dt for the env this lane group owns (see physics_substep of lg_physics.h: the same steps in the same order).
// fbody[NJ + 2] (optional): net contact force on {base (group-summed), link 0 .. NJ-1, foot body}.
// EXT_DETECT: helper waves of the workgroup detect the contact slots while this wave builds the bias and the mass matrix (physics_kernel_chain<.., HELP = true>);
// the rendezvous with them stands in front of the first read of a slot record.
template <bool TMESH, bool EXT_DETECT = false>
LG_DEV void chain_substep(const lg_robot_model* __restrict__ m, const LegModel& lm_, const TerrainView& T, const PhysParams& P, int lane, float* cst, QuadState& s,
                          const float tau[NJ], float mu_robot, float madd, V3* fbody, SelfCol scol, int half = -1) {
  // half >= 0: this lane and lane + 32 of the wave are mirrors (same env and leg, `lane` = their common row): slot `sl` is set up by the half sl & 1
  const float dt = P.dt;
  const V3 pb = v3(s.root[0], s.root[1], s.root[2]);
  const V3 vb = v3(s.root[7], s.root[8], s.root[9]), wb = v3(s.root[10], s.root[11], s.root[12]);
  const M3 Rb = quat_to_mat(s.root + 3);
  LegKin k;
  leg_kinematics(lm_, Rb, pb, vb, wb, s.q, s.qd, k);
  float lm[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) lm[j] = lm_.f(LM_MASS + j);
  // ---- contact detection (reads only the kinematics): before the mass matrix is live
  if (!EXT_DETECT) {
#pragma unroll 1
    for (int sl = 0; sl < CH_NCP; ++sl) ch_detect_slot<TMESH>(sl, lm_, T, P, k, Rb, pb, cst, lane);
  }
  // ---- leg bias (RNEA, zero generalised acceleration, moments about the base origin)
  float bk[NJ]; V3 Fs = v3(0, 0, 0), Ns = v3(0, 0, 0);
  {
    V3 wp = wb, alp = v3(0, 0, 0), aOp = v3(0, 0, 0), Op = pb;
    V3 F[NJ], NP[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const V3 d = k.O[j] - Op;
      const V3 aO = aOp + cross(alp, d) + cross(wp, cross(wp, d));
      const V3 al = alp + s.qd[j] * cross(wp, k.ax[j]);
      const V3 w = k.w[j];
      const V3 rc = k.com[j] - k.O[j];
      const V3 ac = aO + cross(al, rc) + cross(w, cross(w, rc));
      F[j] = lm[j] * (ac - P.grav);
      NP[j] = mul(k.Ic[j], al) + cross(w, mul(k.Ic[j], w)) + cross(k.com[j] - pb, F[j]);
      wp = w; alp = al; aOp = aO; Op = k.O[j];
    }
#pragma unroll
    for (int j = NJ - 1; j >= 0; --j) {
      Fs = Fs + F[j]; Ns = Ns + NP[j];
      bk[j] = dot(k.ax[j], Ns - cross(k.O[j] - pb, Fs));
    }
  }
  // ---- base body
  const float m0 = m->base_mass + madd, iscale = m0 * frcp(m->base_mass);
  const V3 rc0 = mul(Rb, ld3(m->base_com));
  S3 I0 = rotate_inertia(Rb, m->base_inertia);
  I0.xx *= iscale; I0.xy *= iscale; I0.xz *= iscale; I0.yy *= iscale; I0.yz *= iscale; I0.zz *= iscale;
  // ---- joint-space inertia (CRBA): the leg block Mkk (packed lower) and the base coupling Mbk
  float Mkk[NJ * (NJ + 1) / 2], Mbk[6][NJ];
  float mc = 0; V3 hc = v3(0, 0, 0); S3 Icp = S3{0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = NJ - 1; i >= 0; --i) {
    const V3 r = k.com[i] - pb;
    mc += lm[i]; hc = hc + lm[i] * r; Icp = Icp + inertia_about(k.Ic[i], lm[i], r);
    const V3 w = k.ax[i], o = k.O[i] - pb;
    const V3 F = cross(w, hc - mc * o);
    const V3 Nn = mul(Icp, w) - cross(hc, cross(w, o));
    Mbk[0][i] = F.x; Mbk[1][i] = F.y; Mbk[2][i] = F.z; Mbk[3][i] = Nn.x; Mbk[4][i] = Nn.y; Mbk[5][i] = Nn.z;
#pragma unroll
    for (int j = 0; j <= i; ++j) Mkk[LT(i, j)] = dot(k.ax[j], Nn - cross(k.O[j] - pb, F));
  }
  float Mi[NJ * (NJ + 1) / 2];
  { float Lk[21];
#pragma unroll
    for (int i = 0; i < 21; ++i) Lk[i] = Mkk[i];
    chol6(Lk); spd6_inverse_from_chol(Lk, Mi); }
  float Y[NJ][6];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      float y = 0.f;
#pragma unroll
      for (int i = 0; i < NJ; ++i) y += Mi[PKI(j, i)] * Mbk[a][i];
      Y[j][a] = y;
    }
  float Si[21];
  {
    float L[21];
    const float mt = m0 + grp_sum(mc);
    const V3 ht = m0 * rc0 + grp_sum(hc);
    const S3 It = inertia_about(I0, m0, rc0) + grp_sum(Icp);
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) {
        float sk = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) sk += Mbk[a][j] * Y[j][b];
        L[LT(a, b)] = -grp_sum(sk);
      }
    L[LT(0, 0)] += mt; L[LT(1, 1)] += mt; L[LT(2, 2)] += mt;
    L[LT(3, 1)] += -ht.z; L[LT(3, 2)] += ht.y;
    L[LT(4, 0)] += ht.z;  L[LT(4, 2)] += -ht.x;
    L[LT(5, 0)] += -ht.y; L[LT(5, 1)] += ht.x;
    L[LT(3, 3)] += It.xx; L[LT(4, 3)] += It.xy; L[LT(5, 3)] += It.xz;
    L[LT(4, 4)] += It.yy; L[LT(5, 4)] += It.yz; L[LT(5, 5)] += It.zz;
    chol6(L);
    spd6_inverse_from_chol(L, Si);
  }
  // ---- base bias
  float bb[6];
  {
    const V3 ac = cross(wb, cross(wb, rc0));
    const V3 Fb = m0 * (ac - P.grav);
    const V3 Nb = cross(wb, mul(I0, wb)) + cross(rc0, Fb);
    const V3 Ft = Fb + grp_sum(Fs), Nt = Nb + grp_sum(Ns);
    bb[0] = Ft.x; bb[1] = Ft.y; bb[2] = Ft.z; bb[3] = Nt.x; bb[4] = Nt.y; bb[5] = Nt.z;
  }
  const float mu = 0.5f * (mu_robot + P.terrain_mu);
  const float idt = frcp(dt);
  // ---- per-contact solver data of the active slots
  if (EXT_DETECT) lds_barrier();                         // (A2) the helper waves have written the detection block of every slot
  unsigned my_list = 0; int my_count = 0;
#pragma unroll
  for (int sl = 0; sl < CH_NCP; ++sl)
    if (CHS(sl, CH_ACTIVE) != 0.f) { my_list |= (unsigned)sl << (4 * my_count); ++my_count; }
  // (mirrored halves: round `it` sets up slot 2 it on the lower and slot 2 it + 1 on the upper half -- half the rounds; both write the rows they share)
  const int nround = half < 0 ? CH_NCP : (CH_NCP + 1) / 2;
#pragma unroll 1
  for (int it = 0; it < nround; ++it) {
    const int slq = half < 0 ? it : 2 * it + half, sl = slq < CH_NCP ? slq : CH_NCP - 1;
    const bool act = slq < CH_NCP && CHS(sl, CH_ACTIVE) != 0.f;
    if (!__any(act)) continue;
    const int link = lm_.i(LM_CP_LINK + sl);
    const int lk = link < 0 ? -1 : (link > NJ - 1 ? NJ - 1 : link);
    const V3 n = v3(CHS(sl, CH_N), CHS(sl, CH_N + 1), CHS(sl, CH_N + 2)), r = v3(CHS(sl, CH_R), CHS(sl, CH_R + 1), CHS(sl, CH_R + 2));
    const V3 p = r + pb;
    const V3 a0 = fabsf(n.x) < 0.57735f ? v3(1, 0, 0) : v3(0, 1, 0);
    V3 t1 = cross(a0, n); t1 = __builtin_amdgcn_rsqf(dot(t1, t1)) * t1;
    const V3 t2 = cross(n, t1);
    V3 jk[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) jk[j] = j <= lk ? cross(k.ax[j], p - k.O[j]) : v3(0, 0, 0);
    const V3 dirs[3] = {n, t1, t2};
    float Wb[3][6], Wk[3][NJ], jkv[3][NJ];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const V3 d = dirs[c], rd = cross(r, d);
      float z[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) jkv[c][j] = dot(jk[j], d);
#pragma unroll
      for (int j = 0; j < NJ; ++j) { float zz = 0.f;
#pragma unroll
        for (int i = 0; i < NJ; ++i) zz += Mi[PKI(j, i)] * jkv[c][i];
        z[j] = zz; }
      float g[6] = {d.x, d.y, d.z, rd.x, rd.y, rd.z};
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int j = 0; j < NJ; ++j) g[a] -= Mbk[a][j] * z[j];
      symv6(Si, g, Wb[c]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float w = z[j];
#pragma unroll
        for (int a = 0; a < 6; ++a) w -= Y[j][a] * Wb[c][a];
        Wk[c][j] = w;
        CHS(sl, CH_ZC + NJ * c + j) = z[j];
      }
#pragma unroll
      for (int a = 0; a < 6; ++a) CHS(sl, CH_WB + 6 * c + a) = Wb[c][a];
    }
    float A[3][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const V3 d = dirs[b], rd = cross(r, d);
      const float jb[6] = {d.x, d.y, d.z, rd.x, rd.y, rd.z};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float sacc = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) sacc += jkv[b][j] * Wk[c][j];
#pragma unroll
        for (int a = 0; a < 6; ++a) sacc += jb[a] * Wb[c][a];
        A[b][c] = sacc;
      }
    }
    CHS(sl, CH_T1) = t1.x; CHS(sl, CH_T1 + 1) = t1.y; CHS(sl, CH_T1 + 2) = t1.z; CHS(sl, CH_T2) = t2.x; CHS(sl, CH_T2 + 1) = t2.y; CHS(sl, CH_T2 + 2) = t2.z;
#pragma unroll
    for (int j = 0; j < NJ; ++j) { CHS(sl, CH_JK + 3 * j) = jk[j].x; CHS(sl, CH_JK + 3 * j + 1) = jk[j].y; CHS(sl, CH_JK + 3 * j + 2) = jk[j].z; }
    const float a11 = A[1][1] + P.cfm, a12 = A[1][2], a22 = A[2][2] + P.cfm;
    const float idet = frcp(a11 * a22 - a12 * a12);
    CHS(sl, CH_IANN) = frcp(A[0][0] + P.cfm);
    CHS(sl, CH_AN) = A[1][0]; CHS(sl, CH_AN + 1) = A[2][0];
    const bool pyr = P.fric != LG_FRICTION_CONE;
    CHS(sl, CH_B) = pyr ? frcp(a11) : a22 * idet; CHS(sl, CH_B + 1) = pyr ? a12 : -a12 * idet; CHS(sl, CH_B + 2) = pyr ? a12 : -a12 * idet; CHS(sl, CH_B + 3) = pyr ? frcp(a22) : a11 * idet;
  }
  if (half >= 0) ch_wave_sync();                         // the other half's set-up blocks
  // ---- unconstrained velocity v* = v + dt M^-1 (tau - c)
  float vB[6] = {vb.x, vb.y, vb.z, wb.x, wb.y, wb.z};
  float vK[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) vK[j] = s.qd[j];
  {
    float rk[NJ], y[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) rk[j] = tau[j] - bk[j];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { float yy = 0.f;
#pragma unroll
      for (int i = 0; i < NJ; ++i) yy += Mi[PKI(j, i)] * rk[i];
      y[j] = yy; }
    float g[6], g0[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) { float sk = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j) sk += Mbk[a][j] * y[j];
      g0[a] = -bb[a] - grp_sum(sk); }
    symv6(Si, g0, g);
#pragma unroll
    for (int a = 0; a < 6; ++a) vB[a] += dt * g[a];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      float yk = y[j];
#pragma unroll
      for (int a = 0; a < 6; ++a) yk -= Y[j][a] * g[a];
      vK[j] += dt * yk;
    }
  }
  // ---- joint position limits as unilateral rows on qd
  float jl_sgn[NJ], jl_gap[NJ], jl_iA[NJ], jl_lam[NJ], jl_Wb[NJ][6], jl_y[NJ][NJ];
  bool jl_act[NJ], jl_jw[NJ];
  bool jl_any = false;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const float lo = lm_.f(LM_LOWER + j), hi = lm_.f(LM_UPPER + j);
    const float glo = s.q[j] - lo, ghi = hi - s.q[j];
    const float gap = glo <= ghi ? glo : ghi;
    jl_sgn[j] = glo <= ghi ? 1.f : -1.f;
    jl_act[j] = (lo < hi) && (gap + fminf(0.f, dt * jl_sgn[j] * vK[j]) < 0.05f);
    jl_gap[j] = gap; jl_lam[j] = 0.f;
    jl_any |= jl_act[j];
    jl_jw[j] = __ballot(jl_act[j]) != 0ull;
  }
  const bool jl_wave = __ballot(jl_any) != 0ull;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    jl_iA[j] = 0.f;
#pragma unroll
    for (int a = 0; a < 6; ++a) jl_Wb[j][a] = 0.f;
#pragma unroll
    for (int i = 0; i < NJ; ++i) jl_y[j][i] = 0.f;
    if (!jl_jw[j]) continue;
#pragma unroll
    for (int i = 0; i < NJ; ++i) jl_y[j][i] = Mi[PKI(i, j)] * jl_sgn[j];
    float gvec[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) { float sk = 0.f;
#pragma unroll
      for (int i = 0; i < NJ; ++i) sk += Mbk[a][i] * jl_y[j][i];
      gvec[a] = -sk; }
    symv6(Si, gvec, jl_Wb[j]);
    float wj = jl_y[j][j];
#pragma unroll
    for (int a = 0; a < 6; ++a) wj -= Y[j][a] * jl_Wb[j][a];
    jl_iA[j] = frcp(jl_sgn[j] * wj + P.cfm);
  }
  // ---- self-collision rows (lg_config.self_collisions; see the pass of lg_physics.h)
  ChSelfRow sc[2];
  bool sc_wave = false;
  {
    const int gb = lane & ~(GRP - 1), lgi = lane & (GRP - 1);
    const int BIG = 0x7fffffff;
    float p0 = P.contact_offset, p1 = P.contact_offset; int i0 = BIG, i1 = BIG;
    for (int i = lgi; i < scol.n; i += GRP) {
      const unsigned pk = scol.pairs[i];
      float ra, rb;
      const V3 ca = ch_sc_sphere(cst, lm_, gb, pk & 255u, (pk >> 8) & 255u, pb, &ra), cb = ch_sc_sphere(cst, lm_, gb, (pk >> 16) & 255u, pk >> 24, pb, &rb);
      const float ph = norm(ca - cb) - ra - rb;
      if (ph < p0) { p1 = p0; i1 = i0; p0 = ph; i0 = i; } else if (ph < p1) { p1 = ph; i1 = i; }
    }
    const float m1 = grp_min_all(p0); const int w1 = grp_min_all(p0 == m1 ? i0 : BIG);
    const float c2 = i0 == w1 ? p1 : p0; const int ci2 = i0 == w1 ? i1 : i0;
    const float m2 = grp_min_all(c2); const int w2 = grp_min_all(c2 == m2 ? ci2 : BIG);
    const int win[2] = {w1, w2};
    sc_wave = __ballot(w1 != BIG) != 0ull;
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
      ChSelfRow& R = sc[q2];
      R.on = false; R.lam = 0.f; R.phi = 0.f; R.iA = 0.f; R.n = v3(0, 0, 1); R.slot_a = -1; R.slot_b = -1;
#pragma unroll
      for (int j = 0; j < NJ; ++j) { R.f[j] = 0.f; R.Wk[j] = 0.f; }
#pragma unroll
      for (int a = 0; a < 6; ++a) R.Wb[a] = 0.f;
      if (__ballot(win[q2] != BIG) == 0ull) continue;
      const bool on = win[q2] != BIG;
      const unsigned pk = scol.pairs[on ? win[q2] : 0];
      const int la = pk & 255u, sa = (pk >> 8) & 255u, lb = (pk >> 16) & 255u, sb = pk >> 24;
      float ra, rb;
      const V3 ca = ch_sc_sphere(cst, lm_, gb, la, sa, pb, &ra), cb = ch_sc_sphere(cst, lm_, gb, lb, sb, pb, &rb);
      const V3 d = ca - cb; const float dist = norm(d);
      const bool ok = on && dist > 1e-9f;
      const V3 n = sel3(ok, frcp(dist) * d, v3(0, 0, 1));
      const float phi = dist - ra - rb;
      const V3 pc = cb + (rb + 0.5f * phi) * n;
      const int link_a = __float_as_int(lm_.t[(LM_CP_LINK + sa) * GRP + la]), link_b = __float_as_int(lm_.t[(LM_CP_LINK + sb) * GRP + lb]);
      const int ka = link_a < 0 ? -1 : (link_a > NJ - 1 ? NJ - 1 : link_a), kb = link_b < 0 ? -1 : (link_b > NJ - 1 ? NJ - 1 : link_b);
      const bool mine_a = ok && lgi == la, mine_b = ok && lgi == lb;
      float f[NJ], z[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const float jn = dot(n, cross(k.ax[j], pc - k.O[j]));
        f[j] = (mine_a && j <= ka ? jn : 0.f) - (mine_b && j <= kb ? jn : 0.f);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) { float zz = 0.f;
#pragma unroll
        for (int i = 0; i < NJ; ++i) zz += Mi[PKI(j, i)] * f[i];
        z[j] = zz; }
      float gvec[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) { float sk = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) sk += Mbk[a][j] * z[j];
        gvec[a] = -grp_sum(sk); }
      symv6(Si, gvec, R.Wb);
      float A = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float w = z[j];
#pragma unroll
        for (int a = 0; a < 6; ++a) w -= Y[j][a] * R.Wb[a];
        R.Wk[j] = w; R.f[j] = f[j]; A += f[j] * w;
      }
      A = grp_sum(A) + P.cfm;
      R.on = ok; R.phi = phi; R.n = n; R.iA = ok ? frcp(A) : 0.f;
      R.slot_a = mine_a ? (link_a < 0 ? 0 : (link_a > NJ ? NJ + 1 : link_a + 1)) : -1;
      R.slot_b = mine_b ? (link_b < 0 ? 0 : (link_b > NJ ? NJ + 1 : link_b + 1)) : -1;
    }
  }
  // ---- contact / self-collision / limit rows: TGS sub-intervals or PGS sweeps
  int my_steps = 0;
#pragma unroll
  for (int j = 0; j < CH_NCP; ++j) my_steps += __ballot(my_count > j) != 0ull ? 1 : 0;
  float dqB[6] = {0, 0, 0, 0, 0, 0}, dqK[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) dqK[j] = 0.f;
  const bool tgs = P.solver == LG_SOLVER_TGS;
  const int iters = P.iters > 0 ? P.iters : 1;
  const float h = tgs ? dt * frcp((float)iters) : dt, ih = frcp(h);
  const float tgsf = tgs ? 1.f : 0.f;
  const float erp_ih = P.erp * ih;
  const bool pyr = P.fric != LG_FRICTION_CONE;
  int idle_sl = 0;                                       // a slot every lane has a complete record for (set up in this launch): what idle lanes read
#pragma unroll
  for (int sl = CH_NCP - 1; sl >= 0; --sl) if (__ballot(CHS(sl, CH_ACTIVE) != 0.f) != 0ull) idle_sl = sl;
  if (my_steps > 0 || jl_wave || sc_wave) {
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
      for (int step = 0; step < my_steps; ++step) {
        const bool active = step < my_count;
        const int sl = active ? (int)((my_list >> (4 * step)) & 0xfu) : idle_sl;
        // the whole record in ONE batch of 16-byte LDS reads, then arithmetic on registers (a lone wave gets a fifth of the LDS rate on 4-byte reads, and
        // read-next-to-use put every round trip on the dependent chain: the lesson of the three-joint kernels' load_slot_record)
        float rec[CH_FIELDS];
        {
          const float4* rp = reinterpret_cast<const float4*>(&CHS(sl, 0));
#pragma unroll
          for (int q4 = 0; q4 < CH_FIELDS / 4; ++q4) { const float4 v4 = rp[q4]; rec[4 * q4] = v4.x; rec[4 * q4 + 1] = v4.y; rec[4 * q4 + 2] = v4.z; rec[4 * q4 + 3] = v4.w; }
        }
        const V3 n = v3(rec[CH_N], rec[CH_N + 1], rec[CH_N + 2]), r = v3(rec[CH_R], rec[CH_R + 1], rec[CH_R + 2]);
        const V3 t1 = v3(rec[CH_T1], rec[CH_T1 + 1], rec[CH_T1 + 2]), t2 = v3(rec[CH_T2], rec[CH_T2 + 1], rec[CH_T2 + 2]);
        V3 dp = v3(dqB[0], dqB[1], dqB[2]) + cross(v3(dqB[3], dqB[4], dqB[5]), r);
        V3 vp = v3(vB[0], vB[1], vB[2]) + cross(v3(vB[3], vB[4], vB[5]), r);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const V3 jk = v3(rec[CH_JK + 3 * j], rec[CH_JK + 3 * j + 1], rec[CH_JK + 3 * j + 2]);
          dp = dp + dqK[j] * jk; vp = vp + vK[j] * jk;
        }
        const float l0 = rec[CH_L0], l1 = rec[CH_L1], l2 = rec[CH_L2];
        const float sep = fmaf(tgsf, dot(n, dp), rec[CH_GAP]);
        const float bn = sep >= 0.f ? -sep * ih : fminf(-sep * erp_ih, P.max_depen);
        const float u0 = dot(n, vp), u1 = dot(t1, vp), u2 = dot(t2, vp);
        const float ln = fmaxf(l0 - (u0 - bn) * rec[CH_IANN], 0.f);
        const float dn = ln - l0;
        const float w1 = fmaf(rec[CH_AN], dn, u1), w2 = fmaf(rec[CH_AN + 1], dn, u2);
        const float lim = mu * ln;
        float c1 = l1 - (rec[CH_B] * w1 + rec[CH_B + 1] * w2), c2 = l2 - (rec[CH_B + 2] * w1 + rec[CH_B + 3] * w2);      // cone: the inverse tangential block
        { const float m2 = c1 * c1 + c2 * c2; if (m2 > lim * lim) { const float scl = m2 > 0.f ? lim * __builtin_amdgcn_rsqf(m2) : 0.f; c1 *= scl; c2 *= scl; } }
        const float p1 = fminf(fmaxf(l1 - w1 * rec[CH_B], -lim), lim);                                                     // pyramid: (1/A11, A12, A12, 1/A22)
        const float p2 = fminf(fmaxf(l2 - fmaf(rec[CH_B + 1], p1 - l1, w2) * rec[CH_B + 3], -lim), lim);
        const float n1 = pyr ? p1 : c1, n2 = pyr ? p2 : c2;
        const float d0 = active ? dn : 0.f, d1 = active ? n1 - l1 : 0.f, d2 = active ? n2 - l2 : 0.f;
        if (active) { CHS(sl, CH_L0) = ln; CHS(sl, CH_L1) = n1; CHS(sl, CH_L2) = n2; }
        float g[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) g[a] = grp_sum(d0 * rec[CH_WB + a] + d1 * rec[CH_WB + 6 + a] + d2 * rec[CH_WB + 12 + a]);
#pragma unroll
        for (int a = 0; a < 6; ++a) vB[a] += g[a];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          float w = d0 * rec[CH_ZC + j] + d1 * rec[CH_ZC + NJ + j] + d2 * rec[CH_ZC + 2 * NJ + j];
#pragma unroll
          for (int a = 0; a < 6; ++a) w -= Y[j][a] * g[a];
          vK[j] += w;
        }
      }
      if (sc_wave) {
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          ChSelfRow& R = sc[q2];
          float uu = 0.f, ds = 0.f;
#pragma unroll
          for (int j = 0; j < NJ; ++j) { uu += R.f[j] * vK[j]; ds += R.f[j] * dqK[j]; }
          const float u = grp_sum(uu), dsep = grp_sum(ds);
          const float sep = fmaf(tgsf, dsep, R.phi);
          const float bn = sep >= 0.f ? -sep * ih : fminf(-sep * erp_ih, P.max_depen);
          const float ln = fmaxf(R.lam - (u - bn) * R.iA, 0.f);
          const float dl = R.on ? ln - R.lam : 0.f;
          R.lam = R.on ? ln : R.lam;
#pragma unroll
          for (int a = 0; a < 6; ++a) vB[a] = fmaf(dl, R.Wb[a], vB[a]);
#pragma unroll
          for (int j = 0; j < NJ; ++j) vK[j] = fmaf(dl, R.Wk[j], vK[j]);
        }
      }
      if (jl_wave) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (!jl_jw[j]) continue;
          const float u = jl_sgn[j] * vK[j];
          const float gap = fmaf(tgsf * jl_sgn[j], dqK[j], jl_gap[j]);
          const float bn = gap >= 0.f ? -gap * ih : fminf(-gap * erp_ih, 10.f);
          const float ln = fmaxf(jl_lam[j] - (u - bn) * jl_iA[j], 0.f);
          const float dl = jl_act[j] ? ln - jl_lam[j] : 0.f;
          if (jl_act[j]) jl_lam[j] = ln;
          float gq[6];
#pragma unroll
          for (int a = 0; a < 6; ++a) gq[a] = grp_sum(dl * jl_Wb[j][a]);
#pragma unroll
          for (int a = 0; a < 6; ++a) vB[a] += gq[a];
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) {
            float w = dl * jl_y[j][jj];
#pragma unroll
            for (int a = 0; a < 6; ++a) w -= Y[jj][a] * gq[a];
            vK[jj] += w;
          }
        }
      }
      const bool close = tgs || it == iters - 1;
      if (close) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const float vl = lm_.f(LM_VEL_LIMIT + j); if (vl > 0.f) vK[j] = fminf(fmaxf(vK[j], -vl), vl); }
#pragma unroll
        for (int a = 0; a < 6; ++a) dqB[a] = fmaf(h, vB[a], dqB[a]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) dqK[j] = fmaf(h, vK[j], dqK[j]);
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const float vl = lm_.f(LM_VEL_LIMIT + j); if (vl > 0.f) vK[j] = fminf(fmaxf(vK[j], -vl), vl); dqK[j] = dt * vK[j]; }
#pragma unroll
    for (int a = 0; a < 6; ++a) dqB[a] = dt * vB[a];
  }
  // ---- net contact force per body (world frame): the step's impulse / dt
  if (fbody) {
    V3 fb[NJ + 2];
#pragma unroll
    for (int b = 0; b < NJ + 2; ++b) fb[b] = v3(0, 0, 0);
#pragma unroll 1
    for (int sl = 0; sl < CH_NCP; ++sl) {
      const bool on = CHS(sl, CH_ACTIVE) != 0.f;
      if (!__any(on)) continue;
      const V3 n = v3(CHS(sl, CH_N), CHS(sl, CH_N + 1), CHS(sl, CH_N + 2));
      const V3 t1 = v3(CHS(sl, CH_T1), CHS(sl, CH_T1 + 1), CHS(sl, CH_T1 + 2)), t2 = v3(CHS(sl, CH_T2), CHS(sl, CH_T2 + 1), CHS(sl, CH_T2 + 2));
      V3 f = idt * (CHS(sl, CH_L0) * n + CHS(sl, CH_L1) * t1 + CHS(sl, CH_L2) * t2);
      f = sel3(on, f, v3(0, 0, 0));
      const int link = lm_.i(LM_CP_LINK + sl);
      const int slotb = link < 0 ? 0 : (link > NJ ? NJ + 1 : link + 1);
#pragma unroll
      for (int b = 0; b < NJ + 2; ++b) if (b == slotb) fb[b] = fb[b] + f;
    }
    if (sc_wave) {
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const V3 f = (idt * sc[q2].lam) * sc[q2].n;
#pragma unroll
        for (int b = 0; b < NJ + 2; ++b) {
          if (b == sc[q2].slot_a) fb[b] = fb[b] + f;
          if (b == sc[q2].slot_b) fb[b] = fb[b] - f;
        }
      }
    }
    fbody[0] = grp_sum(fb[0]);
#pragma unroll
    for (int b = 1; b < NJ + 2; ++b) fbody[b] = fb[b];
  }
  // ---- pose advance by the step's displacement (semi-implicit Euler)
#pragma unroll
  for (int a = 0; a < 6; ++a) s.root[7 + a] = vB[a];
  s.root[0] += dqB[0]; s.root[1] += dqB[1]; s.root[2] += dqB[2];
  {
    const V3 w = v3(dqB[3], dqB[4], dqB[5]); const float ang = norm(w);
    float sh, ch; sincos_fast(0.5f * ang, &sh, &ch);
    sh = ang > 1e-12f ? sh * frcp(ang) : 0.5f;
    const float dq0 = sh * w.x, dq1 = sh * w.y, dq2 = sh * w.z, dq3 = ch;
    float* qq = s.root + 3;
    const float x = dq3 * qq[0] + dq0 * qq[3] + dq1 * qq[2] - dq2 * qq[1];
    const float y = dq3 * qq[1] - dq0 * qq[2] + dq1 * qq[3] + dq2 * qq[0];
    const float z = dq3 * qq[2] + dq0 * qq[1] - dq1 * qq[0] + dq2 * qq[3];
    const float w4 = dq3 * qq[3] - dq0 * qq[0] - dq1 * qq[1] - dq2 * qq[2];
    const float inv = __builtin_amdgcn_rsqf(x * x + y * y + z * z + w4 * w4);
    qq[0] = x * inv; qq[1] = y * inv; qq[2] = z * inv; qq[3] = w4 * inv;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) { s.qd[j] = vK[j]; s.q[j] += dqK[j]; }
}
