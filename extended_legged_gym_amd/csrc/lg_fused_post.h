// lg_fused_post.h — the post-physics step as the TAIL of the physics kernel (lg_step's fused path).
//
// Included by lg_step.hip after the post-physics helpers (reward_term, reset_env, resample_commands, the height probe, the
// statistics step).  The stand-alone post_kernel stays for every split entry point (lg_post_physics_step, subset / rollout
// steps, lg_step_transition); a full policy step of all envs (`lg_step`) instead ends inside physics_kernel:
//
//   * a physics workgroup already holds the final state of its 16 envs in the main wave's registers (root, q, qd, torques,
//     contact forces): nothing is re-read from HBM, the history rows the post step needs (last actions, commands, feet
//     timers, episode sums ...) and its uniform draws are fetched by the three helper waves while the main wave runs the
//     last Gauss-Seidel sweeps (they are idle then);
//   * the per-DOF / per-body reductions run on the main wave in its quad layout (lane = leg, DPP quad sums);
//   * the serial part (callback, termination, reward terms in config order, reset, episode sums) runs once per env on the
//     quad's first lane, on LDS rows, with the SAME functions the post kernel uses (reward_term, reset_env, ...);
//   * meanwhile the helper waves write the rigid-body rows and run the 187-point height scan;
//   * then all four waves write the env rows back and assemble the observation rows (4 envs per wave).
//
// The stand-alone post kernel costs ~21 us per step at 4096 envs (1024 workgroups, a chain of eight barrier-separated narrow
// stages behind a staging pass) plus a kernel boundary; this tail replaces it.  Same arithmetic, same Philox counters, same
// order of side effects (LR:113-153): the two paths agree to summation order (tests/test_hip_fused_step.py).
#pragma once
#define HF(i) (hot[(i)])
#define HI(i) (__float_as_int(hot[(i)]))

// LDS row of one env during the tail (floats); lives in the memory of `xs` (dead after the last contact set-up)
enum { FS_ROOT = 0, FS_DOF = 13, FS_CF = 37, FS_FRB = FS_CF + NBODY_MAX * 3 /* four feet rows x 13 */, FS_ACT = FS_FRB + 52, FS_LACT = FS_ACT + 12,
       FS_LRV = FS_LACT + 12, FS_CMD = FS_LRV + 6, FS_BLA = FS_CMD + 4, FS_BAA = FS_BLA + 3, FS_AIR = FS_BAA + 3, FS_CT = FS_AIR + 4, FS_BLV = FS_CT + 4,
       FS_BAV = FS_BLV + 3, FS_PG = FS_BAV + 3, FS_SUMS = FS_PG + 3, FS_GAIT = FS_SUMS + LG_MAX_REWARD_TERMS, FS_FN = FS_GAIT + 1,
       FS_GFZ = FS_FN + NBODY_MAX /* gait_foot_z of the previous step */, FS_VAL = FS_GFZ + 4 /* critic value (lg_step_transition) */,
       FS_END = FS_VAL + 1, FS_STRIDE = FS_END + (FS_END % 2 == 0 ? 1 : 0) };
static_assert(EPB * FS_STRIDE <= XS_STRIDE * 64, "the env rows of the fused tail must fit the memory of the mass-factor table");
// uniforms + small integers of one env (fetched before the final barrier): in the memory of `xbias` (64 x 12 floats)
enum { FU_U = 0, FU_PRE = EPB * LG_RS_NOISE, FU_PRE_STRIDE = 16 };   // pre: [0,1] episode length (int64) | [2] level | [3] last_contacts (4 bytes)
static_assert(FU_PRE + EPB * FU_PRE_STRIDE <= 64 * 12, "uniforms + pre-step integers must fit the leg-bias table");
// results of the serial part + heights: in the memory of the contact-slot table (dead after the last force read-out)
enum { FM_RK = 0, FM_PART = LG_MAX_REWARD_TERMS, FM_ROOTZ = FM_PART + PART_STRIDE, FM_DID_RESET, FM_ROOT_DIRTY, FM_LASTC /* 4 */, FM_RAW = FM_LASTC + 4 /* 2 x LG_REW_COUNT */,
       FM_STRIDE = FM_RAW + 2 * LG_REW_COUNT + 1 };
enum { FO_STRIDE = 256, FH_HEIGHTS = 0, FH_MISC = EPB * MAX_P, FH_OBS = FH_MISC + EPB * FM_STRIDE + 3 - (FH_MISC + EPB * FM_STRIDE + 3) % 4 /* 16-B aligned */ };
enum { FH_NOISE = FH_OBS + EPB * FO_STRIDE /* observation-noise uniforms, one FO_STRIDE row per env, drawn by the helper waves (fused_noise_*) */ };
enum { FH_TAB = FH_NOISE + EPB * FO_STRIDE /* the per-entry table of the observation rows (pack_obs_table), FO_STRIDE x 16 bytes, staged by the helper waves */ };
static_assert(FH_TAB % 4 == 0, "16-byte aligned table rows");
static_assert(FH_TAB + 4 * FO_STRIDE <= LG_MAX_CP * CF_FIELDS * 64, "heights + per-env results + observation staging and noise rows must fit the memory of the contact-slot table");
static_assert(LG_REW_COUNT <= 32, "reward-term masks (rew_term_mask, 1u << id) are 32 bits wide");

// ---- helper waves, while the main wave runs the last sweeps: history rows, uniforms, pre-step integers -> LDS
// `ids` (optional): row k of the launch is env ids[k] (subset steps); `ro`: RobotBatchRollout.post_physics_step_rollout semantics (no callback,
// no termination of its own, no reset, rewards outside the episode sums: robot_batch_rollout.py:763-817), Philox stream 2.
LG_DEV int fused_env_of(const int32_t* __restrict__ ids, int k) { return ids ? ids[k] : k; }
LG_DEV void fused_prefetch(const DevCtx* __restrict__ C, float* SR, float* UB, int blk, int n, int htid /* 0..191 */, int64_t step, const float* values,
                           const int32_t* __restrict__ ids, bool ro) {
  // Every global load of this function is issued before the first LDS store: per element a load -> store chain inside a ten-way divergent
  // branch (the former form) was five passes of ten memory round trips each, ~17 k cycles -- longer than the sweeps it was meant to hide
  // behind, so the main wave waited 6.4 k cycles per step at barrier (F).
  const lg_config& g = C->cfg;
  const int e0 = blk * EPB, nenv = max(0, min(EPB, n - e0));
  const int K = g.num_reward_terms;
  enum { PER = 12 + 6 + 4 + 3 + 3 + 4 + 4 + 1 + 4, NF = (EPB * PER + 191) / 192, NS = (EPB * LG_MAX_REWARD_TERMS + 191) / 192 };
  // fixed rows, flat index -> (env, entry): row base pointers are wave-uniform, the row is chosen by selects
  const float LG_G* const b_lact = C->last_actions; const float LG_G* const b_lrv = C->last_root_vel; const float LG_G* const b_cmd = C->commands;
  const float LG_G* const b_bla = C->base_lin_acc; const float LG_G* const b_baa = C->base_ang_acc; const float LG_G* const b_air = C->feet_air;
  const float LG_G* const b_ct = C->feet_ctime; const float LG_G* const b_gait = C->gait_idx; const float LG_G* const b_gfz = C->gait_foot_z;
  float fv[NF]; int fdst[NF];
#pragma unroll
  for (int it = 0; it < NF; ++it) {
    const int idx = htid + 192 * it;
    const int el = idx / PER, o = idx - el * PER;
    const bool ok = el < nenv;
    const size_t e = (size_t)fused_env_of(ids, e0 + (ok ? el : 0));
    const float LG_G* src; int dst;
    if (o < 12) { src = b_lact + e * 12 + o; dst = FS_LACT + o; }
    else if (o < 18) { src = b_lrv + e * 6 + (o - 12); dst = FS_LRV + (o - 12); }
    else if (o < 22) { src = b_cmd + e * 4 + (o - 18); dst = FS_CMD + (o - 18); }
    else if (o < 25) { src = b_bla + e * 3 + (o - 22); dst = FS_BLA + (o - 22); }
    else if (o < 28) { src = b_baa + e * 3 + (o - 25); dst = FS_BAA + (o - 25); }
    else if (o < 32) { src = b_air + e * 4 + (o - 28); dst = FS_AIR + (o - 28); }
    else if (o < 36) { src = b_ct + e * 4 + (o - 32); dst = FS_CT + (o - 32); }
    else if (o < 37) { src = b_gait + e; dst = FS_GAIT; }
    else { src = b_gfz + e * 4 + (o - 37); dst = FS_GFZ + (o - 37); }
    fdst[it] = ok ? el * FS_STRIDE + dst : -1;
    fv[it] = *src;
  }
  // episode sums, (K, N) rows: lane -> (term, env) with the env fastest (consecutive addresses)
  float sv[NS]; int sdst[NS];
#pragma unroll
  for (int it = 0; it < NS; ++it) {
    const int idx = htid + 192 * it;
    const int el = idx & (EPB - 1), o = idx >> 4;
    const bool ok = el < nenv && o < K;
    sdst[it] = ok ? el * FS_STRIDE + FS_SUMS + o : -1;
    sv[it] = C->ep_sums[(size_t)(ok ? o : 0) * C->N + fused_env_of(ids, e0 + (ok ? el : 0))];
  }
  static_assert(EPB == 16, "the (term, env) split of the episode sums assumes 16 envs per workgroup");
  int64_t pl = 0; float plevel = 0.f, pval = 0.f; uint32_t plc = 0; float pflags = 0.f;
  if (htid < nenv) {
    const int e = fused_env_of(ids, e0 + htid);
    pl = C->ep_len[e];
    plevel = (g.curriculum && !ro) ? (float)C->levels[e] : 0.f;
    pval = values ? values[e] : 0.f;
    plc = *reinterpret_cast<const uint32_t*>(C->last_contacts + (size_t)e * 4);
    if (ro) pflags = (float)((int)C->reset_buf[e] | ((int)C->time_out[e] << 8));     // a rollout env carries the flags of its main's last step
  }
  // one Philox call per (env, slot group): 8 groups of the 32 control slots (computed while the loads are in flight); rollout steps draw none
  for (int idx = htid; !ro && idx < nenv * (LG_RS_NOISE / 4); idx += 192) {
    const int el = idx / (LG_RS_NOISE / 4), grp = idx - el * (LG_RS_NOISE / 4);
    uniform_draw4(C, fused_env_of(ids, e0 + el), grp, step, 0u, UB + FU_U + el * LG_RS_NOISE + 4 * grp);
  }
#pragma unroll
  for (int it = 0; it < NF; ++it) if (fdst[it] >= 0) SR[fdst[it]] = fv[it];
#pragma unroll
  for (int it = 0; it < NS; ++it) if (sdst[it] >= 0) SR[sdst[it]] = sv[it];
  if (htid < nenv) {
    float* pre = UB + FU_PRE + htid * FU_PRE_STRIDE;
    *reinterpret_cast<int64_t*>(pre) = pl;
    pre[2] = plevel;
    SR[htid * FS_STRIDE + FS_VAL] = pval;
    *reinterpret_cast<uint32_t*>(pre + 3) = plc;
    pre[4] = pflags;
  }
}

// ---- helper waves: the uniforms of the observation noise (LR:250-252), one Philox call per (env, group of 4 entries) with the post kernel's
// counters.  They depend on nothing of this step, so they are drawn while these waves wait for the main wave at (F) -- into registers, the
// LDS that will hold them is still the contact-slot table -- and parked in LDS behind the barrier; the write-back then reads its four
// uniforms with one ds_read_b128 where every wave ran ~240 Philox calls (4 envs x 59 groups) on the tail of the launch.  A/B: -0.6 % on the
// step; drawn behind the height scan instead (rolled loop, nothing held across the barrier) the main wave waits for them at (G2): +0.5 %.
LG_DEV bool fused_noise_predrawn(const float* hot) {       // (kernel-uniform)
  return HI(HC_ADD_NOISE) != 0 && HI(HC_INJECT) == 0 && HI(HC_NUM_OBS) <= FO_STRIDE;
}
LG_DEV void fused_noise_draw(const float* hot, int blk, int n, int htid /* 0..191 */, int64_t step, float nz[NZ_IT][4], const int32_t* __restrict__ ids, bool ro) {
  const int e0 = blk * EPB, nenv = max(0, min(EPB, n - e0));
  const int G4 = (HI(HC_NUM_OBS) + 3) >> 2;
#pragma unroll
  for (int it = 0; it < NZ_IT; ++it) {
    const int idx = htid + 192 * it;
    const int el = idx / G4, gq = idx - el * G4;
    uint32_t o4[4] = {0u, 0u, 0u, 0u};
    if (el < nenv) philox4((uint32_t)fused_env_of(ids, e0 + el), (uint32_t)step, (uint32_t)((LG_RS_NOISE >> 2) + gq), ro ? 2u : 0u, (uint32_t)HI(HC_SEED_LO), (uint32_t)HI(HC_SEED_HI), o4);
#pragma unroll
    for (int i = 0; i < 4; ++i) nz[it][i] = u01(o4[i]);
  }
}
// ... and the per-entry table of the observation rows: behind the row stores of the write-back a global load of it waits for every store issued
// before it (vmcnt counts loads and stores together), ~1.5 k cycles on the tail of the launch
// What the helper waves need from HBM behind (F) that does not depend on the step: this lane's scan point and its two entries of the observation
// table.  Requested in front of (F), where these waves wait for the main wave anyway; behind it each was a memory round trip on their path to (G2).
LG_DEV void fused_prefetch_static(const DevCtx* __restrict__ C, const float* hot, int htid, FusedPre& F) {
  const int P = C->cfg.measure_heights ? C->P : 0;
  const int p = htid < P ? htid : 0;
  F.bx = 0.f; F.by = 0.f;
  if (P > 0) { F.bx = C->height_points[2 * p]; F.by = C->height_points[2 * p + 1]; }
  F.tab[0] = C->obs_tab[htid]; F.tab[1] = C->obs_tab[min(htid + 192, FO_STRIDE - 1)];     // (the table holds >= 256 entries: pack_obs_table)
  F.inject = C->cfg.inject_sim_state; F.gait_on = C->cfg.gait_enabled; F.feet_early = fused_needs_feet_rows(C); F.heights_early = fused_needs_heights_early(C);
  F.per_leg = C->per_leg; F.B = C->B; F.P = P; F.plane = C->ter.mesh_type == LG_MESH_PLANE; F.vscale = C->ter.vscale;
  F.rigid = C->rigid; F.gfz = C->gait_foot_z; F.act = C->actions; F.tq = C->torques; F.heights = C->heights;
}
LG_DEV void fused_stage_obs_table(const float* hot, float* HB, int htid, const FusedPre& F) {
  if (HI(HC_NUM_OBS) > FO_STRIDE) return;
  float4* T = reinterpret_cast<float4*>(HB + FH_TAB);
  T[htid] = F.tab[0];                                     // (entries past O: kind 3, as the lanes past the last group expect)
  if (htid + 192 < FO_STRIDE) T[htid + 192] = F.tab[1];
}
LG_DEV void fused_noise_park(const float* hot, float* HB, int blk, int n, int htid, const float nz[NZ_IT][4]) {
  const int e0 = blk * EPB, nenv = max(0, min(EPB, n - e0));
  const int G4 = (HI(HC_NUM_OBS) + 3) >> 2;
#pragma unroll
  for (int it = 0; it < NZ_IT; ++it) {
    const int idx = htid + 192 * it;
    const int el = idx / G4, gq = idx - el * G4;
    if (el < nenv) *reinterpret_cast<float4*>(HB + FH_NOISE + el * FO_STRIDE + 4 * gq) = make_float4(nz[it][0], nz[it][1], nz[it][2], nz[it][3]);
  }
}
static_assert(NZ_IT * 192 >= EPB * (FO_STRIDE / 4), "the helper lanes cover every (env, Philox group) of rows up to FO_STRIDE entries");

// ---- helper waves, after the final state is published: the height scan of the workgroup's envs (LR:400-401), one point per lane
LG_DEV void fused_height_scan(const DevCtx* __restrict__ C, const float (*xst)[20], float* HB, int blk, int n, int htid, const int32_t* __restrict__ ids, bool ro,
                              const FusedPre& F) {
  const int P = F.P;
  const float bx = F.bx, by = F.by;
  if (P <= 0) return;
  const int e0 = blk * EPB, nenv = max(0, min(EPB, n - e0));
  if (ro) {                  // rollout steps keep the heights the last main step measured (robot_batch_rollout.py:763-817 does not scan)
    if (htid < P)
      for (int el = 0; el < nenv; ++el) HB[FH_HEIGHTS + el * MAX_P + htid] = F.heights[(size_t)fused_env_of(ids, e0 + el) * P + htid];
    return;
  }
  const bool plane = F.plane != 0;
  const int p = htid;
  const bool okp = p < P;
  // every context member of the store loop in a local: read inside it they are re-loaded after each global store (which might have changed
  // *C, for all the compiler knows) -- two dependent scalar round trips per env, ~8 k of this function's 11 k cycles
  float LG_G* const hts = F.heights; const int Pn = P; const float vs = F.vscale;
  HeightProbe hp[EPB];
#pragma unroll
  for (int el = 0; el < EPB; ++el) {
    const float* r = xst[4 * el];          // the quad's published root: pos 0..2; slot 19 of its first two rows: the normalised yaw-only quaternion (z, w)
    hp[el] = terrain_height_probe(C, r[19], xst[4 * el + 1][19], r[0], r[1], bx, by);
  }
#pragma unroll
  for (int el = 0; el < EPB; ++el) {
    if (okp && el < nenv) {
      const float hv = plane ? 0.f : (float)hp[el].h * vs;
      HB[FH_HEIGHTS + el * MAX_P + p] = hv;
      hts[(size_t)fused_env_of(ids, e0 + el) * Pn + p] = hv;
    }
  }
}

// ---- main wave, quad layout: per-DOF reward features (sums over the 12 DOFs), contact-force norms, base-frame quantities
struct FusedMainIn {
  const float* root; const float* q; const float* qd; const float* tau; const float* last_qd; const V3* fbody; bool fault;
};
LG_DEV void fused_main_part1(const DevCtx* __restrict__ C, const float* hot, const LegModel& lm_, float* S, int l, const FusedMainIn& in, float feat[F_COUNT]) {
  const float dt = HF(HC_DT);
  const int per_leg = C->per_leg;
  // state rows
  if (l == 0) {
#pragma unroll
    for (int i = 0; i < 13; ++i) S[FS_ROOT + i] = in.root[i];
    S[FS_CF] = in.fbody[0].x; S[FS_CF + 1] = in.fbody[0].y; S[FS_CF + 2] = in.fbody[0].z;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) { S[FS_DOF + 2 * (3 * l + j)] = in.q[j]; S[FS_DOF + 2 * (3 * l + j) + 1] = in.qd[j]; }
  V3 last = in.fbody[3];
  if (per_leg == 3) last = last + in.fbody[4];
  {
    float* cl = S + FS_CF + (1 + per_leg * l) * 3;
    cl[0] = in.fbody[1].x; cl[1] = in.fbody[1].y; cl[2] = in.fbody[1].z;
    cl[3] = in.fbody[2].x; cl[4] = in.fbody[2].y; cl[5] = in.fbody[2].z;
    cl[6] = last.x; cl[7] = last.y; cl[8] = last.z;
    if (per_leg == 4) { cl[9] = in.fbody[4].x; cl[10] = in.fbody[4].y; cl[11] = in.fbody[4].z; }
    float* fn = S + FS_FN + 1 + per_leg * l;
    fn[0] = norm(in.fbody[1]); fn[1] = norm(in.fbody[2]); fn[2] = norm(last);
    if (per_leg == 4) fn[3] = norm(in.fbody[4]);
    if (l == 0) S[FS_FN] = norm(in.fbody[0]);
  }
  // per-DOF features (RM: torques, dof_vel, dof_acc, action_rate, dof_pos_limits, dof_vel_limits, torque_limits, stand_still)
  float f[F_COUNT];
#pragma unroll
  for (int k = 0; k < F_COUNT; ++k) f[k] = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int d = 3 * l + j;
    const float q_ = in.q[j], qd = in.qd[j], tq = in.tau[j];
    f[F_TQ2] += tq * tq;
    f[F_QD2] += qd * qd;
    { const float a = (in.last_qd[j] - qd) / dt; f[F_ACC2] += a * a; }
    { const float a = S[FS_LACT + d] - S[FS_ACT + d]; f[F_ARATE2] += a * a; }
    { const float lo = q_ - lm_.f(LM_SOFT_LO + j), hi = q_ - lm_.f(LM_SOFT_HI + j); f[F_POSLIM] += -fminf(lo, 0.f) + fmaxf(hi, 0.f); }
    f[F_VELLIM] += fminf(fmaxf(fabsf(qd) - lm_.f(LM_VEL_LIMIT + j) * HF(HC_SOFT_VEL), 0.f), 1.f);
    f[F_TQLIM] += fmaxf(fabsf(tq) - lm_.f(LM_TORQUE_LIMIT + j) * HF(HC_SOFT_TQ), 0.f);
    f[F_STILL] += fabsf(q_ - lm_.f(LM_DEFAULT_POS + j));
  }
#pragma unroll
  for (int k = 0; k < F_COUNT; ++k) feat[k] = quad_sum(f[k]);
  // base-frame quantities (LR:128-134): lane 0 lin vel, 1 lin acc, 2 ang vel, 3 ang acc; projected gravity on lane 0
  {
    const float* root = in.root; const float* lrv = S + FS_LRV;
    const V3 lin = v3(root[7], root[8], root[9]), ang = v3(root[10], root[11], root[12]);
    const V3 x = l == 0 ? lin : l == 1 ? lin - v3(lrv[0], lrv[1], lrv[2]) : l == 2 ? ang : ang - v3(lrv[3], lrv[4], lrv[5]);
    const V3 r = quat_rotate_inverse(root + 3, x);
    const float ema = 0.9f, oma = (float)(1 - 0.9);
    float* dst = S + (l == 0 ? FS_BLV : l == 1 ? FS_BLA : l == 2 ? FS_BAV : FS_BAA);
    if (l == 1 || l == 3) { dst[0] = dst[0] * ema + oma * r.x / dt; dst[1] = dst[1] * ema + oma * r.y / dt; dst[2] = dst[2] * ema + oma * r.z / dt; }
    else { dst[0] = r.x; dst[1] = r.y; dst[2] = r.z; }
    const V3 gr = quat_rotate_inverse(root + 3, v3(0, 0, -1));
    if (l == 0) { S[FS_PG] = gr.x; S[FS_PG + 1] = gr.y; S[FS_PG + 2] = gr.z; }
  }
}

// ---- the serial part of one env, on ONE lane (LR:113-153 in the reference's order): callback, termination, rewards, reset,
// episode sums.  `S` = the env's LDS row, `U` = its 32 uniforms, `pre` = pre-step integers, `M` = result row, `H` = its heights.
//
// Everything the reward terms read is fetched into registers in ONE batch of LDS / scalar loads first (the feet's contact
// forces and timers, the base-frame vectors, the index lists of the model); the terms themselves (RM:41-234, restated from
// reward_term() above, which stays the reference for the post kernel and is what tests/test_hip_fused_step.py compares this
// with) are then arithmetic on registers.  Evaluated through reward_term() -- pointer-chasing through LDS with a dependent
// scalar load per model index -- this part was a ~12 k-cycle latency chain per step on a lane with nothing else to run.
struct FusedRewardRegs {
  float blv[3], bav[3], pg[3], cmd[4], bla[2], rootz;
  float air[4], ct[4]; bool lastc[4];
  float cfx[4], cfy[4], cfz[4], fnf[4];        // feet: contact force, its norm
  float fz[4], fvx[4], fvy[4], fvz[4];         // feet rows: height, velocity (valid when a term needs them)
  float ncoll, bh;
  float gait_idx, gait_fz[4];
};
// Raw values of every enabled reward term (before its scale), straight-line: RAW0[id] = what a term sees when it is evaluated
// BEFORE _reward_feet_air_time in the config order, RAW1[id] = at / after it (that term rewrites the feet timers and
// last_contacts, RM:150-163, and five terms read them).  No loop, no switch: the terms' inputs are in registers and every
// branch is on the kernel-uniform term mask, so nothing waits on a dependent load.
LG_DEV void fused_reward_all(const DevCtx* __restrict__ C, const float* hot, unsigned mask, FusedRewardRegs& R, const float feat[F_COUNT], int64_t step,
                             float* RAW0, float* RAW1) {
  const float dt = HF(HC_DT);
  const float cmdn = sqrtf(R.cmd[0] * R.cmd[0] + R.cmd[1] * R.cmd[1]);
#define ON(ID) (((mask >> (ID)) & 1u) != 0u)
#define PUT(ID, expr) if (ON(ID)) { const float v_ = (expr); RAW0[ID] = v_; RAW1[ID] = v_; }
  PUT(LG_REW_LIN_VEL_Z, SQ(R.blv[2]))
  const bool stand = HI(HC_STAND) != 0;     // StandAnymal / StandGo2 overrides (anymal.py:264-308), see enum lg_reward_class
  PUT(LG_REW_ANG_VEL_XY, stand ? SQ(R.bav[1]) + SQ(R.bav[2]) : SQ(R.bav[0]) + SQ(R.bav[1]))
  PUT(LG_REW_ORIENTATION, stand ? SQ(R.pg[1]) + SQ(R.pg[2]) : SQ(R.pg[0]) + SQ(R.pg[1]))
  PUT(LG_REW_ORIENTATION_LOAD_ADAPT, SQ(R.pg[0] - R.bla[0] / 9.81f) + SQ(R.pg[1] - R.bla[1] / 9.81f))
  if (ON(LG_REW_BASE_HEIGHT)) {
    float s = R.rootz;
    if (HI(HC_MEASURE_H)) s = R.bh / (float)HI(HC_P);
    const float v_ = SQ(s - HF(HC_BH_TARGET)); RAW0[LG_REW_BASE_HEIGHT] = v_; RAW1[LG_REW_BASE_HEIGHT] = v_;
  }
  PUT(LG_REW_TORQUES, feat[F_TQ2]) PUT(LG_REW_DOF_VEL, feat[F_QD2]) PUT(LG_REW_DOF_ACC, feat[F_ACC2]) PUT(LG_REW_ACTION_RATE, feat[F_ARATE2])
  PUT(LG_REW_DOF_POS_LIMITS, feat[F_POSLIM]) PUT(LG_REW_DOF_VEL_LIMITS, feat[F_VELLIM]) PUT(LG_REW_TORQUE_LIMITS, feat[F_TQLIM])
  PUT(LG_REW_COLLISION, R.ncoll)
  PUT(LG_REW_STAND_STILL, feat[F_STILL] * (cmdn < 0.1f ? 1.f : 0.f))
  PUT(LG_REW_TRACKING_LIN_VEL, expf(-(stand ? SQ(R.cmd[0] + R.blv[1]) + SQ(R.cmd[1] + R.blv[2]) : SQ(R.cmd[0] - R.blv[0]) + SQ(R.cmd[1] - R.blv[1])) / HF(HC_SIGMA)))
  PUT(LG_REW_TRACKING_ANG_VEL, expf(-SQ(R.cmd[2] - (stand ? R.bav[0] : R.bav[2])) / HF(HC_SIGMA)))
  if (ON(LG_REW_FEET_STUMBLE) || ON(LG_REW_FEET_STUMBLE_LIFTUP)) {
    bool any = false; float s = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f) { const bool st = sqrtf(SQ(R.cfx[f]) + SQ(R.cfy[f])) > 5.f * fabsf(R.cfz[f]); any |= st; s += (st ? 1.f : 0.f) * R.fvz[f]; }
    RAW0[LG_REW_FEET_STUMBLE] = RAW1[LG_REW_FEET_STUMBLE] = any ? 1.f : 0.f;
    RAW0[LG_REW_FEET_STUMBLE_LIFTUP] = RAW1[LG_REW_FEET_STUMBLE_LIFTUP] = s;
  }
  if (ON(LG_REW_FEET_CONTACT_FORCES)) {
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f) s += fmaxf(R.fnf[f] - HF(HC_MAX_CF), 0.f);
    RAW0[LG_REW_FEET_CONTACT_FORCES] = RAW1[LG_REW_FEET_CONTACT_FORCES] = s;
  }
  if (ON(LG_REW_FOUR_FOOTUP)) {
    bool all = true;
#pragma unroll
    for (int f = 0; f < 4; ++f) all &= R.cfz[f] < 1.f;
    RAW0[LG_REW_FOUR_FOOTUP] = RAW1[LG_REW_FOUR_FOOTUP] = 0.1f * (all ? 1.f : 0.f);
  }
  if (ON(LG_REW_NO_FLY)) {           // cassie.py:42-45 (a term of the biped's class; here for any config that scales it)
    int nc = 0;
#pragma unroll
    for (int f = 0; f < 4; ++f) nc += R.cfz[f] > 0.1f ? 1 : 0;
    RAW0[LG_REW_NO_FLY] = RAW1[LG_REW_NO_FLY] = nc == 1 ? 1.f : 0.f;
  }
  if (ON(LG_REW_GAIT_SCHEDULER)) {   // gait_scheduler.py:74-81 on the foot heights / phase stored by the previous step
    float s = 0.f;
    if (HI(HC_GAIT_ON) && step > 1) {
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const float ph = fmodf(R.gait_idx + HF(HC_GAIT_PHASE + f), 1.0f);
        const float tgt = ph < 0.5f ? HF(HC_GAIT_SWING) * sinf(6.28318530717958647692f * ph) : 0.f;
        s += SQ(tgt - R.gait_fz[f]);
      }
    }
    RAW0[LG_REW_GAIT_SCHEDULER] = RAW1[LG_REW_GAIT_SCHEDULER] = s;
  }
  // ---- the terms that read the feet timers / last_contacts: once on the old values, once on what feet_air_time leaves
#define FSYNC(a, b) (fminf(SQ(R.air[a] - R.air[b]), 4.f) + fminf(SQ(R.ct[a] - R.ct[b]), 4.f))
#define FASYN(a, b) (fminf(SQ(R.air[a] - R.ct[b]), 4.f) + fminf(SQ(R.ct[a] - R.air[b]), 4.f))
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float* RAW = pass == 0 ? RAW0 : RAW1;
    if (ON(LG_REW_BASE_FOOT_HEIGHT)) {
      float s = 0.f; int n = 0;
#pragma unroll
      for (int f = 0; f < 4; ++f) if (R.ct[f] > 1e-3f) { s += R.fz[f]; ++n; }
      const float est = n > 0 ? s / (float)n : R.rootz - HF(HC_BH_TARGET);
      RAW[LG_REW_BASE_FOOT_HEIGHT] = SQ((R.rootz - est) - HF(HC_BH_TARGET));
    }
    if (ON(LG_REW_FEET_SLIP)) {
      float s = 0.f;
#pragma unroll
      for (int f = 0; f < 4; ++f) { const bool cfl = (R.cfz[f] > 1.f) || R.lastc[f]; const float vn = sqrtf(SQ(R.fvx[f]) + SQ(R.fvy[f])); s += (cfl ? 1.f : 0.f) * SQ(vn); }
      RAW[LG_REW_FEET_SLIP] = s;
    }
    if (ON(LG_REW_JUMP_AIR)) {
      float s = 0.f;
#pragma unroll
      for (int f = 0; f < 4; ++f) { const bool cfl = (R.cfz[f] > 1.f) || R.lastc[f]; s += (cfl ? 0.f : 1.f) * (R.air[f] - 0.5f); }
      RAW[LG_REW_JUMP_AIR] = fmaxf(s - 2.f, 0.f);
    }
    if (ON(LG_REW_GAIT_2_STEP)) {
      const float sr = (FSYNC(0, 3) + FSYNC(1, 2)) / 2;
      const float ar = (FASYN(0, 1) + FASYN(0, 2) + FASYN(3, 2) + FASYN(3, 1)) / 4;
      const float other = HI(HC_HEADING) ? R.cmd[3] : R.cmd[2];
      const bool on = cmdn > 0.1f || fabsf(other) >= 0.05f;
      RAW[LG_REW_GAIT_2_STEP] = (sr + ar) * (on ? 1.f : 0.f);
    }
    if (ON(LG_REW_PENALTY_IN_THE_AIR))             // anymal.py:301-308
      RAW[LG_REW_PENALTY_IN_THE_AIR] = ((R.cfz[1] > 1.f) || R.lastc[1] || (R.cfz[3] > 1.f) || R.lastc[3]) ? 0.f : 1.f;
    if (pass == 0 && ON(LG_REW_FEET_AIR_TIME)) {   // RM:150-163, stateful; stand classes: feet 1 and 3, feet_contact_time untouched
      float s = 0.f;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        if (stand && (f & 1) == 0) continue;
        const bool contact = R.cfz[f] > 1.f; const bool cfl = contact || R.lastc[f];
        R.lastc[f] = contact;
        const float first = (R.air[f] > 0.f && cfl) ? 1.f : 0.f;
        const float a = R.air[f] + dt, ct = R.ct[f] + dt;
        s += (a - 0.5f) * first;
        R.air[f] = a * (cfl ? 0.f : 1.f); if (!stand) R.ct[f] = ct * (cfl ? 1.f : 0.f);
      }
      RAW0[LG_REW_FEET_AIR_TIME] = RAW1[LG_REW_FEET_AIR_TIME] = s * (cmdn > 0.1f ? 1.f : 0.f);
    }
  }
#undef FSYNC
#undef FASYN
#undef PUT
#undef ON
}

// ro: the rollout variant (see fused_prefetch); krow: this env's row of the launch; rew_out / rew_stride: lg_rollout_batch's reward column.
LG_DEV void fused_env_serial(const DevCtx* __restrict__ C, const float* hot_lds, int e, float* S, const float* U, const float* pre, float* M, const float* H,
                             const float feat[F_COUNT], bool fault, int64_t step, const PostSink& K, bool ro, int krow, float* rew_out, int rew_stride) {
  // The scalar block of the tail's constants (HC_DT .. HC_NUM_EXTRA) in registers, fetched as ONE batch of 16-byte LDS reads: read where they are
  // used -- ~100 ds_read_b32 at a uniform address, each with its own wait, on a wave with nothing to overlap them with -- they were most of this
  // function's time (121 s_waitcnt in ~3 600 instructions).  HF / HI below index this copy with compile-time constants (unused words cost nothing);
  // the per-term tables (HC_IDS, HC_SCALES: run-time index) stay in LDS.
  float hv[(HC_IDS + 3) & ~3];
  {
    const float4* h4 = reinterpret_cast<const float4*>(hot_lds);
#pragma unroll
    for (int i = 0; i < (HC_IDS + 3) / 4; ++i) { const float4 t = h4[i]; hv[4 * i] = t.x; hv[4 * i + 1] = t.y; hv[4 * i + 2] = t.z; hv[4 * i + 3] = t.w; }
  }
  const float* const hot = hv;
  // (the context members this function stores through, requested up front: see fused_writeback_obs)
  uint8_t LG_G* const s_reset = C->reset_buf; uint8_t LG_G* const s_tout = C->time_out; int64_t LG_G* const s_eplen = C->ep_len; float LG_G* const s_rew = C->rew;
#ifdef LG_STAMPS
  unsigned long long* stamps = (blockIdx.x == 0 && threadIdx.x == 0) ? C->stamps : nullptr;    // (diagnostic: the serial part's own phases, ids 58-63)
#endif
  STAMP_DECL
  const float dt = HF(HC_DT);
  const int P = HI(HC_P);
  const unsigned term_mask = (unsigned)HI(HC_TERM_MASK);
  const int kterm = HI(HC_KTERM); const float term_scale = HF(HC_TERM_SCALE);
  float* root = S + FS_ROOT; float* cmd = S + FS_CMD;
  const float* fn = S + FS_FN; const float* cf = S + FS_CF;
  // ---- one batch of loads: model index lists, feet quantities, base-frame vectors, counters
  FusedRewardRegs R;
  int fidx[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) fidx[f] = HI(HC_FEET + f);
  float ncoll = 0.f; bool term = false;
#pragma unroll
  for (int i = 0; i < NBODY_MAX; ++i) {          // fixed bounds (a robot of this instance has at most NBODY_MAX bodies): the index loads and the LDS reads behind them go out together
    const bool pen = i < HI(HC_NPEN), trm = i < HI(HC_NTERM);
    const float fp = fn[HI(HC_PEN + i)], ft = fn[HI(HC_TERMB + i)];
    ncoll += (pen && fp > 0.1f) ? 1.f : 0.f;
    term |= trm && ft > 1.f;
  }
  R.ncoll = ncoll;
  { const uint8_t* lc = reinterpret_cast<const uint8_t*>(pre + 3);
#pragma unroll
    for (int f = 0; f < 4; ++f) R.lastc[f] = lc[f] != 0; }
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    R.cfx[f] = cf[3 * fidx[f]]; R.cfy[f] = cf[3 * fidx[f] + 1]; R.cfz[f] = cf[3 * fidx[f] + 2]; R.fnf[f] = fn[fidx[f]];
    R.air[f] = S[FS_AIR + f]; R.ct[f] = S[FS_CT + f];
    R.fz[f] = S[FS_FRB + 13 * f + 2]; R.fvx[f] = S[FS_FRB + 13 * f + 7]; R.fvy[f] = S[FS_FRB + 13 * f + 8]; R.fvz[f] = S[FS_FRB + 13 * f + 9];
    R.gait_fz[f] = S[FS_GFZ + f];
  }
  R.gait_idx = S[FS_GAIT];
#pragma unroll
  for (int i = 0; i < 3; ++i) { R.blv[i] = S[FS_BLV + i]; R.bav[i] = S[FS_BAV + i]; R.pg[i] = S[FS_PG + i]; }
  R.bla[0] = S[FS_BLA]; R.bla[1] = S[FS_BLA + 1];
  R.rootz = root[2];
  const int64_t eplen = *reinterpret_cast<const int64_t*>(pre) + (ro ? 0 : 1);        // LR:122 (not in rollout steps)
  bool root_dirty = false;
  STAMP(58);
  // ---- _post_physics_step_callback (LR:386-403)
  if (!ro && (int)eplen % HI(HC_RESAMPLING_STEPS) == 0) resample_commands(C, cmd, U, LG_RS_CMD_CB);
  if (!ro && HI(HC_HEADING)) {
    float q[4] = {root[3], root[4], root[5], root[6]};
    V3 f = quat_apply(q, v3(1, 0, 0));
    float x = 0.5f * wrap_to_pi(cmd[3] - atan2f(f.y, f.x));
    cmd[2] = fminf(fmaxf(x, -1.f), 1.f);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) R.cmd[i] = cmd[i];
  const bool push_hit = (step >> 32) == 0 ? ((uint32_t)step % (uint32_t)HI(HC_PUSH_INTERVAL) == 0u) : (step % HI(HC_PUSH_INTERVAL) == 0);
  if (!ro && HI(HC_PUSH) && push_hit) {                                              // LR:402-403, 491-496
    root[7] = rand_float(-HF(HC_MAX_PUSH), HF(HC_MAX_PUSH), U[LG_RS_PUSH]);
    root[8] = rand_float(-HF(HC_MAX_PUSH), HF(HC_MAX_PUSH), U[LG_RS_PUSH + 1]);
    root_dirty = true;
  }
  STAMP(59);
  // ---- check_termination (LR:155-160)
  term |= HI(HC_FLIP) && R.pg[2] > 0.f;
  term |= fault;                                   // physics fault / lost env flagged by this launch
  bool tout = (float)eplen > HF(HC_MAX_EPLEN);
  bool row_reset;                                  // what reset_buf[e] holds behind this step (the dense row of lg_step_subset_rows)
  if (ro) {                                        // rollout envs never terminate on their own: the flags keep their last values (robot_batch_rollout.py:806-809)
    const int fl = (int)pre[4];
    tout = (fl >> 8) != 0; term = ((fl & 0xff) != 0 || fault) && !tout;
    if (fault) s_reset[e] = 1;
    row_reset = fault || (fl & 0xff) != 0;
  } else {
    s_tout[e] = tout ? 1 : 0; s_reset[e] = (term || tout) ? 1 : 0;
    s_eplen[e] = eplen;
    row_reset = term || tout;
  }
  if (K.reset_rows) { K.reset_rows[krow] = row_reset ? 1 : 0; K.tout_rows[krow] = tout ? 1 : 0; }
  // ---- compute_reward (LR:215-232): terms in config order; _reward_feet_air_time rewrites the feet timers where it stands in
  // that order (RM:150-163), so earlier terms see the old values and later ones the new, as in the reference
  R.bh = 0.f;
  if ((term_mask >> LG_REW_BASE_HEIGHT) & 1u) for (int p = 0; p < P; ++p) R.bh += root[2] - H[p];
  STAMP(60);
  float rew = 0.f;
  const int K_ = HI(HC_K), kfat = HI(HC_KFAT);
  float* RAW0 = M + FM_RAW; float* RAW1 = RAW0 + LG_REW_COUNT;
  fused_reward_all(C, hot, term_mask, R, feat, step, RAW0, RAW1);
  if ((term_mask >> LG_REW_ASYNC_GAIT_SCHEDULER) & 1u) RAW0[LG_REW_ASYNC_GAIT_SCHEDULER] = RAW1[LG_REW_ASYNC_GAIT_SCHEDULER] = async_gait_value(C->cfg, S + FS_DOF);
#pragma unroll 4
  for (int k = 0; k < K_; ++k) {                 // the reference's sum, in config order (LR:218-224)
    const int id = __float_as_int(hot_lds[HC_IDS + k]);
    const float raw = k < kfat ? RAW0[id] : RAW1[id];
    const float r = id != LG_REW_TERMINATION ? raw * hot_lds[HC_SCALES + k] : 0.f;
    M[FM_RK + k] = r;
    rew += r;
  }
  STAMP(61);
  if (HI(HC_ONLY_POS)) rew = fmaxf(rew, 0.f);
  if (kterm >= 0) {
    const float r = ((term || tout) && !tout ? 1.f : 0.f) * term_scale;
    rew += r; M[FM_RK + kterm] = r;
  }
  s_rew[e] = rew;
  if (rew_out) rew_out[(size_t)krow * rew_stride] = rew;
  if (K.rewards) {                               // PPO.process_env_step (ppo.py:165, 179-183): three roundings
#pragma clang fp contract(off)
    const float boot = K.gamma * (S[FS_VAL] * (tout ? 1.f : 0.f));
    K.rewards[e] = rew + boot;
    K.dones[e] = (term || tout) ? 1.f : 0.f;
  }
#pragma unroll
  for (int f = 0; f < 4; ++f) { S[FS_AIR + f] = R.air[f]; S[FS_CT + f] = R.ct[f]; }       // what _reward_feet_air_time left (reset_env zeroes them)
  float level = pre[2];
  const bool do_reset = !ro && (term || tout);
  if (do_reset) {
    EnvView V;
    uint8_t lastc_unused[4] = {0, 0, 0, 0};
    V.root = root; V.dof = S + FS_DOF; V.cmd = cmd; V.air = S + FS_AIR; V.ctime = S + FS_CT;
    V.blv = S + FS_BLV; V.bav = S + FS_BAV; V.pg = S + FS_PG; V.tq = nullptr; V.act = S + FS_ACT; V.lact = S + FS_LACT; V.bla = S + FS_BLA;
    V.ldv = nullptr; V.cf = cf; V.rb = S + FS_FRB; V.lastc = lastc_unused; V.feet_rows = 1;
    reset_env(C, V, e, 1, U, false);
    root_dirty = true;
    if (HI(HC_CURRICULUM)) level = (float)C->levels[e];
  }
  STAMP(62);
  M[FM_ROOTZ] = root[2];
  M[FM_DID_RESET] = do_reset ? 1.f : 0.f;
  M[FM_ROOT_DIRTY] = root_dirty ? 1.f : 0.f;
#pragma unroll
  for (int f = 0; f < 4; ++f) M[FM_LASTC + f] = R.lastc[f] ? 1.f : 0.f;
  if (HI(HC_GAIT_ON) && !ro) {                                                          // anymal.py:107-110
    float x = fmodf(S[FS_GAIT] + dt / HF(HC_GAIT_PERIOD), 1.0f); if (x < 0.f) x += 1.0f;
    S[FS_GAIT] = x;
  }
  // ---- episode sums and the statistics of LR:200-206
#pragma unroll 4
  for (int k = 0; k < K_; ++k) {
    const float tot = S[FS_SUMS + k] + (ro ? 0.f : M[FM_RK + k]);                 // compute_reward_rollout does not touch the episode sums
    M[FM_PART + k] = do_reset ? tot : 0.f;
    S[FS_SUMS + k] = do_reset ? 0.f : tot;
  }
  M[FM_PART + K_] = do_reset ? 1.f : 0.f;
  M[FM_PART + K_ + 1] = level;
  M[FM_PART + K_ + 2] = do_reset ? (float)eplen : 0.f;
  STAMP(63);
}

// ---- all four waves: write-back of the env rows + observation rows, 4 envs per wave; statistics + arrival by wave 0
// Returns true on the LAST workgroup of the launch to arrive (it then runs finalize_from_acc).
LG_DEV bool fused_writeback_obs(const DevCtx* __restrict__ C, const float* hot, const float* SR, const float* HB, int blk, int n, int tid, int64_t step, unsigned long long* stamps, float* obs_out, bool obs_by_row,
                                const int32_t* __restrict__ ids, bool ro, bool arrive) {
  // arrive: false in all but the last step of a persistent rollout launch (no arrival ticket: workgroups do not wait for each other between steps)
  STAMP_DECL
  const int wv = tid >> 6, ln = tid & 63;
  const int e0 = blk * EPB, nenv = max(0, min(EPB, n - e0));
  const float* MB = HB + FH_MISC;
  // Every context member this function uses, in locals, requested as ONE batch: read where they are used, each is re-loaded behind the stores
  // in front of it (a store might have changed *C, for all the compiler knows -- the pointer comes through late_ctx) and is a scalar round trip
  // of its own: ~30 of them, one after the other, on the tail of the launch.
  float LG_G* const w_root = C->root; float LG_G* const w_dof = C->dof; float LG_G* const w_lact = C->last_actions; float LG_G* const w_ldv = C->last_dof_vel;
  float LG_G* const w_lrv = C->last_root_vel; float LG_G* const w_cmd = C->commands; float LG_G* const w_air = C->feet_air; float LG_G* const w_ct = C->feet_ctime;
  float LG_G* const w_gait = C->gait_idx; float LG_G* const w_blv = C->base_lin_vel; float LG_G* const w_bav = C->base_ang_vel; float LG_G* const w_pg = C->proj_grav;
  float LG_G* const w_bla = C->base_lin_acc; float LG_G* const w_baa = C->base_ang_acc; float LG_G* const w_sums = C->ep_sums; float LG_G* const w_cf = C->cforce;
  uint8_t LG_G* const w_lastc = C->last_contacts; long long LG_G* const w_acc = C->acc; float LG_G* const w_lvl = C->lvl_part; unsigned LG_G* const w_tickets = C->tickets;
  const int w_N = C->N, w_B = C->B;
  // The observation rows' per-lane table and the context members of that phase are requested HERE: behind the row stores below their loads
  // would queue up behind ~20 stores (vmcnt counts both, in order), and a scalar load between the LDS reads of the entry loop makes every
  // `s_waitcnt lgkmcnt(0)` for it wait for the LDS reads too, i.e. the four envs of the wave stop overlapping.
  const int O = HI(HC_NUM_OBS), G4 = (O + 3) >> 2;
  const bool tab_lds = O <= FO_STRIDE;                         // the table is in LDS (fused_stage_obs_table)
  float4 tb0[4];
  if (!tab_lds) {
#pragma unroll
    for (int i = 0; i < 4; ++i) tb0[i] = C->obs_tab[4 * ln + i];
  }
  const float LG_G* const x_extra = C->extra_obs; const float LG_G* const x_inj = C->rand_inject; float LG_G* const x_obs = C->obs;
  // Row stores, 4 envs per wave.  One store instruction covers a row of ALL FOUR envs of the wave (lane = (env q, entry i): rows are
  // 1-24 entries long), small rows share an instruction (the destination is chosen by selects): 12 stores per wave where a store per
  // (env, row) was 60, each with a handful of active lanes.  All LDS reads first (one round trip), then the stores.
  {
    const int K_ = HI(HC_K);
    const int kb = e0 + 4 * wv;                              // first row of this wave; row kb + q is env EQ(q)
#define EQ(q) ((size_t)fused_env_of(ids, min(kb + (q), n - 1)))
#define ROWI(LEN) const int q = ln / (LEN), i = ln - q * (LEN);
#define ROWQ(LEN) ROWI(LEN) const bool ok = q < 4 && 4 * wv + q < nenv; const float* S = SR + (4 * wv + (ok ? q : 0)) * FS_STRIDE;
    float v_root, v_d0, v_d1, v_a = 0.f, v_b = 0.f, v_act, v_lrv, v_ldv, v_s0 = 0.f, v_s1 = 0.f, v_lc;
    bool k_root, k_d, k_a, k_b, k_lrv, k_s0, k_s1, k_lc;
    float LG_G* p_a = nullptr; float LG_G* p_b = nullptr;
    { ROWQ(13) v_root = S[FS_ROOT + i]; k_root = ok; }
    { ROWQ(12) v_d0 = S[FS_DOF + i]; v_d1 = S[FS_DOF + 12 + i]; v_act = S[FS_ACT + i]; v_ldv = S[FS_DOF + 2 * i + 1]; k_d = ok; }
    { ROWQ(6) v_lrv = S[FS_ROOT + 7 + i]; k_lrv = ok; }
    {   // commands | feet_air_time | feet_contact_time | gait_idx: 13 lanes per env
      ROWQ(13)
      const int off = i < 4 ? FS_CMD + i : i < 8 ? FS_AIR + i - 4 : i < 12 ? FS_CT + i - 8 : FS_GAIT;
      v_a = S[off]; k_a = ok;
      const size_t e = EQ(ok ? q : 0);
      p_a = i < 4 ? w_cmd + e * 4 + i : i < 8 ? w_air + e * 4 + (i - 4) : i < 12 ? w_ct + e * 4 + (i - 8) : w_gait + e;
    }
    {   // base_lin_vel | base_ang_vel | projected_gravity | base_lin_acc | base_ang_acc: 15 lanes per env
      ROWQ(15)
      const int r = i / 3, c = i - 3 * r;
      const int off = (r == 0 ? FS_BLV : r == 1 ? FS_BAV : r == 2 ? FS_PG : r == 3 ? FS_BLA : FS_BAA) + c;
      v_b = S[off]; k_b = ok;
      const size_t e = EQ(ok ? q : 0);
      float LG_G* base = r == 0 ? w_blv : r == 1 ? w_bav : r == 2 ? w_pg : r == 3 ? w_bla : w_baa;
      p_b = base + e * 3 + c;
    }
    {   // episode sums, (K, N) rows: lane = (term, env) with the env fastest
      const int q = ln & 3, k0 = ln >> 2, k1 = k0 + 16;
      const bool okq = 4 * wv + q < nenv;
      const float* S = SR + (4 * wv + (okq ? q : 0)) * FS_STRIDE;
      k_s0 = okq && k0 < K_; k_s1 = okq && k1 < K_;
      v_s0 = S[FS_SUMS + (k_s0 ? k0 : 0)]; v_s1 = S[FS_SUMS + (k_s1 ? k1 : 0)];
    }
    { const int q = ln >> 2, f = ln & 3; k_lc = q < 4 && 4 * wv + q < nenv; v_lc = MB[(4 * wv + (k_lc ? q : 0)) * FM_STRIDE + FM_LASTC + f]; }
    static_assert(LG_MAX_REWARD_TERMS <= 32, "two passes of 16 terms cover the episode sums");
    { ROWI(13) if (k_root) w_root[EQ(q) * 13 + i] = v_root; }
    { ROWI(12) if (k_d) { const size_t e = EQ(q); w_dof[e * 24 + i] = v_d0; w_dof[e * 24 + 12 + i] = v_d1; w_lact[e * 12 + i] = v_act; w_ldv[e * 12 + i] = v_ldv; } }
    { ROWI(6) if (k_lrv) w_lrv[EQ(q) * 6 + i] = v_lrv; }
    if (arrive) {   // net contact forces (B x 3 floats per env, the global layout): dense rows from LDS; from the main wave's registers these were 12-15
        // stores of one dword per lane at a 36-48 byte stride, ~2.5 k cycles of that wave's issue alone.  (Not in the inner steps of a persistent rollout
        // launch, round 6: nothing reads them before the last step stores them again.)
      const int B3 = w_B * 3;
      float LG_G* const cf = w_cf;
#pragma unroll
      for (int it = 0; it < (4 * NBODY_MAX * 3 + 63) / 64; ++it) {
        const int idx = ln + 64 * it;
        const int q = (idx >= B3) + (idx >= 2 * B3) + (idx >= 3 * B3), i = idx - q * B3;
        if (idx < 4 * B3 && 4 * wv + q < nenv) cf[EQ(q) * B3 + i] = SR[(4 * wv + q) * FS_STRIDE + FS_CF + i];
      }
    }
    if (k_a) *p_a = v_a;
    if (k_b) *p_b = v_b;
    { const int q = ln & 3, k0 = ln >> 2;
      if (k_s0) w_sums[(size_t)k0 * w_N + EQ(q)] = v_s0;
      if (k_s1) w_sums[(size_t)(k0 + 16) * w_N + EQ(q)] = v_s1; }
    if (k_lc) w_lastc[EQ(ln >> 2) * 4 + (ln & 3)] = v_lc != 0.f ? 1 : 0;
#undef ROWQ
#undef ROWI
#undef EQ
  }
  STAMP(32);
  // statistics of the workgroup's envs (fixed env order), arrival
  const int KP = HI(HC_K) + 3;
  bool any_reset = false;
  for (int el = 0; el < nenv; ++el) any_reset |= MB[el * FM_STRIDE + FM_DID_RESET] != 0.f;
  if (wv == FUSED_STATS_WAVE && ln < KP && any_reset) {
    float sacc = 0.f;
    for (int el = 0; el < nenv; ++el) sacc += MB[el * FM_STRIDE + FM_PART + ln];
    __hip_atomic_fetch_add(w_acc + ln, __double2ll_rn((double)sacc * ACC_SCALE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (wv == FUSED_STATS_WAVE && ln == KP + 1) {
    float sacc = 0.f;
    for (int el = 0; el < nenv; ++el) sacc += MB[el * FM_STRIDE + FM_PART + HI(HC_K) + 1];
    st_dev(w_lvl + blk, sacc);
  }
  unsigned arrival = 0;
  const unsigned shard = (unsigned)blk & 7u, nsh = min(8u, gridDim.x);
  const unsigned want = (gridDim.x + 7u - shard) >> 3;
  if (wv == FUSED_STATS_WAVE) __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): the statistics stores / atomics of this wave have completed
  if (tid == 64 * FUSED_STATS_WAVE && arrive) arrival = __hip_atomic_fetch_add(w_tickets + 32 * shard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  STAMP(33);
  // An inner step of a persistent rollout launch (lg_rollout_batch: `arrive` is false in all but the last step) ends here: its observation rows would be
  // overwritten by the next step before anybody could read them (round 6: -5 % on the horizon; what the NEXT step's prefetch reads -- the state and history
  // rows above -- is stored every step).
  if (!arrive) return false;

  // observation rows (LR:234-252, :107-108): proprio | heights | extra, + uniform noise, clipped.  A lane forms the 4 entries of
  // ONE Philox call (entries 4 gq .. 4 gq + 3, the post kernel's mapping, so the noise is the same draw for draw); what does not
  // depend on the env (noise scales, source offsets, scales) is formed once per lane in front of the env loop.  The entries go
  // through an LDS row so that the global stores are dense (lane = entry): four 16-byte-strided dword stores per env were the
  // most expensive part of this phase (the memory pipeline handles 4 lanes per 64-byte segment).
  const bool inject = HI(HC_INJECT) != 0, add_noise = HI(HC_ADD_NOISE) != 0, predrawn = fused_noise_predrawn(hot);
  const float clip = HF(HC_CLIP_OBS);
  float* OB = const_cast<float*>(HB) + FH_OBS + 4 * wv * FO_STRIDE;  // this wave's staging rows, one per env: with ONE row the four envs of the wave
                                                                      // ran one after the other (row write -> wait -> row reads -> stores, four times)
  for (int g0 = 0; g0 < G4; g0 += 64) {
    const int gq = g0 + ln;
    float4 tb[4];                                  // this lane's four entries of the host-packed table (pack_obs_table)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (tab_lds) tb[i] = reinterpret_cast<const float4*>(HB + FH_TAB)[min(4 * gq + i, FO_STRIDE - 1)];
      else if (g0 == 0) tb[i] = tb0[i];
      else tb[i] = C->obs_tab[4 * gq + i];
    }
    STAMP(49);
    if (!inject && !x_extra && (predrawn || !add_noise)) {
      // the usual case as straight-line code (no injected uniforms, no extra observations, noise parked by the helper waves): nothing between the
      // LDS reads of the four envs but arithmetic, rows past the last env of a ragged workgroup recomputed from the last one (their staging row is private)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int el = min(4 * wv + q, max(nenv - 1, 0));
        const float* S = SR + el * FS_STRIDE; const float* H = HB + FH_HEIGHTS + el * MAX_P;
        const float rootz = MB[el * FM_STRIDE + FM_ROOTZ] - 0.5f;
        float4 t = make_float4(0.5f, 0.5f, 0.5f, 0.5f);
        if (add_noise) t = *reinterpret_cast<const float4*>(HB + FH_NOISE + el * FO_STRIDE + 4 * min(gq, FO_STRIDE / 4 - 1));
        const float u[4] = {t.x, t.y, t.z, t.w};
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int code = __float_as_int(tb[i].x), off = code & 0xffff, kind = code >> 16;
          const float val = (kind == 1 ? H : S)[kind <= 1 ? off : 0];
          const float op = (val - tb[i].z) * tb[i].y;
          const float oh = fminf(fmaxf(rootz - val, -1.f), 1.f) * tb[i].y;
          float v = kind == 0 ? op : (kind == 1 ? oh : 0.f);
          if (add_noise) v += (2.f * u[i] - 1.f) * tb[i].w;
          o[i] = fminf(fmaxf(v, -clip), clip);
        }
        *reinterpret_cast<float4*>(OB + q * FO_STRIDE + 4 * ln) = make_float4(o[0], o[1], o[2], o[3]);
      }
    } else
#pragma unroll
    for (int q = 0; q < 4; ++q) {                 // (unrolled: the four envs' LDS reads interleave)
      const int el = 4 * wv + q;
      if (el >= nenv) continue;
      const int e = fused_env_of(ids, e0 + el);
      const float* S = SR + el * FS_STRIDE; const float* H = HB + FH_HEIGHTS + el * MAX_P;
      const float rootz = MB[el * FM_STRIDE + FM_ROOTZ];
      float u[4] = {0.5f, 0.5f, 0.5f, 0.5f}, ex[4] = {0.f, 0.f, 0.f, 0.f}, val[4];
      int kind[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int code = __float_as_int(tb[i].x), off = code & 0xffff;
        kind[i] = code >> 16;
        val[i] = (kind[i] == 1 ? H : S)[kind[i] <= 1 ? off : 0];
      }
      if (add_noise && inject) {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (4 * gq + i < O) u[i] = x_inj[(size_t)e * (LG_RS_NOISE + O) + LG_RS_NOISE + 4 * gq + i];
      }
      if (x_extra) {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (kind[i] == 2) ex[i] = x_extra[(size_t)e * HI(HC_NUM_EXTRA) + (__float_as_int(tb[i].x) & 0xffff)];
      }
      if (add_noise && !inject) {
        if (predrawn) {                            // (O <= FO_STRIDE: one pass, g0 == 0) drawn by the helper waves in front of (F)
          const float4 t = *reinterpret_cast<const float4*>(HB + FH_NOISE + el * FO_STRIDE + 4 * min(gq, FO_STRIDE / 4 - 1));
          u[0] = t.x; u[1] = t.y; u[2] = t.z; u[3] = t.w;
        } else {
          uint32_t o4[4];
          philox4((uint32_t)e, (uint32_t)step, (uint32_t)((LG_RS_NOISE >> 2) + gq), ro ? 2u : 0u, (uint32_t)HI(HC_SEED_LO), (uint32_t)HI(HC_SEED_HI), o4);
#pragma unroll
          for (int i = 0; i < 4; ++i) u[i] = u01(o4[i]);
        }
      }
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float op = (val[i] - tb[i].z) * tb[i].y;                                              // proprioceptive entry (LR:237-244), post-reset rows
        const float oh = fminf(fmaxf((rootz - 0.5f) - val[i], -1.f), 1.f) * tb[i].y;      // height entry (LR:245-247)
        float v = kind[i] == 0 ? op : (kind[i] == 1 ? oh : ex[i]);
        if (add_noise) v += (2.f * u[i] - 1.f) * tb[i].w;
        o[i] = fminf(fmaxf(v, -clip), clip);
      }
      *reinterpret_cast<float4*>(OB + q * FO_STRIDE + 4 * ln) = make_float4(o[0], o[1], o[2], o[3]);
    }
    STAMP(50);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0): the rows are in LDS (same wave: the LDS pipeline is in order)
    __builtin_amdgcn_wave_barrier();
    float ov[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) ov[q][j] = OB[q * FO_STRIDE + ln + 64 * j];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int el = 4 * wv + q;
      if (el >= nenv) continue;
      const int e = fused_env_of(ids, e0 + el);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = 4 * g0 + ln + 64 * j;
        if (idx < O) {
          x_obs[(size_t)e * O + idx] = ov[q][j];
          if (obs_out) obs_out[(size_t)(obs_by_row ? e0 + el : e) * O + idx] = ov[q][j];     // RolloutStorage.observations[t + 1] | row of lg_step_subset_rows
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  STAMP(34);
  bool last = false;
  if (tid == 64 * FUSED_STATS_WAVE && arrive && arrival == want - 1u) {
    if (__hip_atomic_fetch_add(w_tickets + 32 * 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsh - 1u) {
      last = true;
      for (int i = 0; i < 9; ++i) __hip_atomic_store(w_tickets + 32 * i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  return last;
}
