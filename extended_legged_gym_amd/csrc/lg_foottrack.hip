// lg_foottrack.hip — what FootTrackElSpider adds to one env step (reference envs/elspider_air/elspider.py:547-676 with the foothold planner of type 1,
// utils/raibert_planner.py:304-497, utils/math_utils.py:217-288), as three launches around lg_post_physics_step instead of ~350 small PyTorch launches
// (4.7 ms per step at 4096 envs in round 5's first version): the 0.5 m stray rule, the five planner reward terms before the positivity clip, their episode
// sums / extras means, the planner's re-anchoring at reset envs, the 94-entry observation row with the class's noise vector, and the planner's own step
// (two random walks, base pose, gait, six footholds).  One lane per env, everything of an env in registers.  Arithmetic and order are those of the torch
// layer (envs/elspider_air/elspider.py: FootTrackElSpider._after_native over utils/raibert_planner.py), which vectors recorded from the reference's classes
// pin and which stays as the checker (tests/test_elspider.py); the random draws of a step are handed in (uniforms for the base walk, normals for the
// foothold walk, uniforms for the observation noise), so both see the same numbers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lgstep.h"
#include "lg_device.h"

namespace {

struct Q4 { float x, y, z, w; };
struct F3 { float x, y, z; };
LG_DEV F3 f3(float x, float y, float z) { return F3{x, y, z}; }
LG_DEV F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
LG_DEV F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
LG_DEV F3 operator*(float s, F3 a) { return f3(s * a.x, s * a.y, s * a.z); }
LG_DEV F3 cross3(F3 a, F3 b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
LG_DEV float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
LG_DEV float norm3(F3 a) { return sqrtf(dot3(a, a)); }
// isaacgym.torch_utils, restated in utils/isaac_torch_utils.py (same operation order)
LG_DEV Q4 qmul(Q4 a, Q4 b) {
  const float ww = (a.z + a.x) * (b.x + b.y), yy = (a.w - a.y) * (b.w + b.z), zz = (a.w + a.y) * (b.w - b.z), xx = ww + yy + zz;
  const float qq = 0.5f * (xx + (a.z - a.x) * (b.x - b.y));
  return Q4{qq - xx + (a.x + a.w) * (b.x + b.w), qq - yy + (a.w - a.x) * (b.y + b.z), qq - zz + (a.z + a.y) * (b.w - b.x), qq - ww + (a.z - a.y) * (b.y - b.z)};
}
LG_DEV Q4 qconj(Q4 q) { return Q4{-q.x, -q.y, -q.z, q.w}; }
LG_DEV F3 qapply(Q4 q, F3 v) {                           // quat_apply: v + w t + q_vec x t, t = 2 q_vec x v
  const F3 u = f3(q.x, q.y, q.z), t = 2.f * cross3(u, v);
  return v + q.w * t + cross3(u, t);
}
LG_DEV F3 qrot(Q4 q, F3 v, float sgn) {                  // quat_rotate (+1) / quat_rotate_inverse (-1)
  const F3 u = f3(q.x, q.y, q.z);
  const F3 a = (2.f * q.w * q.w - 1.f) * v, b = (q.w * 2.f) * cross3(u, v), c = (dot3(u, v) * 2.f) * u;
  return f3(a.x + sgn * b.x + c.x, a.y + sgn * b.y + c.y, a.z + sgn * b.z + c.z);
}
LG_DEV Q4 qnormalize(Q4 q) { const float n = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-9f); return Q4{q.x / n, q.y / n, q.z / n, q.w / n}; }
LG_DEV Q4 q_about_z(float angle) { const float h = angle / 2.f; return qnormalize(Q4{0.f, 0.f, sinf(h), cosf(h)}); }     // quat_from_angle_axis(angle, (0, 0, 1))
LG_DEV Q4 ypr(float yaw, float pitch, float roll) {      // math_utils.ypr_to_quat: q_roll * q_pitch * q_yaw
  const float y = yaw * 0.5f, p = pitch * 0.5f, r = roll * 0.5f;
  return qmul(qmul(Q4{sinf(r), 0.f, 0.f, cosf(r)}, Q4{0.f, sinf(p), 0.f, cosf(p)}), Q4{0.f, 0.f, sinf(y), cosf(y)});
}
LG_DEV Q4 ldq(const float* p) { return Q4{p[0], p[1], p[2], p[3]}; }
LG_DEV F3 ld3(const float* p) { return f3(p[0], p[1], p[2]); }
LG_DEV void stq(float* p, Q4 q) { p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w; }
LG_DEV void st3(float* p, F3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
LG_DEV float fremf(float a) { return a - floorf(a); }    // torch.remainder(a, 1.0)

__global__ __launch_bounds__(256) void foottrack_stray_kernel(int n, const float* __restrict__ root, const float* __restrict__ pbase, float* __restrict__ diff,
                                                              uint8_t* __restrict__ stray) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const F3 d = ld3(root + (size_t)e * 13) - ld3(pbase + (size_t)e * 3);
  const float r = norm3(d);
  diff[e] = r;
  stray[e] = r > 0.5f ? 1 : 0;
}

// RandomWalker.step of one env: DIM coordinates; `fresh` = the target a due env takes
template <int DIM, bool CLAMP>
LG_DEV void walker_step(float* cur, float* tgt, float* timer, const float* fresh, float dt, float interval, float max_vel, const float* lo, const float* hi) {
  float t = *timer - dt;
  const bool due = t <= 0.f;
  if (due) t = interval;
  *timer = t;
  float d[DIM]; float s = 0.f;
#pragma unroll
  for (int i = 0; i < DIM; ++i) { if (due) tgt[i] = fresh[i]; d[i] = tgt[i] - cur[i]; s += d[i] * d[i]; }
  const float dist = sqrtf(s);
  const float f = fminf(dist, max_vel) / (dist + 1e-6f);
#pragma unroll
  for (int i = 0; i < DIM; ++i) {
    float c = cur[i] + d[i] * f * dt;
    if (CLAMP) c = fminf(fmaxf(c, lo[i]), hi[i]);
    cur[i] = c;
  }
}

__global__ __launch_bounds__(128) void foottrack_env_kernel(lg_foottrack_params P, lg_foottrack_state S, int n, int planner_stepped,
                                                            const float* __restrict__ nat_obs, const float* __restrict__ nat_rew, const uint8_t* __restrict__ reset,
                                                            const uint8_t* __restrict__ time_out, const float* __restrict__ rigid, const float* __restrict__ cforce,
                                                            const float* __restrict__ root, const float* __restrict__ commands, int cmd_stride,
                                                            const float* __restrict__ u_base, const float* __restrict__ n_foot, const float* __restrict__ noise_u,
                                                            const float* __restrict__ noise_vec, float* __restrict__ obs_out, float* __restrict__ rew_out,
                                                            float* __restrict__ sums, double* __restrict__ acc) {
  const int e = blockIdx.x * 128 + threadIdx.x;
  double a5[5] = {0, 0, 0, 0, 0}, cnt = 0.0;
  if (e < n) {
    const int B = P.num_bodies;
    // ---- planner state of this env
    F3 bpos = ld3(S.base_pos + (size_t)e * 3), bshift = ld3(S.base_pos_shift + (size_t)e * 3);
    Q4 bq = ldq(S.base_quat + (size_t)e * 4), bqs = ldq(S.base_quat_shift + (size_t)e * 4);
    F3 xw = ld3(S.base_x_world + (size_t)e * 3), yw = ld3(S.base_y_world + (size_t)e * 3);
    F3 foot[6]; float phase[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { foot[i] = ld3(S.foot_pos + ((size_t)e * 6 + i) * 3); phase[i] = S.gait_phases[(size_t)e * 6 + i]; }
    float gidx = S.gait_idx[e];
    float bw[6], fw[18];
#pragma unroll
    for (int i = 0; i < 6; ++i) bw[i] = S.bw_cur[(size_t)e * 6 + i];
#pragma unroll
    for (int i = 0; i < 18; ++i) fw[i] = S.fw_cur[(size_t)e * 18 + i];
    // ---- the five terms (elspider.py:660-674) on the pre-reset pose, the planner as the last step left it
    const float* rb = rigid + (size_t)e * B * 13;
    const F3 rpos = ld3(rb); const Q4 rq = ldq(rb + 3);
    float t[5];
    t[0] = norm3(bshift - rpos);
    { const Q4 dq = qmul(rq, qconj(bqs)); t[1] = sqrtf(dq.x * dq.x + dq.y * dq.y + dq.z * dq.z); }
    float r_foot = 0.f, r_z = 0.f, r_sw = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int fb = P.feet_indices[i];
      const F3 fp = ld3(rb + (size_t)fb * 13);
      r_foot += expf(-norm3(foot[i] - fp) / P.reward_sigma);
      r_z += fabsf(foot[i].z - fp.z);
      const bool contact = cforce[((size_t)e * B + fb) * 3 + 2] > 1.f;
      const bool filt = contact || S.last_contacts[(size_t)e * 6 + i] != 0;
      S.last_contacts[(size_t)e * 6 + i] = contact ? 1 : 0;
      const float swing = (planner_stepped && phase[i] < 0.5f) ? 1.f : 0.f;      // (before its first step the planner holds all six feet down)
      r_sw += filt ? swing : 0.f;
    }
    t[2] = r_foot; t[3] = r_z; t[4] = r_sw;
    const bool rs = reset[e] != 0;
    const float term = P.scale_termination * ((rs && !time_out[e]) ? 1.f : 0.f);
    float rew = nat_rew[e] - term;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      if (P.scales[k] != 0.f) {
        const float r = t[k] * P.scales[k];
        rew += r;
        const float sk = sums[(size_t)k * n + e] + r;
        sums[(size_t)k * n + e] = rs ? 0.f : sk;
        if (rs) a5[k] = (double)sk;
      }
    }
    if (P.only_positive_rewards) rew = fmaxf(rew, 0.f);
    rew_out[e] = rew + term;
    if (rs) cnt = 1.0;
    // ---- re-anchor at a reset env's new pose (raibert_planner.py:386-402): the walkers' current values, the world axes the last step left
    const float* rt = root + (size_t)e * 13;
    const F3 npos = ld3(rt); const Q4 nq = ldq(rt + 3);
    if (rs) {
      bpos = f3(npos.x, npos.y, bw[2]);
      bshift = bpos + bw[0] * xw + bw[1] * yw;
      xw = qapply(nq, f3(1, 0, 0));
      bq = q_about_z(atan2f(xw.y, xw.x));
      bqs = qmul(bq, ypr(bw[3], bw[4], bw[5]));
#pragma unroll
      for (int i = 0; i < 6; ++i) foot[i] = qrot(bq, f3(fw[3 * i], fw[3 * i + 1], fw[3 * i + 2]), 1.f) + bpos;
    }
    // ---- the 94-entry row (elspider.py:561-581): native [0:9] | planner 31 | native [12:66], the class's noise vector, clip
    {
      float* o = obs_out + (size_t)e * 94;
      const float* no = nat_obs + (size_t)e * 66;
      float row[94];
#pragma unroll
      for (int i = 0; i < 9; ++i) row[i] = no[i];
      const F3 rel = qrot(nq, bshift - npos, -1.f);
      row[9] = rel.x; row[10] = rel.y; row[11] = rel.z;
      const Q4 qr = qmul(qconj(nq), bqs);
      row[12] = qr.x; row[13] = qr.y; row[14] = qr.z; row[15] = qr.w;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const F3 fr = qrot(nq, foot[i] - npos, -1.f);
        row[16 + 3 * i] = fr.x; row[17 + 3 * i] = fr.y; row[18 + 3 * i] = fr.z;
        row[34 + i] = phase[i] > 0.5f ? 1.f : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 54; ++i) row[40 + i] = no[12 + i];
#pragma unroll
      for (int i = 0; i < 94; ++i) {
        float v = row[i];
        if (P.add_noise) v += (2.f * noise_u[(size_t)e * 94 + i] - 1.f) * noise_vec[i];
        o[i] = fminf(fmaxf(v, -P.clip_observations), P.clip_observations);
      }
    }
    // ---- the planner's step (raibert_planner.py:404-446)
    {
      float fresh6[6], fresh18[18];
#pragma unroll
      for (int i = 0; i < 6; ++i) fresh6[i] = u_base[(size_t)e * 6 + i] * (P.base_bounds[1][i] - P.base_bounds[0][i]) + P.base_bounds[0][i];
#pragma unroll
      for (int i = 0; i < 18; ++i) fresh18[i] = P.foot_mean[i] + P.foot_sigma[i] * n_foot[(size_t)e * 18 + i];
      float tg6[6], tg18[18];
#pragma unroll
      for (int i = 0; i < 6; ++i) tg6[i] = S.bw_tgt[(size_t)e * 6 + i];
#pragma unroll
      for (int i = 0; i < 18; ++i) tg18[i] = S.fw_tgt[(size_t)e * 18 + i];
      float tm6 = S.bw_timer[e], tm18 = S.fw_timer[e];
      walker_step<6, true>(bw, tg6, &tm6, fresh6, P.dt, P.base_interval, P.base_max_vel, P.base_bounds[0], P.base_bounds[1]);
      walker_step<18, false>(fw, tg18, &tm18, fresh18, P.dt, P.foot_interval, P.foot_max_vel, nullptr, nullptr);
      S.bw_timer[e] = tm6; S.fw_timer[e] = tm18;
#pragma unroll
      for (int i = 0; i < 6; ++i) { S.bw_cur[(size_t)e * 6 + i] = bw[i]; S.bw_tgt[(size_t)e * 6 + i] = tg6[i]; }
#pragma unroll
      for (int i = 0; i < 18; ++i) { S.fw_cur[(size_t)e * 18 + i] = fw[i]; S.fw_tgt[(size_t)e * 18 + i] = tg18[i]; }
    }
    const float* cmd = commands + (size_t)e * cmd_stride;
    const float cx = cmd[0], cy = cmd[1], cw = cmd[2];
    xw = qapply(bq, f3(1, 0, 0)); yw = qapply(bq, f3(0, 1, 0));
    F3 pmid[6]; Q4 qmid[6];
    const F3 vel = f3(xw.x * cx + yw.x * cy, xw.y * cx + yw.y * cy, xw.z * cx + yw.z * cy);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const float dur = fremf(1.75f - phase[i]) * P.gait_period;
      pmid[i] = f3(bpos.x + vel.x * dur, bpos.y + vel.y * dur, bpos.z + vel.z * dur);
      qmid[i] = qmul(q_about_z(cw * dur), bq);
    }
    bq = qmul(q_about_z(cw * P.dt), bq);
    bqs = qmul(bq, ypr(bw[3], bw[4], bw[5]));
    bpos = f3(bpos.x + vel.x * P.dt, bpos.y + vel.y * P.dt, bw[2]);
    bshift = bpos + bw[0] * xw + bw[1] * yw;
    gidx = fremf(gidx + P.dt / P.gait_period);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      phase[i] = fremf(gidx + P.phase_offsets[i]);
      const bool sw = phase[i] < 0.5f;
      const F3 nominal = qrot(qmid[i], f3(fw[3 * i], fw[3 * i + 1], fw[3 * i + 2]), 1.f) + pmid[i];
      if (sw) {
        foot[i].x = nominal.x * P.swing_ema + foot[i].x * (1.f - P.swing_ema);
        foot[i].y = nominal.y * P.swing_ema + foot[i].y * (1.f - P.swing_ema);
      }
      foot[i].z = sw ? P.swing_height * sinf(2.f * 3.14159265358979323846f * phase[i]) : 0.f;      // sin_swing_traj (phase < 0.5 here)
    }
    // ---- state back
    st3(S.base_pos + (size_t)e * 3, bpos); st3(S.base_pos_shift + (size_t)e * 3, bshift);
    stq(S.base_quat + (size_t)e * 4, bq); stq(S.base_quat_shift + (size_t)e * 4, bqs);
    st3(S.base_x_world + (size_t)e * 3, xw); st3(S.base_y_world + (size_t)e * 3, yw);
    S.gait_idx[e] = gidx;
#pragma unroll
    for (int i = 0; i < 6; ++i) { st3(S.foot_pos + ((size_t)e * 6 + i) * 3, foot[i]); S.gait_phases[(size_t)e * 6 + i] = phase[i]; }
  }
  // extras means: sums of the reset envs' episode sums and their count (double atomics; a handful of resets per step)
  if (cnt != 0.0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) if (a5[k] != 0.0) atomicAdd(acc + k, a5[k]);
    atomicAdd(acc + 5, cnt);
  }
}

__global__ void foottrack_finish_kernel(lg_foottrack_params P, float* __restrict__ extras, double* __restrict__ acc) {
  const int k = threadIdx.x;
  const double cnt = acc[5];
  __syncthreads();
  if (k < 5 && cnt > 0.0 && P.scales[k] != 0.f) extras[k] = (float)(acc[k] / cnt) / P.max_episode_length_s;     // (torch.mean over the reset envs, then / max_episode_length_s)
  __syncthreads();
  if (k < 6) acc[k] = 0.0;
}

// the launches run on the device the rows live on, whatever device is current in the calling thread (like every other entry point of the library)
static int device_of(const void* p) {
  hipPointerAttribute_t pa;
  return hipPointerGetAttributes(&pa, p) == hipSuccess ? pa.device : -1;
}

}  // namespace

extern "C" {

int lg_foottrack_stray(int32_t n, const float* root_states, const float* planner_base_pos, float* pos_diff, uint8_t* stray, void* stream) {
  if (n <= 0 || !root_states || !planner_base_pos || !pos_diff || !stray) return LG_ERR_INVALID;
  const int dev = device_of(root_states);
  if (dev < 0) return LG_ERR_INVALID;
  DeviceScope ds_(dev);
  if (!ds_.ok) return LG_ERR_HIP;
  (void)hipGetLastError();
  hipLaunchKernelGGL(foottrack_stray_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, root_states, planner_base_pos, pos_diff, stray);
  return hipGetLastError() == hipSuccess ? LG_OK : LG_ERR_HIP;
}

int lg_foottrack_layer_step(const lg_foottrack_params* params, const lg_foottrack_state* state, int32_t n, int32_t planner_stepped, const float* nat_obs,
                            const float* nat_rew, const uint8_t* reset, const uint8_t* time_out, const float* rigid_body_state, const float* contact_forces,
                            const float* root_states, const float* commands, int32_t cmd_stride, const float* u_base, const float* n_foot, const float* noise_u,
                            const float* noise_scale_vec, float* obs_out, float* rew_out, float* sums, float* extras, double* acc, void* stream) {
  if (!params || !state || n <= 0 || !nat_obs || !nat_rew || !reset || !time_out || !rigid_body_state || !contact_forces || !root_states || !commands ||
      !u_base || !n_foot || !obs_out || !rew_out || !sums || !extras || !acc) return LG_ERR_INVALID;
  if (params->add_noise && (!noise_u || !noise_scale_vec)) return LG_ERR_INVALID;
  const int dev = device_of(obs_out);
  if (dev < 0 || dev != device_of(state->base_pos)) return LG_ERR_INVALID;
  DeviceScope ds_(dev);
  if (!ds_.ok) return LG_ERR_HIP;
  (void)hipGetLastError();
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(foottrack_env_kernel, dim3((n + 127) / 128), dim3(128), 0, st, *params, *state, n, planner_stepped, nat_obs, nat_rew, reset, time_out,
                     rigid_body_state, contact_forces, root_states, commands, cmd_stride, u_base, n_foot, noise_u, noise_scale_vec, obs_out, rew_out, sums, acc);
  hipLaunchKernelGGL(foottrack_finish_kernel, dim3(1), dim3(64), 0, st, *params, extras, acc);
  return hipGetLastError() == hipSuccess ? LG_OK : LG_ERR_HIP;
}

}  // extern "C"
