// lg_pose.hip — what PoseAnymal / PoseGo2 add to one env step (reference envs/anymal_c/anymal.py:146-250, envs/go2/go2.py:146-246), as two
// launches behind lg_step instead of ~25 small PyTorch launches: pose-command draws at callback / reset time, the `orientation` and
// `base_height` terms against the COMMANDED pose, the positivity clip, the episode sums / extras means of the two terms, and the 52-entry
// observation row with the class's noise vector.  Arithmetic and order are those of `pose_layer_step` (envs/anymal_c/anymal.py), which the
// golden vectors of the reference class pin and which stays as the checker (tests/test_pose_layer.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lgstep.h"
#include "lg_device.h"

namespace {

// expected projected gravity for (pitch, roll) shifts: quat_rotate_inverse(q_pitch * q_roll, (0, 0, -1)); the heading factor is a rotation
// about the gravity axis and drops out (pose_expected_gravity)
LG_DEV void expected_gravity_xy(float pitch, float roll, float* gx, float* gy) {
  // q_pitch = (0, -sin(p/2), 0, cos(p/2)), q_roll = (sin(r/2), 0, 0, cos(r/2)), xyzw; q = q_pitch * q_roll
  const float sp = -sinf(0.5f * pitch), cp = cosf(0.5f * pitch), sr = sinf(0.5f * roll), cr = cosf(0.5f * roll);
  // quat_mul(a = (0, sp, 0, cp), b = (sr, 0, 0, cr)): x = aw bx + ax bw + ay bz - az by, ...
  const float qx = cp * sr, qy = sp * cr, qz = -sp * sr, qw = cp * cr;
  // quat_rotate_inverse(q, v): v (2 w^2 - 1) - 2 w (q_vec x v) + 2 q_vec (q_vec . v), v = (0, 0, -1)
  const float a = 2.f * qw * qw - 1.f;
  const float cx = qy * -1.f - qz * 0.f, cy = qz * 0.f - qx * -1.f;      // q_vec x v
  const float d = -qz;                                                    // q_vec . v
  *gx = 0.f * a - 2.f * qw * cx + 2.f * qx * d;
  *gy = 0.f * a - 2.f * qw * cy + 2.f * qy * d;
}

__global__ __launch_bounds__(256) void pose_env_kernel(lg_pose_params P, int n, float* __restrict__ pose_cmd, int cmd_stride, float* __restrict__ sums,
                                                       const float* __restrict__ nat_rew, const uint8_t* __restrict__ reset, const uint8_t* __restrict__ time_out,
                                                       const int64_t* __restrict__ eplen_before, const float* __restrict__ base_z, int base_z_stride,
                                                       const float* __restrict__ pg, const float* __restrict__ heights, const float* __restrict__ u,
                                                       float* __restrict__ rew_out, double* __restrict__ acc) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  double a0 = 0.0, a1 = 0.0, cnt = 0.0;
  if (e < n) {
    float c[4];
    const bool cb = (eplen_before[e] + 1) % P.resampling_steps == 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float lo = P.ranges[k][0], span = P.ranges[k][1] - P.ranges[k][0];
      c[k] = cb ? lo + span * u[(size_t)e * 8 + k] : pose_cmd[(size_t)e * cmd_stride + k];
    }
    float gx, gy;
    expected_gravity_xy(c[1], c[2], &gx, &gy);
    const float dx = gx - pg[(size_t)e * 3], dy = gy - pg[(size_t)e * 3 + 1];
    const float r_orient = dx * dx + dy * dy;
    const float bz = base_z[(size_t)e * base_z_stride];
    float bh = bz;
    if (heights) {
      float s = 0.f;
      for (int p = 0; p < P.num_heights; ++p) s += bz - heights[(size_t)e * P.num_heights + p];
      bh = s / (float)P.num_heights;
    }
    const float r_height = (bh - c[3]) * (bh - c[3]);
    const float t0 = r_orient * P.scale_orientation, t1 = r_height * P.scale_base_height;
    const bool rs = reset[e] != 0;
    const float term = P.scale_termination * ((rs && !time_out[e]) ? 1.f : 0.f);
    float rew = nat_rew[e] - term + t0 + t1;
    if (P.only_positive_rewards) rew = fmaxf(rew, 0.f);
    rew_out[e] = rew + term;
    const float s0 = sums[e] + t0, s1 = sums[(size_t)n + e] + t1;
    if (rs) { a0 = s0; a1 = s1; cnt = 1.0; }
    sums[e] = rs ? 0.f : s0; sums[(size_t)n + e] = rs ? 0.f : s1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float lo = P.ranges[k][0], span = P.ranges[k][1] - P.ranges[k][0];
      pose_cmd[(size_t)e * cmd_stride + k] = rs ? lo + span * u[(size_t)e * 8 + 4 + k] : c[k];
    }
  }
  // sums of the envs reset in this step -> extras means (legged_robot.py:200-206): wave sums, one double atomic per wave
  for (int off = 32; off > 0; off >>= 1) { a0 += __shfl_xor(a0, off); a1 += __shfl_xor(a1, off); cnt += __shfl_xor(cnt, off); }
  if ((threadIdx.x & 63) == 0 && cnt > 0.0) { atomicAdd(acc, a0); atomicAdd(acc + 1, a1); atomicAdd(acc + 2, cnt); }
}

// observation rows: [native 0..11 | pose_cmd 4 | native 12..] + noise, clipped; lane = (env, entry).  Block 0 also turns the accumulators
// into the extras means and clears them for the next step.
__global__ __launch_bounds__(256) void pose_obs_kernel(lg_pose_params P, int n, int O, const float* __restrict__ pose_cmd, int cmd_stride,
                                                       const float* __restrict__ nat_obs, const float* __restrict__ noise_u,
                                                       const float* __restrict__ noise_scale, float* __restrict__ obs_out,
                                                       float* __restrict__ extras, double* __restrict__ acc) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const double cnt = acc[2];
    if (cnt > 0.0) { extras[0] = (float)(acc[0] / cnt / (double)P.max_episode_length_s); extras[1] = (float)(acc[1] / cnt / (double)P.max_episode_length_s); }
    acc[0] = 0.0; acc[1] = 0.0; acc[2] = 0.0;
  }
  const int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gi >= (int64_t)n * O) return;
  const int e = (int)(gi / O), i = (int)(gi - (int64_t)e * O);
  float v = i < 12 ? nat_obs[(size_t)e * (O - 4) + i] : (i < 16 ? pose_cmd[(size_t)e * cmd_stride + (i - 12)] : nat_obs[(size_t)e * (O - 4) + (i - 4)]);
  if (noise_u) v += (2.f * noise_u[gi] - 1.f) * noise_scale[i];
  obs_out[gi] = fminf(fmaxf(v, -P.clip_observations), P.clip_observations);
}

}  // namespace

extern "C" int lg_pose_layer_step(const lg_pose_params* p, int32_t n, float* pose_cmd, int32_t cmd_stride, float* sums, float* extras,
                                  const float* nat_obs, const float* nat_rew, const uint8_t* reset, const uint8_t* time_out,
                                  const int64_t* eplen_before, const float* base_z, int32_t base_z_stride, const float* projected_gravity,
                                  const float* measured_heights, const float* u, const float* noise_u, const float* noise_scale_vec,
                                  float* obs_out, float* rew_out, double* acc, void* stream) {
  if (!p || n <= 0 || !pose_cmd || !sums || !extras || !nat_obs || !nat_rew || !reset || !time_out || !eplen_before || !base_z ||
      !projected_gravity || !u || !obs_out || !rew_out || !acc || p->resampling_steps <= 0 || (noise_u && !noise_scale_vec) ||
      (measured_heights && p->num_heights <= 0))
    return LG_ERR_INVALID;
  const int O = (p->num_proprio > 0 ? p->num_proprio : 48) + 4 + (measured_heights ? p->num_heights : 0);      // PoseElSpider: 66 + 4 (elspider.py:448-467)
  // the launches run on the device the rows live on, whatever device is current in the calling thread (an env on cuda:1 driven from a
  // thread whose current device is cuda:0), like every other entry point of the library
  hipPointerAttribute_t pa_out, pa_cmd;
  if (hipPointerGetAttributes(&pa_out, obs_out) != hipSuccess || hipPointerGetAttributes(&pa_cmd, pose_cmd) != hipSuccess) return LG_ERR_INVALID;
  if (pa_out.device != pa_cmd.device) return LG_ERR_INVALID;
  DeviceScope ds_(pa_out.device);
  if (!ds_.ok) return LG_ERR_HIP;
  hipStream_t st = (hipStream_t)stream;
  (void)hipGetLastError();                             // (a stale error of an earlier call is not this call's)
  hipLaunchKernelGGL(pose_env_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, *p, n, pose_cmd, cmd_stride, sums, nat_rew, reset, time_out,
                     eplen_before, base_z, base_z_stride, projected_gravity, measured_heights, u, rew_out, acc);
  const int64_t tot = (int64_t)n * O;
  hipLaunchKernelGGL(pose_obs_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, *p, n, O, pose_cmd, cmd_stride, nat_obs, noise_u,
                     noise_scale_vec, obs_out, extras, acc);
  return (hipGetLastError() == hipSuccess && hipPeekAtLastError() == hipSuccess) ? LG_OK : LG_ERR_HIP;
}
