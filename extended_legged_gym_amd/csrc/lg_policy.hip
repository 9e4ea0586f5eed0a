// lg_policy.hip — gfx950 kernels for the rollout-collection side of PPO (include/lgpolicy.h): fused MLP forward on the
// fp32 matrix cores, PPO.act (both networks + Gaussian sampling + log-prob) in one launch, GAE returns.
//
// MLP kernel.  Workgroup = 8 waves = 32 rows of the batch through ALL layers.  The activations of the current layer live in LDS
// in an image a lane reads with ONE ds_read_b128 per four k-steps: [k/16][row][k%4][(k/4)%4] -- the four values a lane feeds to
// four successive v_mfma_f32_16x16x4_f32 (A operand: lane = (row, k%4)) are adjacent, and the 64 lanes of a wave read 1 KB
// contiguous (conflict-free).  The next layer's image is written to a second LDS buffer by the epilogue (bias + activation on the
// accumulator fragments).  Weights are re-tiled once on the host the same way ([16-column chunk][k/16][lane][(k/4)%4]): one
// coalesced global_load_dwordx4 per wave and eight MFMAs.  Each wave owns every eighth 16-column chunk of a layer and keeps two
// independent accumulators (rows 0-15 and 16-31), which is what the 16x16x4 instruction needs to issue back to back.
// Exact fp32: the MFMA is a k-ordered fmaf chain.
//
// Why 16-byte operand loads and two waves per SIMD (tools/micro/mfma_loop.hip, measured on MI355X): data returning from a global
// load holds the matrix pipe of that SIMD for ~16 cycles per VGPR written and a ds_read for ~4-14, whatever the wave does
// meanwhile -- a two-chain MFMA loop at 36 cycles per MFMA runs at 54 with one global_load_dword + one ds_read2_b32 per MFMA pair
// (the round-1 loop: 80 us for the 235-512-256-128 pair at 4096 rows), at 46 with one dwordx4 + two b128 per eight MFMAs, and at
// 41 with a second wave on the SIMD to fill the gaps.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "lg_device.h"
#include "../../include/lgpolicy.h"
#include "../../include/lgstep.h"

#ifndef LG_AB
#define LG_AB 0
#endif
#define MLP_ROWS 32          // batch rows per workgroup
#define MLP_THREADS 512      // eight waves: two per SIMD
#define MLP_MAXW 512         // widest layer
#define MLP_IMG (MLP_MAXW * MLP_ROWS)          // floats of one activation image
// element (row m, input k) of the activation image
#define IMG(m, k) (((((k) >> 4) * MLP_ROWS + (m)) * 4 + ((k) & 3)) * 4 + (((k) >> 2) & 3))

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MlpDev {
  int L, act;
  int dims[LG_MLP_MAX_LAYERS + 1];
  int kpad[LG_MLP_MAX_LAYERS];       // input width rounded up to 64 (four blocks of four k-steps of 4)
  int nchunks[LG_MLP_MAX_LAYERS];    // output width rounded up to 16, / 16
  const float* w[LG_MLP_MAX_LAYERS]; // tiled weights [chunk][k/16][lane][4]
  const float* b[LG_MLP_MAX_LAYERS]; // bias, padded to 16 * nchunks
};

struct lg_mlp {
  MlpDev h;
  int device = 0;
  std::vector<void*> allocs;
  std::string err;
};

static thread_local std::string g_pol_err;

LG_DEV float apply_act(float x, int act) {
  switch (act) {
    case LG_ACT_ELU: {   // x > 0 ? x : expm1(x): the degree-6 Taylor polynomial for |x| < 0.25 (truncation < 5e-8 relative), exp(x) - 1
                         // below that (fast exp: ~1e-7 absolute on a result of magnitude >= 0.22); libm's expm1f is ~30 instructions
      const float p = x * (1.f + x * (0.5f + x * (1.f / 6 + x * (1.f / 24 + x * (1.f / 120 + x * (1.f / 720))))));
      return x > 0.f ? x : (x > -0.25f ? p : __expf(x) - 1.f);
    }
    case LG_ACT_RELU: return fmaxf(x, 0.f);
    case LG_ACT_TANH: return tanhf(x);
    case LG_ACT_LRELU: return x > 0.f ? x : 0.01f * x;
    case LG_ACT_SELU: return 1.0507009873554805f * (x > 0.f ? x : 1.6732632423543772f * expm1f(x));
  }
  return x;
}

// 32 rows of x through the whole network; result rows (width dims[L], <= 16 * nchunks) left in `yrows` / written to `y_global`
LG_DEV void mlp_tile(const MlpDev& M, const float* __restrict__ x, int64_t row0, int64_t n, float* buf0, float* buf1, float* yrows /* [32][16*?] */,
                     float* __restrict__ y_global) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#if LG_AB == 8      // diagnostic build: phase times of one workgroup (100 MHz wall clock), printed from the kernel
  unsigned long long tstamp[8]; int ts_n = 0;
#define TS() do { __builtin_amdgcn_s_waitcnt(0); tstamp[ts_n++] = wall_clock64(); } while (0)
#else
#define TS()
#endif
  TS();
  // stage the input tile: x (n, K0) row-major -> activation image.  Eight independent loads in flight per lane.
  const int K0 = M.dims[0], K0p = M.kpad[0];
  for (int base = 0; base < MLP_ROWS * K0p; base += 8 * MLP_THREADS) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * MLP_THREADS + tid, r = idx / K0p, k = idx - r * K0p;
      const int64_t row = row0 + r;
      v[u] = (idx < MLP_ROWS * K0p && row < n && k < K0) ? x[row * K0 + k] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * MLP_THREADS + tid, r = idx / K0p, k = idx - r * K0p;
      if (idx < MLP_ROWS * K0p) buf0[IMG(r, k)] = v[u];
    }
  }
  lds_barrier();
  TS();
  float* in = buf0; float* out = buf1;
// volatile asm keeps the two accumulator chains interleaved as written (the compiler otherwise issues the dependent MFMAs of one
// accumulator back to back: 40-cycle dependent latency instead of the 32-cycle issue rate)
#define MFMA_IN_ORDER(ACC, A, B) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
// One block = 16 inputs = four k-steps = eight MFMAs on the fragments (W, A0, A1), in ascending k.  The fragments of the block
// TWO ahead (of this chunk, or of the wave's next chunk) are requested in the middle of the block: sched_barrier pins the order,
// the compiler's s_waitcnt before a block's first MFMA then only waits for loads issued sixteen MFMAs earlier.
#define MLP_BLOCK(W, A0, A1, NW, NA0, NA1, WN, AN)                                                            \
      {                                                                                                       \
        MFMA_IN_ORDER(acc0, A0.x, W.x); MFMA_IN_ORDER(acc1, A1.x, W.x);                                       \
        MFMA_IN_ORDER(acc0, A0.y, W.y); MFMA_IN_ORDER(acc1, A1.y, W.y);                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        NW = *(WN); NA0 = *(AN); NA1 = (AN)[64];                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        MFMA_IN_ORDER(acc0, A0.z, W.z); MFMA_IN_ORDER(acc1, A1.z, W.z);                                       \
        MFMA_IN_ORDER(acc0, A0.w, W.w); MFMA_IN_ORDER(acc1, A1.w, W.w);                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
      }
  for (int l = 0; l < M.L; ++l) {
    const int nblk = M.kpad[l] >> 4, nch = M.nchunks[l];          // blocks of 16 inputs per chunk: a multiple of 4
    const bool last = l == M.L - 1;
    const int nout = M.dims[l + 1];
    // A fragments: lane (row = lane & 15, k%4 = lane >> 4) reads float4 (block * 32 + row) * 4 + k%4: the wave covers 64 consecutive
    // float4 of the block; rows 16-31 sit 64 float4 further
    const float4* ap = reinterpret_cast<const float4*>(in) + (lane & 15) * 4 + (lane >> 4);
    const float4* wl = reinterpret_cast<const float4*>(M.w[l]) + lane;
    float4 w0, w1, w2, w3, p0, p1, p2, p3, q0, q1, q2, q3;     // four rotating fragment sets (weights, rows 0-15, rows 16-31)
    if (wv < nch) {
      const float4* wc = wl + (size_t)wv * nblk * 64;
      w0 = wc[0]; p0 = ap[0]; q0 = ap[64];
      w1 = wc[64]; p1 = ap[128]; q1 = ap[128 + 64];
    }
    for (int c = wv; c < nch; c += MLP_THREADS / 64) {
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const float4* wc = wl + (size_t)c * nblk * 64;
      const int cn = c + MLP_THREADS / 64 < nch ? c + MLP_THREADS / 64 : c;           // the wave's last chunk re-reads its own first blocks
      const float4* wnext = wl + (size_t)cn * nblk * 64;
      const int col = c * 16 + (lane & 15);
      const float bias = M.b[l][col];
      for (int kb = 0; kb < nblk; kb += 4) {
        const bool more = kb + 4 < nblk;
        MLP_BLOCK(w0, p0, q0, w2, p2, q2, wc + (size_t)(kb + 2) * 64, ap + (kb + 2) * 128)
        MLP_BLOCK(w1, p1, q1, w3, p3, q3, wc + (size_t)(kb + 3) * 64, ap + (kb + 3) * 128)
        MLP_BLOCK(w2, p2, q2, w0, p0, q0, more ? wc + (size_t)(kb + 4) * 64 : wnext, more ? ap + (kb + 4) * 128 : ap)
        MLP_BLOCK(w3, p3, q3, w1, p1, q1, more ? wc + (size_t)(kb + 5) * 64 : wnext + 64, more ? ap + (kb + 5) * 128 : ap + 128)
      }
      // the MFMAs above are opaque to the compiler's hazard recogniser: give the last one its result latency before the
      // accumulators are read by the epilogue
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
      // epilogue: C[m = 4 * (lane >> 4) + i][col = lane & 15]
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = 4 * (lane >> 4) + i;
        float v0 = acc0[i] + bias, v1 = acc1[i] + bias;
        if (!last) {
          v0 = apply_act(v0, M.act); v1 = apply_act(v1, M.act);
          if (col >= nout) { v0 = 0.f; v1 = 0.f; }                 // padded columns feed zeros to the next layer
          out[IMG(m, col)] = v0;
          out[IMG(m + 16, col)] = v1;
        } else if (col < nout) {
          if (yrows) { yrows[m * 16 * nch + col] = v0; yrows[(m + 16) * 16 * nch + col] = v1; }
          if (y_global) {
            if (row0 + m < n) y_global[(row0 + m) * nout + col] = v0;
            if (row0 + m + 16 < n) y_global[(row0 + m + 16) * nout + col] = v1;
          }
        }
      }
    }
    TS();
    lds_barrier();
    float* t = in; in = out; out = t;
  }
#if LG_AB == 8
  if (blockIdx.x == 5 && lane == 0 && M.L == 4) printf("wg %d,%d wave %d: stage %llu  L0 %llu  L1 %llu  L2 %llu  L3 %llu  (10 ns ticks)\n", blockIdx.x, blockIdx.y, wv,
      tstamp[1] - tstamp[0], tstamp[2] - tstamp[1], tstamp[3] - tstamp[2], tstamp[4] - tstamp[3], tstamp[5] - tstamp[4]);
#endif
#undef MLP_BLOCK
}

__global__ __launch_bounds__(MLP_THREADS) void mlp_forward_kernel(MlpDev M, const float* __restrict__ x, int64_t n, float* __restrict__ y) {
  __shared__ __attribute__((aligned(16))) float buf0[MLP_IMG];
  __shared__ __attribute__((aligned(16))) float buf1[MLP_IMG];
  mlp_tile(M, x, (int64_t)blockIdx.x * MLP_ROWS, n, buf0, buf1, nullptr, y);
}

// PPO.act: blockIdx.y = 0 actor (+ sampling, log-prob), 1 critic
__global__ __launch_bounds__(MLP_THREADS) void policy_act_kernel(MlpDev A, MlpDev Cr, const float* __restrict__ obs, const float* __restrict__ cobs,
                                                         int64_t n, const float* __restrict__ stdv, uint32_t seed_lo, uint32_t seed_hi,
                                                         uint32_t call_lo, uint32_t call_hi, int deterministic, float* __restrict__ actions,
                                                         float* __restrict__ mean, float* __restrict__ logp, float* __restrict__ values) {
  __shared__ __attribute__((aligned(16))) float buf0[MLP_IMG];
  __shared__ __attribute__((aligned(16))) float buf1[MLP_IMG];
  __shared__ float yrows[MLP_ROWS * 16 * 2];       // output rows of the actor (<= 32 actions)
  __shared__ float lp[MLP_ROWS][32];
  const int64_t row0 = (int64_t)blockIdx.x * MLP_ROWS;
  if (blockIdx.y == 1) { mlp_tile(Cr, cobs, row0, n, buf0, buf1, nullptr, values); return; }
  const int na = A.dims[A.L], stride = 16 * A.nchunks[A.L - 1];
  mlp_tile(A, obs, row0, n, buf0, buf1, yrows, mean);
  // one lane per (row, action): z from Philox + Box-Muller, two normals per counter word pair
  const int tid = threadIdx.x;
  for (int idx = tid; idx < MLP_ROWS * 32; idx += MLP_THREADS) {
    const int r = idx >> 5, a = idx & 31;
    float term = 0.f;
    if (a < na && row0 + r < n) {
      const float mu = yrows[r * stride + a], sd = stdv[a];
      float act = mu;
      if (!deterministic) {
        uint32_t o[4];
        philox4((uint32_t)(row0 + r), (uint32_t)((uint64_t)(row0 + r) >> 32), (uint32_t)(a >> 1), call_lo ^ (call_hi * 0x9E3779B9u), seed_lo, seed_hi, o);
        const float u1 = fmaxf(u01(o[0]), 5.9604645e-8f), u2 = u01(o[1]);
        const float rad = sqrtf(-2.f * logf(u1));
        const float z = (a & 1) ? rad * sinf(6.28318530717958647692f * u2) : rad * cosf(6.28318530717958647692f * u2);
        act = mu + sd * z;
      }
      actions[(row0 + r) * na + a] = act;
      const float d = act - mu;
      term = -(d * d) / (2.f * sd * sd) - logf(sd) - 0.91893853320467274178f;      // Normal.log_prob
    }
    lp[r][a] = term;
  }
  lds_barrier();
  if (tid < MLP_ROWS && row0 + tid < n) {
    float sacc = 0.f;
    for (int a = 0; a < na; ++a) sacc += lp[tid][a];
    logp[row0 + tid] = sacc;
  }
}

// GAE (rollout_storage.py:145-160): one lane per env, the T-step recursion in registers
__global__ __launch_bounds__(256) void gae_kernel(const float* __restrict__ rew, const float* __restrict__ dones, const float* __restrict__ val,
                                                  const float* __restrict__ last, int T, int64_t n, float gamma, float lam,
                                                  float* __restrict__ ret, float* __restrict__ adv) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float a = 0.f, next = last[e];
  for (int t = T - 1; t >= 0; --t) {
    const float v = val[(size_t)t * n + e];
    const float nt = 1.f - dones[(size_t)t * n + e];
    const float delta = rew[(size_t)t * n + e] + nt * gamma * next - v;
    a = delta + nt * gamma * lam * a;
    ret[(size_t)t * n + e] = a + v;
    adv[(size_t)t * n + e] = (a + v) - v;          // self.returns - self.values, as the reference computes it
    next = v;
  }
}

// mean / unbiased std over all entries, then (x - mean) / (std + 1e-8): one workgroup, fixed-order tree (deterministic)
__global__ __launch_bounds__(1024) void normalize_kernel(float* __restrict__ adv, int64_t count) {
  __shared__ double s1[1024], s2[1024];
  const int tid = threadIdx.x;
  double a = 0.0;
  for (int64_t i = tid; i < count; i += 1024) a += (double)adv[i];
  s1[tid] = a;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) { if (tid < off) s1[tid] += s1[tid + off]; __syncthreads(); }
  const double mean = s1[0] / (double)count;
  double q = 0.0;
  for (int64_t i = tid; i < count; i += 1024) { const double d = (double)adv[i] - mean; q += d * d; }
  s2[tid] = q;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) { if (tid < off) s2[tid] += s2[tid + off]; __syncthreads(); }
  const double sd = count > 1 ? sqrt(s2[0] / (double)(count - 1)) : 0.0;
  const float m = (float)mean, inv = 1.f / ((float)sd + 1e-8f);
  for (int64_t i = tid; i < count; i += 1024) adv[i] = (adv[i] - m) * inv;
}

extern "C" {

const char* lg_mlp_last_error(lg_mlp* m) { return m ? m->err.c_str() : g_pol_err.c_str(); }

void lg_mlp_destroy(lg_mlp* m) {
  if (!m) return;
  DeviceScope ds_(m->device);
  for (void* p : m->allocs) (void)hipFree(p);
  delete m;
}

lg_mlp* lg_mlp_create(int32_t L, const int32_t* dims, const float* const* weights, const float* const* biases, int32_t activation,
                      int device_id) {
  if (L <= 0 || L > LG_MLP_MAX_LAYERS || !dims || !weights || !biases) { g_pol_err = "bad layer list"; return nullptr; }
  if (activation < LG_ACT_ELU || activation > LG_ACT_SELU) { g_pol_err = "unknown activation"; return nullptr; }
  for (int l = 0; l <= L; ++l) if (dims[l] <= 0 || dims[l] > MLP_MAXW) { g_pol_err = "layer width out of range (1..512)"; return nullptr; }
  if (dims[L] > 32) { /* fine for lg_mlp_forward; lg_policy_act checks its own limit */ }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_pol_err = "no HIP device: the policy kernels have no CPU path"; return nullptr; }
  if (device_id < 0 || device_id >= ndev) { g_pol_err = "bad device"; return nullptr; }
  DeviceScope ds_(device_id);
  if (!ds_.ok) { g_pol_err = "bad device"; return nullptr; }
  lg_mlp* m = new lg_mlp();
  m->device = device_id; m->h.L = L; m->h.act = activation;
  for (int l = 0; l <= L; ++l) m->h.dims[l] = dims[l];
  for (int l = 0; l < L; ++l) {
    const int K = dims[l], N = dims[l + 1], Kp = (K + 63) & ~63, nb = Kp / 16;
    // hidden layers produce the next layer's whole padded input (zero weights and bias -> act(0) = 0 in the padding)
    const int nch = l == L - 1 ? (N + 15) / 16 : ((N + 63) & ~63) / 16;
    m->h.kpad[l] = Kp; m->h.nchunks[l] = nch;
    // B fragment of v_mfma_f32_16x16x4_f32: lane holds B[k = lane >> 4][n = lane & 15] = W[n][k]; the fragments of the four
    // k-steps of a 16-input block sit in one float4 per lane
    std::vector<float> tw((size_t)nch * nb * 64 * 4, 0.f), tb((size_t)nch * 16, 0.f);
    for (int c = 0; c < nch; ++c)
      for (int b = 0; b < nb; ++b)
        for (int ln = 0; ln < 64; ++ln)
          for (int s = 0; s < 4; ++s) {
            const int nn = c * 16 + (ln & 15), kk = b * 16 + s * 4 + (ln >> 4);
            if (nn < N && kk < K) tw[(((size_t)c * nb + b) * 64 + ln) * 4 + s] = weights[l][(size_t)nn * K + kk];
          }
    for (int i = 0; i < N; ++i) tb[i] = biases[l][i];
    void *dw = nullptr, *db = nullptr;
    if (hipMalloc(&dw, tw.size() * 4) != hipSuccess || hipMalloc(&db, tb.size() * 4) != hipSuccess ||
        hipMemcpy(dw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(db, tb.data(), tb.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
      if (dw) m->allocs.push_back(dw);
      if (db) m->allocs.push_back(db);
      g_pol_err = "weight upload failed"; lg_mlp_destroy(m); return nullptr;
    }
    m->allocs.push_back(dw); m->allocs.push_back(db);
    m->h.w[l] = (const float*)dw; m->h.b[l] = (const float*)db;
  }
  return m;
}

#define POL_TRY(m, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { (m)->err = std::string(#expr) + ": " + hipGetErrorString(_e); return LG_ERR_HIP; } } while (0)

int lg_mlp_forward(lg_mlp* m, const float* x, int64_t n, float* y, void* stream) {
  if (!m || !x || !y || n < 0) return LG_ERR_INVALID;
  DeviceScope ds_(m->device);
  if (n == 0) return LG_OK;
  hipLaunchKernelGGL(mlp_forward_kernel, dim3((unsigned)((n + MLP_ROWS - 1) / MLP_ROWS)), dim3(MLP_THREADS), 0, (hipStream_t)stream, m->h, x, n, y);
  POL_TRY(m, hipGetLastError());
  return LG_OK;
}

int lg_policy_act(lg_mlp* actor, lg_mlp* critic, const float* obs, const float* critic_obs, int64_t n, const float* std_, uint64_t seed,
                  uint64_t call, int32_t deterministic, float* actions, float* action_mean, float* logp, float* values, void* stream) {
  if (!actor || !critic || !obs || !critic_obs || !std_ || !actions || !action_mean || !logp || !values || n < 0) return LG_ERR_INVALID;
  DeviceScope ds_(actor->device);
  if (actor->h.dims[actor->h.L] > 32) { actor->err = "lg_policy_act supports up to 32 actions"; return LG_ERR_UNSUPPORTED; }
  if (n == 0) return LG_OK;
  hipLaunchKernelGGL(policy_act_kernel, dim3((unsigned)((n + MLP_ROWS - 1) / MLP_ROWS), 2), dim3(MLP_THREADS), 0, (hipStream_t)stream, actor->h, critic->h,
                     obs, critic_obs, n, std_, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)call, (uint32_t)(call >> 32), deterministic,
                     actions, action_mean, logp, values);
  POL_TRY(actor, hipGetLastError());
  return LG_OK;
}

int lg_compute_returns(const float* rewards, const float* dones, const float* values, const float* last_values, int32_t T, int64_t n,
                       float gamma, float lam, int32_t normalize, float* returns, float* advantages, void* stream) {
  if (!rewards || !dones || !values || !last_values || !returns || !advantages || T <= 0 || n <= 0) return LG_ERR_INVALID;
  hipPointerAttribute_t pa;                        // no context in this call: run where the rows live
  if (hipPointerGetAttributes(&pa, rewards) != hipSuccess) return LG_ERR_INVALID;
  DeviceScope ds_(pa.device);
  hipLaunchKernelGGL(gae_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rewards, dones, values, last_values, T, n,
                     gamma, lam, returns, advantages);
  if (normalize) hipLaunchKernelGGL(normalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, advantages, (int64_t)T * n);
  return hipGetLastError() == hipSuccess ? LG_OK : LG_ERR_HIP;
}

// the sigma rows of all T transitions (ppo.py:155: action_std broadcast over the envs): one launch for the whole rollout
__global__ __launch_bounds__(256) void fill_sigma_kernel(int64_t rows, int A, const float* __restrict__ std, float* __restrict__ sigma) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * A) return;
  sigma[i] = std[i % A];
}

extern "C" int lg_step_transition(lg_ctx* ctx, const float* actions, float* next_observations, const float* values, float gamma,
                                  float* rewards, float* dones, void* stream);

int lg_collect_rollout(lg_ctx* env, lg_mlp* actor, lg_mlp* critic, const float* std, uint64_t seed, uint64_t first_call, int32_t T,
                       float gamma, float lam, int32_t normalize_advantage, const lg_rollout* out, void* stream) {
  if (!env || !actor || !critic || !std || !out || T <= 0) return LG_ERR_INVALID;
  DeviceScope ds_(actor->device);
  if (!out->observations || !out->actions || !out->rewards || !out->dones || !out->values || !out->actions_log_prob || !out->mu ||
      !out->sigma || !out->last_values) { actor->err = "lg_collect_rollout: null output row"; return LG_ERR_INVALID; }
  void* p; int64_t shp[4]; int32_t nd, dt;
  if (lg_get_tensor(env, LG_T_OBS_BUF, &p, shp, &nd, &dt) != LG_OK) return LG_ERR_INVALID;
  const float* obs = (const float*)p; const int64_t n = shp[0], O = shp[1];
  const int A = actor->h.dims[actor->h.L];
  int64_t ashp[4];
  if (lg_get_tensor(env, LG_T_ACTIONS, &p, ashp, &nd, &dt) != LG_OK) return LG_ERR_INVALID;
  if (actor->h.dims[0] != O || critic->h.dims[0] != O || critic->h.dims[critic->h.L] != 1 || A != ashp[1]) {
    actor->err = "lg_collect_rollout: network widths do not match the env (obs width, one action per DOF, scalar value)"; return LG_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  // the first observation row is copied from the env; every later one is written by the step itself (lg_step_transition),
  // as are the reward (with the time-out bootstrap) and done rows: three launches per step (act, physics, post-physics)
  POL_TRY(actor, hipMemcpyAsync(out->observations, obs, (size_t)n * O * 4, hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(fill_sigma_kernel, dim3((unsigned)(((int64_t)T * n * A + 255) / 256)), dim3(256), 0, st, (int64_t)T * n, A, std, out->sigma);
  for (int t = 0; t < T; ++t) {
    float* obs_t = out->observations + (size_t)t * n * O;
    float* act_t = out->actions + (size_t)t * n * A;
    int rc = lg_policy_act(actor, critic, obs_t, obs_t, n, std, seed, first_call + (uint64_t)t, 0, act_t, out->mu + (size_t)t * n * A,
                           out->actions_log_prob + (size_t)t * n, out->values + (size_t)t * n, stream);
    if (rc != LG_OK) return rc;
    rc = lg_step_transition(env, act_t, t + 1 < T ? out->observations + (size_t)(t + 1) * n * O : nullptr, out->values + (size_t)t * n, gamma,
                            out->rewards + (size_t)t * n, out->dones + (size_t)t * n, stream);
    if (rc != LG_OK) { actor->err = std::string("lg_collect_rollout: lg_step_transition failed: ") + lg_last_error(env); return rc; }
  }
  int rc = lg_mlp_forward(critic, obs, n, out->last_values, stream);
  if (rc != LG_OK) return rc;
  if (out->returns && out->advantages)
    rc = lg_compute_returns(out->rewards, out->dones, out->values, out->last_values, T, n, gamma, lam, normalize_advantage, out->returns,
                            out->advantages, stream);
  POL_TRY(actor, hipGetLastError());
  return rc;
}

}  // extern "C"


// ============================================================================================ sampling planner arithmetic (lgpolicy.h)
// plans[i, h, a] = sum_k phi[h, k] nodes[i, k, a]: one lane per output element, the K node rows of a sample are read coalesced along a
__global__ __launch_bounds__(256) void plan_from_nodes_kernel(const float* __restrict__ nodes, const float* __restrict__ phi, int64_t n, int K, int H, int A,
                                                              float* __restrict__ plans) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * H * A) return;
  const int a = (int)(idx % A); const int64_t ih = idx / A; const int h = (int)(ih % H); const int64_t i = ih / H;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc = fmaf(phi[h * K + k], nodes[(i * K + k) * A + a], acc);
  plans[idx] = acc;
}

// One wave per main env: its R samples' mean rewards, standardised, softmax at the given temperature, weighted mean of the node rows.
// R, H, K * A are tens to hundreds: the whole problem of a main env is a few KB, read once.
__global__ __launch_bounds__(64) void mppi_update_kernel(const float* __restrict__ rewards, const float* __restrict__ nodes, int R, int H, int KA, float temperature,
                                                         float* __restrict__ new_nodes, float* __restrict__ weights) {
  extern __shared__ float w_lds[];                 // R weights
  const int m = blockIdx.x, lane = threadIdx.x;
  const float* rw = rewards + (size_t)m * R * H;
  auto wave_sum = [](float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; };
  auto wave_max = [](float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; };
  float s1 = 0.f;
  for (int i = lane; i < R; i += 64) {
    float r = 0.f;
    for (int h = 0; h < H; ++h) r += rw[(size_t)i * H + h];
    r /= (float)H;
    w_lds[i] = r; s1 += r;
  }
  const float mean = wave_sum(s1) / (float)R;
  float s2 = 0.f;
  for (int i = lane; i < R; i += 64) { const float d = w_lds[i] - mean; s2 += d * d; }
  const float sd = sqrtf(wave_sum(s2) / (float)R);
  const float scale = sd > 1e-12f ? 1.f / (sd * temperature) : 0.f;
  float mx = -3.0e38f;
  for (int i = lane; i < R; i += 64) { const float z = (w_lds[i] - mean) * scale; w_lds[i] = z; mx = fmaxf(mx, z); }
  mx = wave_max(mx);
  float se = 0.f;
  for (int i = lane; i < R; i += 64) { const float e = __expf(w_lds[i] - mx); w_lds[i] = e; se += e; }
  const float inv = 1.f / wave_sum(se);
  for (int i = lane; i < R; i += 64) { const float w = w_lds[i] * inv; w_lds[i] = w; weights[(size_t)m * R + i] = w; }
  __syncthreads();
  const float* nd = nodes + (size_t)m * R * KA;
  for (int j = lane; j < KA; j += 64) {
    float acc = 0.f;
    for (int i = 0; i < R; ++i) acc = fmaf(w_lds[i], nd[(size_t)i * KA + j], acc);
    new_nodes[(size_t)m * KA + j] = acc;
  }
}

static int device_of(const void* p) {
  hipPointerAttribute_t pa;
  return hipPointerGetAttributes(&pa, p) == hipSuccess ? pa.device : -1;
}

int lg_plan_from_nodes(const float* nodes, const float* phi, int64_t n, int32_t K, int32_t H, int32_t A, float* plans, void* stream) {
  if (!nodes || !phi || !plans || n <= 0 || K <= 0 || H <= 0 || A <= 0) return LG_ERR_INVALID;
  const int dev = device_of(nodes);
  if (dev < 0) return LG_ERR_INVALID;
  DeviceScope ds_(dev);
  const int64_t total = n * H * A;
  hipLaunchKernelGGL(plan_from_nodes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, nodes, phi, n, K, H, A, plans);
  return hipGetLastError() == hipSuccess ? LG_OK : LG_ERR_HIP;
}

int lg_mppi_update(const float* rewards, const float* nodes, int32_t num_main, int32_t R, int32_t H, int32_t K, int32_t A, float temperature,
                   float* new_nodes, float* weights, void* stream) {
  if (!rewards || !nodes || !new_nodes || !weights || num_main <= 0 || R <= 0 || H <= 0 || K <= 0 || A <= 0 || !(temperature > 0.f)) return LG_ERR_INVALID;
  if ((size_t)R * sizeof(float) > 60 * 1024) return LG_ERR_UNSUPPORTED;          // the weights of one main env live in LDS
  const int dev = device_of(rewards);
  if (dev < 0) return LG_ERR_INVALID;
  DeviceScope ds_(dev);
  hipLaunchKernelGGL(mppi_update_kernel, dim3((unsigned)num_main), dim3(64), (size_t)R * sizeof(float), (hipStream_t)stream, rewards, nodes, R, H, K * A, temperature,
                     new_nodes, weights);
  return hipGetLastError() == hipSuccess ? LG_OK : LG_ERR_HIP;
}

// lg_mppi_sample_plans: one workgroup per sample row; the row's K x A nodes in LDS between the draw and the interpolation
__global__ __launch_bounds__(128) void mppi_sample_plans_kernel(const float* __restrict__ mean, const float* __restrict__ sigma_nodes, float sigma_scale,
                                                                const float* __restrict__ phi, int R, int K, int H, int A, uint32_t seed_lo, uint32_t seed_hi,
                                                                uint32_t call_lo, uint32_t call_hi, float* __restrict__ nodes, float* __restrict__ plans) {
  extern __shared__ float nd[];                      // [K * A]
  const int i = blockIdx.x, m = i / R, smp = i - m * R, KA = K * A;
  for (int j = threadIdx.x; j < KA; j += blockDim.x) {
    float z = 0.f;
    if (smp != 0) {
      uint32_t o[4];
      philox4((uint32_t)i, call_lo, (uint32_t)(j >> 1), call_hi, seed_lo, seed_hi, o);
      const float u1 = fmaxf(u01(o[0]), 5.9604645e-8f), u2 = u01(o[1]);
      const float rad = sqrtf(-2.f * logf(u1));
      z = (j & 1) ? rad * sinf(6.28318530717958647692f * u2) : rad * cosf(6.28318530717958647692f * u2);
    }
    const float v = mean[(size_t)m * KA + j] + (sigma_scale * sigma_nodes[j / A]) * z;
    nd[j] = v;
    nodes[(size_t)i * KA + j] = v;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < H * A; j += blockDim.x) {
    const int h = j / A, a = j - h * A;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(phi[h * K + k], nd[k * A + a], acc);
    plans[(size_t)i * H * A + j] = acc;
  }
}

int lg_mppi_sample_plans(const float* mean, const float* sigma_nodes, float sigma_scale, const float* phi, int32_t num_main, int32_t R, int32_t K, int32_t H,
                         int32_t A, uint64_t seed, uint64_t call, float* nodes, float* plans, void* stream) {
  if (!mean || !sigma_nodes || !phi || !nodes || !plans || num_main <= 0 || R <= 0 || K <= 0 || H <= 0 || A <= 0) return LG_ERR_INVALID;
  if ((size_t)K * A * sizeof(float) > 48 * 1024) return LG_ERR_UNSUPPORTED;
  const int dev = device_of(mean);
  if (dev < 0) return LG_ERR_INVALID;
  DeviceScope ds_(dev);
  hipLaunchKernelGGL(mppi_sample_plans_kernel, dim3((unsigned)(num_main * R)), dim3(128), (size_t)K * A * sizeof(float), (hipStream_t)stream, mean, sigma_nodes,
                     sigma_scale, phi, R, K, H, A, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)call, (uint32_t)(call >> 32), nodes, plans);
  return hipGetLastError() == hipSuccess ? LG_OK : LG_ERR_HIP;
}

// the diffusion passes of one control step, enqueued by one call (see lgpolicy.h)
int lg_planner_diffuse(lg_ctx* ctx, float* mean, const float* sigma_nodes, const float* phi, int32_t num_main, int32_t R, int32_t K, int32_t H, int32_t A,
                       int32_t n_diffuse, float traj_diffuse_factor, float temperature, uint64_t seed, uint64_t call0, const int32_t* env_ids,
                       int32_t rollouts_per_main, float pos_drift, float* nodes, float* plans, float* rewards, float* weights, void* stream) {
  if (!ctx || !mean || !env_ids || !rewards || !weights || n_diffuse < 0 || rollouts_per_main != R) return LG_ERR_INVALID;
  float scale = 1.f;
  for (int pass = 0; pass < n_diffuse; ++pass) {
    int rc = lg_mppi_sample_plans(mean, sigma_nodes, scale, phi, num_main, R, K, H, A, seed, call0 + (uint64_t)pass, nodes, plans, stream);
    if (rc != LG_OK) return rc;
    rc = lg_rollout_batch(ctx, plans, H, env_ids, num_main * R, rollouts_per_main, pos_drift, rewards, stream);
    if (rc != LG_OK) return rc;
    rc = lg_mppi_update(rewards, nodes, num_main, R, H, K, A, temperature, mean, weights, stream);      // (the update reads `nodes`, not `mean`: in place)
    if (rc != LG_OK) return rc;
    scale *= traj_diffuse_factor;
  }
  return LG_OK;
}
