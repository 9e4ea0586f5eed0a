// Diagnostic micro-benchmark: what slows a two-chain v_mfma_f32_16x16x4_f32 loop when operand loads are interleaved.
// MODE bit 0: one ds_read2 per MFMA pair, bit 1: one global_load per pair (3 = the MLP loop of lg_policy.hip), 0: MFMAs only
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(ACC, A, B) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
template <int MODE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ w, float* out, int iters) {
  __shared__ float lds[128 * 33 * 4];
  for (int i = threadIdx.x; i < 128 * 33 * 4; i += blockDim.x) lds[i] = 0.001f * i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  float b[8], a0[8], a1[8], nb[8], na0[8], na1[8];
  const float* a0p = lds + (lane & 15) * 4 + (lane >> 4);
  const float* wt = w + (size_t)((blockIdx.x * 8 + wv) & 1023) * 4096 + lane;
  for (int j = 0; j < 8; ++j) { b[j] = wt[j * 64]; a0[j] = a0p[j * 132]; a1[j] = a0p[j * 132 + 64]; }
  for (int i = 0; i < iters; ++i) {
    const float* wn = wt + ((i & 7) * 8) * 64; const float* an = a0p + ((i & 15) * 8) * 132;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      MFMA(acc0, a0[j], b[j]); MFMA(acc1, a1[j], b[j]);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE & 2) nb[j] = wn[j * 64]; else nb[j] = b[j];
      if (MODE & 1) { na0[j] = an[j * 132]; na1[j] = an[j * 132 + 64]; } else { na0[j] = a0[j]; na1[j] = a1[j]; }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { b[j] = nb[j]; a0[j] = na0[j]; a1[j] = na1[j]; }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  out[blockIdx.x * 512 + threadIdx.x] = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
}
// the same loop with 16-byte loads: one global_load_dwordx4 and two ds_read_b128 per FOUR MFMA pairs
template <int MODE>
__global__ __launch_bounds__(512) void kv(const float* __restrict__ w, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[128 * 33 * 4];
  for (int i = threadIdx.x; i < 128 * 33 * 4; i += blockDim.x) lds[i] = 0.001f * i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  float4 b[2], a0[2], a1[2], nb[2], na0[2], na1[2];
  const float4* a0p = reinterpret_cast<const float4*>(lds) + lane;               // [k/16][row half][lane] float4 = four k-steps of one lane
  const float4* wt = reinterpret_cast<const float4*>(w + (size_t)((blockIdx.x * 8 + wv) & 1023) * 4096) + lane;
  for (int j = 0; j < 2; ++j) { b[j] = wt[j * 64]; a0[j] = a0p[j * 128]; a1[j] = a0p[j * 128 + 64]; }
  for (int i = 0; i < iters; ++i) {
    const float4* wn = wt + ((i & 7) * 2) * 64; const float4* an = a0p + ((i & 15) * 2) * 128;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      MFMA(acc0, a0[j].x, b[j].x); MFMA(acc1, a1[j].x, b[j].x);
      MFMA(acc0, a0[j].y, b[j].y); MFMA(acc1, a1[j].y, b[j].y);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE & 2) nb[j] = wn[j * 64]; else nb[j] = b[j];
      if (MODE & 1) { na0[j] = an[j * 128]; na1[j] = an[j * 128 + 64]; } else { na0[j] = a0[j]; na1[j] = a1[j]; }
      __builtin_amdgcn_sched_barrier(0);
      MFMA(acc0, a0[j].z, b[j].z); MFMA(acc1, a1[j].z, b[j].z);
      MFMA(acc0, a0[j].w, b[j].w); MFMA(acc1, a1[j].w, b[j].w);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) { b[j] = nb[j]; a0[j] = na0[j]; a1[j] = na1[j]; }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  out[blockIdx.x * 512 + threadIdx.x] = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
}
template <int MODE>
void runv(float* w, float* d, int threads = 256) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 1000 * 256 / threads;
  hipLaunchKernelGGL(kv<MODE>, dim3(256), dim3(threads), 0, 0, w, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kv<MODE>, dim3(256), dim3(threads), 0, 0, w, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 16 * (threads / 256);
  printf("16-byte loads, mode %d, %d waves per CU: %.1f us, %.2f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz)\n", MODE, threads / 64, ms * 1e3, ms * 1e6 / n, ms * 1e6 / n * 2.4);
}
template <int MODE>
void run(float* w, float* d, int threads = 256) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 1000 * 256 / threads;
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, w, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, w, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 16 * (threads / 256);
  printf("mode %d, %d waves per CU: %.1f us, %.2f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz)\n", MODE, threads / 64, ms * 1e3, ms * 1e6 / n, ms * 1e6 / n * 2.4);
}
int main() {
  float *w, *d; hipMalloc(&w, 4 * 1024 * 4096); hipMemset(w, 0, 4 * 1024 * 4096); hipMalloc(&d, 4 * 256 * 512);
  run<0>(w, d); run<1>(w, d); run<2>(w, d); run<3>(w, d);
  run<0>(w, d, 512); run<1>(w, d, 512); run<2>(w, d, 512); run<3>(w, d, 512);
  runv<0>(w, d); runv<1>(w, d); runv<2>(w, d); runv<3>(w, d); runv<3>(w, d, 512);
  return 0;
}
