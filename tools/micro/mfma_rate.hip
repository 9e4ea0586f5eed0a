// Diagnostic micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 with 1, 2 and 4 independent accumulator chains per wave,
// 4 waves per workgroup, one workgroup per CU.   hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float av = a + threadIdx.x, bv = b;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int c = 0; c < CH; ++c) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[c]) : "v"(av), "v"(bv));
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float s = 0;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH>
void run(const char* name, int wgs) {
  float* d; hipMalloc(&d, 4 * 256 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  hipLaunchKernelGGL(k<CH>, dim3(wgs), dim3(256), 0, 0, d, iters, 1.f, 0.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<CH>, dim3(wgs), dim3(256), 0, 0, d, iters, 1.f, 0.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 8 * CH;
  printf("%s chains %d  wgs %d: %.1f us, %.2f ns per MFMA per wave (%.1f cycles at 2.4 GHz), %.1f TFLOP/s\n", name, CH, wgs, ms * 1e3, ms * 1e6 / n,
         ms * 1e6 / n * 2.4, n * 4 * wgs * 2048 / (ms * 1e-3) / 1e12);
  hipFree(d);
}
int main() {
  run<1>("16x16x4f32", 256); run<2>("16x16x4f32", 256); run<4>("16x16x4f32", 256); run<2>("16x16x4f32", 64); run<2>("16x16x4f32", 512);
  return 0;
}
