// Practical MFMA ceiling on this MI355X: back-to-back v_mfma_f32_32x32x16_bf16 on 4 accumulators, random-ish
// operands in registers, W waves per SIMD, every CU busy.  Reports sustained PFLOP/s and the in-kernel clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk) {
  const int lane = threadIdx.x & 63;
  bf16x8 a, b, c, d;
  for (int j = 0; j < 8; ++j) {
    a[j] = (__bf16)(0.37f * ((lane * 7 + j * 13) % 17) - 2.9f); b[j] = (__bf16)(0.11f * ((lane * 5 + j * 3) % 23) - 1.2f);
    c[j] = (__bf16)(0.23f * ((lane * 3 + j * 11) % 19) - 2.1f); d[j] = (__bf16)(0.19f * ((lane * 11 + j * 7) % 13) - 1.1f);
  }
  f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, d, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, b, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, d, acc3, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j] + acc2[j] + acc3[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 256 * 1024 * 4 * sizeof(float)); hipMalloc(&clk, 16);
  for (int wps = 1; wps <= 3; ++wps) {
    const int blocks = 256 * wps, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 100, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = 5.0 * blocks * 4.0 * iters * 24.0 * 32768.0;
    printf("waves/SIMD %d: %.1f TFLOP/s bf16 MFMA (%.2f ms), in-kernel clock %.2f GHz\n", wps, flops / (ms * 1e-3) / 1e12, ms,
           (double)h[0] / (double)h[1] * 0.1);
  }
  return 0;
}
