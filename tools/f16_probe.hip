// Hardware facts the two-term fp16 GEMM (csrc/gemm_f16.hip) relies on, checked on the MI355X itself:
//   1. v_cvt f32 -> f16 keeps fp16 SUBNORMALS (the low term of a small element is one);
//   2. v_mfma_f32_32x32x16_f16 multiplies subnormal fp16 inputs exactly (no flush to zero);
//   3. the f16 MFMA issues at the bf16 rate (sustained TFLOP/s on non-trivial operands, in-kernel clock).
// Build + run:  hipcc -O3 --offload-arch=gfx950 tools/f16_probe.hip -o /tmp/f16_probe && /tmp/f16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// out[0] = (float)(f16)tiny, out[1..] = one accumulator element of A (all `tiny`, converted in-kernel) x B (all `big`)
__global__ void denorm_kernel(const float* in, float* out) {
  const float tiny = in[0], big = in[1];
  const _Float16 ht = (_Float16)tiny, hb = (_Float16)big;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = ht; b[j] = hb; }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = (float)ht; out[1] = acc[0]; out[2] = tiny * big * 16.f; }
}

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, unsigned long long* clk) {
  const int lane = threadIdx.x & 63;
  f16x8 a, b, c, d; bf16x8 ab, bb, cb, db;
  for (int j = 0; j < 8; ++j) {
    const float va = 0.37f * ((lane * 7 + j * 13) % 17) - 2.9f, vb = 0.11f * ((lane * 5 + j * 3) % 23) - 1.2f;
    const float vc = 0.23f * ((lane * 3 + j * 11) % 19) - 2.1f, vd = 0.19f * ((lane * 11 + j * 7) % 13) - 1.1f;
    a[j] = (_Float16)va; b[j] = (_Float16)vb; c[j] = (_Float16)vc; d[j] = (_Float16)vd;
    ab[j] = (__bf16)va; bb[j] = (__bf16)vb; cb[j] = (__bf16)vc; db[j] = (__bf16)vd;
  }
  f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      if (KIND == 0) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, d, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(c, b, acc2, 0, 0, 0); acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(c, d, acc3, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, db, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cb, bb, acc2, 0, 0, 0); acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cb, db, acc3, 0, 0, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j] + acc2[j] + acc3[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
  float *in, *out; unsigned long long* clk;
  hipMalloc(&in, 64); hipMalloc(&out, 256 * 1024 * 4 * sizeof(float)); hipMalloc(&clk, 16);
  int bad = 0;
  const float cases[][2] = {{9.5367431640625e-07f /* 2^-20 */, 1024.f}, {5.9604644775390625e-08f /* 2^-24 */, 4096.f},
                            {3.0517578125e-05f /* 2^-15 */, 2.f}, {-1.1920928955078125e-07f /* -2^-23 */, 3.f}};
  for (auto& cs : cases) {
    hipMemcpy(in, cs, 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(denorm_kernel, dim3(1), dim3(64), 0, 0, in, out);
    float h[3]; hipMemcpy(h, out, 12, hipMemcpyDeviceToHost);
    const bool ok = h[0] == cs[0] && h[1] == h[2];
    printf("subnormal %.6e x %.1f: cvt -> %.6e, mfma sum16 -> %.6e (exact %.6e)  %s\n", cs[0], cs[1], h[0], h[1], h[2], ok ? "OK" : "FLUSHED/WRONG");
    bad += !ok;
  }
  for (int kind = 0; kind < 2; ++kind)
    for (int wps = 1; wps <= 2; ++wps) {
      const int blocks = 256 * wps, iters = 4000;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto launch = [&](int it) {
        if (kind == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, it, clk);
        else hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, it, clk);
      };
      launch(100); hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 5; ++r) launch(iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      const double flops = 5.0 * blocks * 4.0 * iters * 24.0 * 32768.0;
      printf("%s 32x32x16, %d waves/SIMD: %.1f TFLOP/s (%.2f ms), in-kernel clock %.2f GHz\n", kind == 0 ? "f16 " : "bf16", wps,
             flops / (ms * 1e-3) / 1e12, ms, (double)h[0] / (double)h[1] * 0.1);
    }
  return bad ? 1 : 0;
}
