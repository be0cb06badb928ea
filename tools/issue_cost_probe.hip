// What does an instruction COST next to the matrix instructions on gfx950?  A loop of 24 v_mfma_f32_32x32x16_f16 per iteration (8
// independent accumulators x 3, the K-step of csrc/gemm_f16.hip) at 2 waves per SIMD, and the same loop with X instructions of one
// kind added per iteration: cost = (cycles with - cycles without) / X, in shader cycles per wave-instruction (s_memtime).  If the
// added instructions hid under the MFMAs the cost would be ~0.
// Build + run:  hipcc -O3 --offload-arch=gfx950 tools/issue_cost_probe.hip -o /tmp/icp && /tmp/icp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { NONE = 0, VALU_FMA, VALU_CVT, DS_READ128, DS_WRITE64, GLD_VADDR, GLD_SADDR, LDSDMA_SADDR, GST_VADDR, SALU, LDSDMA_VADDR, GLD1_VADDR, GLD2_VADDR,
       DS_READ64, DS_WRITE2_64, GLD_VADDR_HALF };

template <int KIND, int X>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ src, float* __restrict__ dst, float* out, int iters,
                                                unsigned long long* clk) {
  __shared__ __attribute__((aligned(16))) char lds[32768];
  const int lane = threadIdx.x & 63, t = threadIdx.x;
  f16x8 a, b, c, d;
  for (int j = 0; j < 8; ++j) {
    a[j] = (_Float16)(0.37f * ((lane * 7 + j * 13) % 17) - 2.9f); b[j] = (_Float16)(0.11f * ((lane * 5 + j * 3) % 23) - 1.2f);
    c[j] = (_Float16)(0.23f * ((lane * 3 + j * 11) % 19) - 2.1f); d[j] = (_Float16)(0.19f * ((lane * 11 + j * 7) % 13) - 1.1f);
  }
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int g = 0; g < 16; ++g) acc[i][g] = 0.f;
  float v0 = lane * 0.5f, v1 = 1.0001f, v2 = 0.25f, v3 = 3.f;
  f32x4 ld[8];
  for (int i = 0; i < 8; ++i) ld[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // every workgroup reads / writes its own 64 KB window (L2 resident after the first iteration)
  const char* gsrc = reinterpret_cast<const char*>(src) + (size_t)blockIdx.x * 65536;
  char* gdst = reinterpret_cast<char*>(dst) + (size_t)blockIdx.x * 65536;
  const unsigned int voff = (unsigned int)t * 16;
  const unsigned int lbase = (unsigned int)(size_t)((__attribute__((address_space(3))) char*)lds);
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int x = 0; x < X; ++x) {
      if (KIND == VALU_FMA) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x & 1 ? v0 : v3) : "v"(v1), "v"(v2)); }
      if (KIND == VALU_CVT) { asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(x & 1 ? v0 : v3) : "v"(v1), "v"(v2)); }
      if (KIND == SALU) { asm volatile("s_add_u32 s20, s20, 1" ::: "s20"); }
      if (KIND == DS_READ128) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[x & 7]) : "v"(voff), "n"((x & 3) * 4096) : "memory"); }
      if (KIND == DS_WRITE64) { asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(voff), "v"(*reinterpret_cast<double*>(&ld[x & 7])), "n"((x & 3) * 4096) : "memory"); }
      if (KIND == GLD_VADDR) { const char* p = gsrc + voff + (x & 7) * 4096; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[x & 7]) : "v"(p) : "memory"); }
      if (KIND == GLD_SADDR) { const char* p = gsrc + (x & 7) * 4096; asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ld[x & 7]) : "v"(voff), "s"(p) : "memory"); }
      if (KIND == LDSDMA_SADDR) {
        const char* p = gsrc + (x & 7) * 4096; const unsigned int m0 = lbase + wave * 1024 + (x & 3) * 4096;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0), "v"((unsigned int)(lane * 16)), "s"(p) : "memory", "m0");
      }
      if (KIND == LDSDMA_VADDR) {
        const char* p = gsrc + lane * 16 + (x & 7) * 4096; const unsigned int m0 = lbase + wave * 1024 + (x & 3) * 4096;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m0), "v"(p) : "memory", "m0");
      }
      if (KIND == GLD1_VADDR) { const char* p = gsrc + voff + (x & 7) * 4096; asm volatile("global_load_dword %0, %1, off" : "=v"(ld[x & 7][0]) : "v"(p) : "memory"); }
      if (KIND == GLD2_VADDR) { const char* p = gsrc + voff + (x & 7) * 4096; asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(*reinterpret_cast<double*>(&ld[x & 7])) : "v"(p) : "memory"); }
      if (KIND == GLD_VADDR_HALF) { const char* p = gsrc + voff + (x & 7) * 4096;
        asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, 0xffffffff\n\tglobal_load_dwordx4 %0, %1, off\n\ts_mov_b64 exec, s[20:21]" : "=v"(ld[x & 7]) : "v"(p) : "memory", "s20", "s21"); }
      if (KIND == DS_READ64) { asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(*reinterpret_cast<double*>(&ld[x & 7])) : "v"(voff), "n"((x & 3) * 4096) : "memory"); }
      if (KIND == DS_WRITE2_64) { asm volatile("ds_write2st64_b64 %0, %1, %2 offset0:%3 offset1:%4" :: "v"(voff), "v"(*reinterpret_cast<double*>(&ld[x & 7])), "v"(*reinterpret_cast<double*>(&ld[(x + 1) & 7])), "n"(x & 3), "n"(8 + (x & 3)) : "memory"); }
      if (KIND == GST_VADDR) { char* p = gdst + voff + (x & 7) * 4096; asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(ld[x & 7]) : "memory"); }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, d, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c, b, acc[2], 0, 0, 0); acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c, d, acc[3], 0, 0, 0);
      acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc[4], 0, 0, 0); acc[5] = __builtin_amdgcn_mfma_f32_32x32x16_f16(d, a, acc[5], 0, 0, 0);
      acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, c, acc[6], 0, 0, 0); acc[7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(d, c, acc[7], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = v0 + v3;
  for (int i = 0; i < 8; ++i) { for (int g = 0; g < 16; ++g) s += acc[i][g]; s += ld[i][0] + ld[i][3]; }
  out[blockIdx.x * 256 + t] = s;
  if (t == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int KIND, int X>
static double run(const char* name, float* src, float* dst, float* out, unsigned long long* clk, double base) {
  const int iters = 2000;
  hipLaunchKernelGGL((probe<KIND, X>), dim3(512), dim3(256), 0, 0, src, dst, out, iters, clk);
  hipLaunchKernelGGL((probe<KIND, X>), dim3(512), dim3(256), 0, 0, src, dst, out, iters, clk);
  hipDeviceSynchronize();
  unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  const double per_iter = (double)c / iters;                       // s_memtime ticks (100 MHz on gfx9: convert with the ratio below)
  if (base > 0) printf("%-44s %9.1f ticks/iter   %+7.2f ticks per added instruction (x %d)\n", name, per_iter, (per_iter - base) / X, X);
  else printf("%-44s %9.1f ticks/iter\n", name, per_iter);
  return per_iter;
}

int main() {
  float *src, *dst, *out; unsigned long long* clk;
  hipMalloc(&src, 512 * 65536); hipMalloc(&dst, 512 * 65536); hipMalloc(&out, 512 * 256 * 4); hipMalloc(&clk, 16);
  hipMemset(src, 0, 512 * 65536);
  printf("24 v_mfma_f32_32x32x16_f16 per iteration, 512 workgroups x 4 waves (2 waves per SIMD); ticks = s_memtime\n");
  const double b = run<NONE, 1>("MFMA only", src, dst, out, clk, 0);
  printf("(24 MFMAs x 2 waves per SIMD = 48 x 32 = 1536 shader cycles per iteration if the pipe is full: 1 tick = %.2f cycles)\n", 1536.0 / b);
  run<VALU_FMA, 32>("+ v_fma_f32", src, dst, out, clk, b);
  run<VALU_CVT, 32>("+ v_cvt_pk_f16_f32", src, dst, out, clk, b);
    run<DS_READ128, 12>("+ ds_read_b128", src, dst, out, clk, b);
  run<DS_WRITE64, 8>("+ ds_write_b64", src, dst, out, clk, b);
  run<GLD_VADDR, 6>("+ global_load_dwordx4 (64-bit vaddr)", src, dst, out, clk, b);
  run<GLD_SADDR, 6>("+ global_load_dwordx4 (saddr + voffset)", src, dst, out, clk, b);
  run<LDSDMA_SADDR, 4>("+ global_load_lds_dwordx4 (saddr + voffset)", src, dst, out, clk, b);
  run<GST_VADDR, 4>("+ global_store_dwordx4 (64-bit vaddr)", src, dst, out, clk, b);
  run<LDSDMA_VADDR, 4>("+ global_load_lds_dwordx4 (64-bit vaddr)", src, dst, out, clk, b);
  run<GLD1_VADDR, 6>("+ global_load_dword (64-bit vaddr)", src, dst, out, clk, b);
  run<GLD2_VADDR, 6>("+ global_load_dwordx2 (64-bit vaddr)", src, dst, out, clk, b);
  run<GLD_VADDR_HALF, 6>("+ global_load_dwordx4, 32 lanes active", src, dst, out, clk, b);
  run<DS_READ64, 12>("+ ds_read_b64", src, dst, out, clk, b);
  run<DS_WRITE2_64, 4>("+ ds_write2st64_b64", src, dst, out, clk, b);
  run<GLD_VADDR, 12>("+ global_load_dwordx4 (64-bit vaddr) x 12", src, dst, out, clk, b);
  run<LDSDMA_SADDR, 8>("+ global_load_lds_dwordx4 (saddr) x 8", src, dst, out, clk, b);
  return 0;
}
