// The two-instruction split of csrc/gemm_f16.hip (v_fma_mixlo_f16 / v_fma_mixhi_f16) against the plain form
// "scale, convert to fp16, convert back, subtract, convert" - bit for bit, on the MI355X itself, over random words of every
// exponent (NaNs excluded: payloads may differ), subnormal inputs, zeros, infinities and the scales amax_scale can return.
// Zeros of opposite sign are counted separately (see check_kernel).
// Build + run:  hipcc -O3 --offload-arch=gfx950 tools/f16_split_check.hip -o /tmp/f16_split_check && /tmp/f16_split_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_plain(const f32x4 v, const float s, f16x4& hi, f16x4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x = v[i] * s;
    const _Float16 h = (_Float16)x;
    hi[i] = h;
    lo[i] = (_Float16)(x - (float)h);
  }
}
__device__ __forceinline__ void split_mix(const f32x4 v, const float s, f16x4& hi, f16x4& lo) {
  unsigned int h01, h23, l01, l23;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h01) : "v"(v[0]), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h23) : "v"(v[2]), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h01) : "v"(v[1]), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h23) : "v"(v[3]), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l01) : "v"(v[0]), "v"(s), "v"(h01));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l23) : "v"(v[2]), "v"(s), "v"(h23));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l01) : "v"(v[1]), "v"(s), "v"(h01));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l23) : "v"(v[3]), "v"(s), "v"(h23));
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 hp = {h01, h23}, lp = {l01, l23};
  hi = __builtin_bit_cast(f16x4, hp);
  lo = __builtin_bit_cast(f16x4, lp);
}

__global__ void check_kernel(const f32x4* x, const float* scales, int nscale, long long n, unsigned long long* bad, uint2* first) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f32x4 v = x[i];
  for (int k = 0; k < nscale; ++k) {
    f16x4 h0, l0, h1, l1;
    split_plain(v, scales[k], h0, l0);
    split_mix(v, scales[k], h1, l1);
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    const u16x4 a = __builtin_bit_cast(u16x4, h0), b = __builtin_bit_cast(u16x4, h1);
    const u16x4 c = __builtin_bit_cast(u16x4, l0), d = __builtin_bit_cast(u16x4, l1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // the GEMM's domain: the scale puts the tensor's bound into [2^14, 2^15) (times a DropPath factor 1 / (1 - p)), so |x s| < 2^16;
      // beyond it the plain form turns inf - inf into NaN where the FMA keeps an infinity - counted apart, never reached
      if (!(fabsf(v[e] * scales[k]) < 65536.f)) { atomicAdd(bad + 2, 1ull); continue; }
      // (a zero may differ in SIGN: where x s underflows fp32, "x s - hi" is (-0) - (-0) = +0 in the plain form and the FMA's exact
      //  tiny negative rounds to -0; a zero term contributes the same to every product sum whatever its sign)
      const bool zh = ((a[e] | b[e]) & 0x7fffu) == 0, zl = ((c[e] | d[e]) & 0x7fffu) == 0;
      if ((a[e] != b[e] && !zh) || (c[e] != d[e] && !zl)) {
        if (atomicAdd(bad, 1ull) == 0ull) *first = make_uint2((unsigned int)i, (unsigned int)k);
      } else if (a[e] != b[e] || c[e] != d[e]) atomicAdd(bad + 1, 1ull);
    }
  }
}

int main() {
  const long long n = 1ll << 22;                        // 16 M words x 40 scales
  std::vector<unsigned int> w(4 * n);
  unsigned long long st = 0x9e3779b97f4a7c15ull;
  for (auto& u : w) {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    u = (unsigned int)(st >> 32);
    if (((u >> 23) & 0xffu) == 0xffu && (u & 0x7fffffu)) u &= 0xff800000u;          // NaN -> infinity
  }
  const unsigned int edge[] = {0u, 0x80000000u, 1u, 0x007fffffu, 0x00800000u, 0x7f800000u, 0xff800000u, 0x7f7fffffu, 0x477fe000u,
                               0x477ff000u, 0x477fefffu, 0x38800000u, 0x387fffffu, 0x33800000u, 0x337fffffu, 0x3f800000u};
  for (size_t i = 0; i < sizeof(edge) / 4; ++i) w[i] = edge[i];
  std::vector<float> sc;
  for (int e = -126; e <= 127; e += 7) { unsigned int u = (unsigned int)(e + 127) << 23; float f; memcpy(&f, &u, 4); sc.push_back(f); }
  sc.push_back(1.f); sc.push_back(16384.f); sc.push_back(1.f / 16384.f);
  f32x4* dx; float* ds; unsigned long long* dbad; uint2* dfirst;
  hipMalloc(&dx, 16 * n); hipMalloc(&ds, sc.size() * 4); hipMalloc(&dbad, 24); hipMalloc(&dfirst, 8);
  hipMemcpy(dx, w.data(), 16 * n, hipMemcpyHostToDevice);
  hipMemcpy(ds, sc.data(), sc.size() * 4, hipMemcpyHostToDevice);
  hipMemset(dbad, 0, 24); hipMemset(dfirst, 0, 8);
  check_kernel<<<(unsigned int)((n + 255) / 256), 256>>>(dx, ds, (int)sc.size(), n, dbad, dfirst);
  unsigned long long bad2[3] = {0, 0, 0}; uint2 first;
  if (hipMemcpy(bad2, dbad, 24, hipMemcpyDeviceToHost) != hipSuccess) { printf("launch failed\n"); return 2; }
  const unsigned long long bad = bad2[0];
  hipMemcpy(&first, dfirst, 8, hipMemcpyDeviceToHost);
  printf("%lld float4 words x %zu scales: %llu elements in the domain |x s| < 2^16 (%llu outside), %llu mismatches, %llu zeros of opposite sign",
         n, sc.size(), 4ull * n * sc.size() - bad2[2], bad2[2], bad, bad2[1]);
  if (bad) printf(" (first: word %u = %08x %08x %08x %08x, scale %g)", first.x, w[4 * first.x], w[4 * first.x + 1], w[4 * first.x + 2], w[4 * first.x + 3], sc[first.y]);
  printf("\n");
  return bad ? 1 : 0;
}
