// Shared device helpers for the NPVP gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NPVP_OK 0
#define NPVP_ERR_ARG (-1)
#define NPVP_ERR_LAUNCH (-2)
#define NPVP_ERR_WORKSPACE (-3)

// host side: record the last error string (thread local), see api.hip
extern "C" void npvp_set_error(const char* msg);

// diagnostics: every kernel this library launches goes through NPVP_LAUNCH, which counts it (npvp_launch_count, api.hip; one
// relaxed atomic add per launch).  bench.py reads the counter around a step to put `launches_per_step` into its record.
extern "C" long long npvp_launch_count(void);
namespace npvp { extern long long g_launches; }
#define NPVP_LAUNCH(...)                                                    \
  do {                                                                      \
    __atomic_fetch_add(&npvp::g_launches, 1ll, __ATOMIC_RELAXED);           \
    hipLaunchKernelGGL(__VA_ARGS__);                                        \
  } while (0)

#define NPVP_CHECK_ARG(cond, msg)                    \
  do {                                               \
    if (!(cond)) {                                   \
      npvp_set_error(msg);                           \
      return NPVP_ERR_ARG;                           \
    }                                                \
  } while (0)

#define NPVP_CHECK_LAUNCH()                          \
  do {                                               \
    hipError_t e__ = hipGetLastError();              \
    if (e__ != hipSuccess) {                         \
      npvp_set_error(hipGetErrorString(e__));        \
      return NPVP_ERR_LAUNCH;                        \
    }                                                \
  } while (0)

namespace npvp {

constexpr int WAVE = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- amax slots (precision-6 GEMM operands, include/npvp_hip.h): the tensor's bound is the maximum of 32 words that lie
// 64 bytes apart (a slot is 2 KB: each word in its own memory sector).  Producers raise single words with an integer atomic
// max on the bit pattern of a non-negative float (order independent: deterministic); consumers read the 32 words with one
// strided load.  Same-address atomics - and L1-bypassing loads of one line - queue at ~10 ns each, and a large launch has
// 10^4..10^5 waves, hence: 32 words, one sector each; look before raising; one commit per BLOCK where every thread reaches
// the end of the kernel (amax_slot_commit_block), per wave otherwise.
constexpr int AMAX_WORDS = 32, AMAX_STRIDE = 16;
__device__ __forceinline__ float amax_slot_read(const float* slot) {
  const float v = wave_max(slot[(threadIdx.x & (AMAX_WORDS - 1)) * AMAX_STRIDE]);
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
// Raising a word: an atomic only when the bound is above what the word held when the wave / block STARTED (amax_peek_*: an
// L1-bypassing load issued first thing, its latency hidden under the kernel's work; a load at the end of the kernel would
// sit in the memory queue behind the kernel's own streaming stores).  After the first resident set of a launch almost no
// bound is above the word, so a launch costs about one atomic per resident wave / block.  A stale peek costs an unnecessary
// atomic, never a wrong bound.  A null slot peeks as "already at the maximum".
__device__ __forceinline__ unsigned int amax_peek(const float* slot, unsigned int idx) {
  if (!slot) return 0xffffffffu;
  return __hip_atomic_load(reinterpret_cast<const unsigned int*>(slot) + (idx & (AMAX_WORDS - 1)) * AMAX_STRIDE, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned int amax_peek_wave(const float* slot) { return amax_peek(slot, blockIdx.x * 4u + (threadIdx.x >> 6)); }
__device__ __forceinline__ unsigned int amax_peek_block(const float* slot) { return amax_peek(slot, blockIdx.x); }
__device__ __forceinline__ void amax_word_raise(float* slot, unsigned int idx, float m, unsigned int peeked) {
  const unsigned int bits = __float_as_uint(m);
  if (bits > peeked) atomicMax(reinterpret_cast<unsigned int*>(slot) + (idx & (AMAX_WORDS - 1)) * AMAX_STRIDE, bits);
}
__device__ __forceinline__ void amax_slot_commit(float* slot, float m, unsigned int peeked) {          // per wave (wave-uniform call)
  if (!slot) return;
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) amax_word_raise(slot, blockIdx.x * 4u + (threadIdx.x >> 6), m, peeked);
}
// per block: EVERY thread of the block must call it; red = blockDim.x / 64 floats of LDS that nobody else is using
__device__ __forceinline__ void amax_slot_commit_block(float* slot, float m, float* red, unsigned int peeked) {
  if (!slot) return;
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (unsigned int i = 1; i < (blockDim.x >> 6); ++i) m = fmaxf(m, red[i]);
    amax_word_raise(slot, blockIdx.x, m, peeked);
  }
}
// the power of two s with amax * s in [2^14, 2^15) (1 for a zero / tiny / non-finite bound)
__host__ __device__ __forceinline__ float amax_scale(float amax) {
  unsigned int u;
  __builtin_memcpy(&u, &amax, 4);
  const unsigned int E = (u >> 23) & 0xffu;
  if (E < 16u || E == 255u) return 1.f;
  u = (268u - E) << 23;
  float s;
  __builtin_memcpy(&s, &u, 4);
  return s;
}

// 1 / s for a power of two s with exponent field 1..253 (what amax_scale returns): no division sequence
__host__ __device__ __forceinline__ float pow2_recip(float s) {
  unsigned int u;
  __builtin_memcpy(&u, &s, 4);
  u = 0x7f000000u - u;
  float r;
  __builtin_memcpy(&r, &u, 4);
  return r;
}

// running bound of four stored values
__device__ __forceinline__ float amax4(float m, const float4& v) {
  return fmaxf(fmaxf(fmaxf(m, fabsf(v.x)), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
}

// acc += a * b as exactly one v_fmac_f32 (the optimiser can neither pair it into v_pk_fma_f32 nor split it).
__device__ __forceinline__ void fmac_scalar(float& acc, float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
#else
  acc = fmaf(a, b, acc);
#endif
}

// Block-wide sum for blockDim.x = NW*64 threads; `red` is NW floats of LDS.
// Every thread gets the total.  Contains two barriers.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();                      // protect `red` from a previous use
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += red[i];
  return t;
}

// mean / rstd of a frame from the J partial (mean_j, M2_j) pairs its producer left (each over nb values): exactly the arithmetic
// of frame_stats_finalize_kernel, evaluated by the CONSUMER (wave-uniform loads) instead of a launch of its own
__device__ __forceinline__ void frame_stats_merge(const float* __restrict__ part, long long f, int J, float nb, float eps, float& mu,
                                                  float& rs) {
  float m = 0.f;
  for (int j = 0; j < J; ++j) m += part[(f * J + j) * 2];
  m /= J;
  float m2 = 0.f;
  for (int j = 0; j < J; ++j) { const float d = part[(f * J + j) * 2] - m; m2 += part[(f * J + j) * 2 + 1] + nb * d * d; }
  mu = m;
  rs = rsqrtf(m2 / (nb * J) + eps);
}

// erf-GELU (nn.GELU default) and its derivative.  libm's erff costs ~45 VALU instructions with divergent branches and
// made the frame-LN kernels VALU bound (frame-LN backward statistics: 94 us for 336 MB = 3.6 TB/s).  Phi(x) is
// evaluated branch-free from ONE exponential shared with the density:
//     E = exp(-x^2/2),  t = 1/(1 + p|x|/sqrt2),  1 - erf(|x|/sqrt2) = (a1 t + ... + a5 t^5) E      (A&S 7.1.26)
//     Phi(x) = x >= 0 ? 1 - q/2 : q/2,   gelu = x Phi,   gelu' = Phi + x E / sqrt(2 pi)
// measured in fp32: max abs error of Phi 3.0e-7, of gelu 4.2e-7 over [-12,12] (torch.nn.functional.gelu in fp32: 1.2e-6);
// the negative branch has no cancellation.
__device__ __forceinline__ float gelu_phi(float x, float& E) {
  const float au = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, au, 1.0f));
  E = __expf(-0.5f * x * x);
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float hq = 0.5f * p * t * E;
  return x >= 0.f ? 1.0f - hq : hq;
}
__device__ __forceinline__ float gelu_f(float x) {
  float E;
  return x * gelu_phi(x, E);
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float E;
  const float phi = gelu_phi(x, E);
  return fmaf(x * 0.39894228040143267794f, E, phi);
}

// Counter-based RNG for dropout masks: stateless, so backward replays the mask of
// forward from (seed, salt, element index).  `seed` lives in device memory so that
// a captured HIP graph sees a fresh value on every replay.
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu;
  h ^= h >> 13; h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}
__host__ __device__ __forceinline__ uint32_t rng_u32(uint64_t seed, uint32_t salt, uint64_t idx) {
  // Two keys from (seed, salt) - wave-uniform values: the compiler keeps them on the scalar unit and hoists them out of loops -
  // then TWO finalizer rounds per element with a key entering before each (until round 4: three rounds, six 32-bit multiplies
  // per element - quarter-rate instructions; in the fused MlpDWBN backward, which is VALU bound, the mask was a third of the
  // kernel's issue time).  Same battery as the three-round form on sequential and strided counters (chi-square of the top /
  // middle / low byte 0.83 .. 1.21 per degree of freedom, keep rates within 2 sigma, cross-site / cross-step and lagged
  // correlations of the masks within 4 sigma at p = 0.1 and 0.5; tests/test_hip_dropout.py::test_mask_statistics).
  const uint32_t k0 = mix32((uint32_t)seed ^ mix32(salt * 0x9e3779b9u + 0x7f4a7c15u));
  const uint32_t k1 = mix32((uint32_t)(seed >> 32) + mix32(salt ^ 0x85ebca6bu));
  const uint32_t h = mix32((uint32_t)idx ^ k0);
  return mix32(h + k1 + (uint32_t)(idx >> 32) * 0x9e3779b9u);
}
// keep-scale for inverted dropout: 1/(1-p) if kept, 0 if dropped
__device__ __forceinline__ float drop_scale(uint64_t seed, uint32_t salt, uint64_t idx, uint32_t thresh, float inv_keep) {
  return rng_u32(seed, salt, idx) >= thresh ? inv_keep : 0.f;
}
__host__ __device__ __forceinline__ uint32_t drop_threshold(float p) {
  // P(u32 < thresh) = p
  double t = (double)p * 4294967296.0;
  if (t < 0) t = 0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (uint32_t)t;
}

// Where a dropout / drop-path mask is keyed.
//   mode 0: per element          key = row * ncols + col       (nn.Dropout)
//   mode 1: per row group        key = (row / g1) % g2         (DropPath, ref/models/VidHRFormer.py:513-525:
//           per sample g1 = T*P, g2 = N; the enc-dec site drops whole time-steps (:239): g1 = P, g2 = T)
struct DropSpec {
  unsigned int thresh; float inv_keep; unsigned int salt; int mode; int g1; int g2;
  unsigned int m1, m2; int s1, s2;      // magic multipliers for row / g1 and (row / g1) / g2 (rows < 2^31), see div_magic
};
// n / d for n < 2^31 as (n * m) >> s with m = ceil(2^s / d), s = 31 + ceil(log2 d): exact (m d - 2^s < d <= 2^(s-31), so the
// excess n (m d - 2^s) / (d 2^s) stays below 1 / d).  The group of a token row used to be two 64-bit divisions per call - ~250
// vector instructions per row in a GEMM epilogue whose whole K loop has 2 800.
__host__ __device__ __forceinline__ void div_magic(unsigned int d, unsigned int& m, int& s) {
  int l = 0;
  while (l < 31 && (1u << l) < d) ++l;
  s = 31 + l;
  m = (unsigned int)(((1ull << s) + d - 1ull) / d);
}
__device__ __forceinline__ unsigned int div_by_magic(unsigned int n, unsigned int m, int s) {
  return (unsigned int)(((unsigned long long)n * m) >> s);
}
__device__ __forceinline__ unsigned int drop_group(const DropSpec& d, long long row) {
  const unsigned int q1 = div_by_magic((unsigned int)row, d.m1, d.s1);
  return q1 - div_by_magic(q1, d.m2, d.s2) * (unsigned int)d.g2;
}
__device__ __forceinline__ float drop_spec_scale(const DropSpec& d, uint64_t seed, long long row, int col, int ncols) {
  const uint64_t key = d.mode == 0 ? (uint64_t)row * (uint64_t)ncols + (uint64_t)col : (uint64_t)drop_group(d, row);
  return rng_u32(seed, d.salt, key) >= d.thresh ? d.inv_keep : 0.f;
}
inline DropSpec make_drop_spec(float p, unsigned int salt, int mode, int g1, int g2) {
  DropSpec d;
  d.thresh = p > 0.f ? drop_threshold(p) : 0u;
  d.inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  d.salt = salt; d.mode = mode; d.g1 = g1 > 0 ? g1 : 1; d.g2 = g2 > 0 ? g2 : 1;
  div_magic((unsigned int)d.g1, d.m1, d.s1);
  div_magic((unsigned int)d.g2, d.m2, d.s2);
  return d;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// out[c] = sum_b in[b*stride + c], c < ncols (defined in norm.hip)
// out[c] (+)= sum_b in[b*stride + c]; columns >= split go to out_b[c - split] (two parameter gradients from one
// partial buffer in one launch); accum = add to what the outputs already hold (gradient accumulation in place).
int launch_sum_rows(const float* in, float* out, int nb, int stride, int ncols, hipStream_t stream, int accum = 0,
                    float* out_b = nullptr, int split = 0);

}  // namespace npvp
