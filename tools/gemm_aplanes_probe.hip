// MEASUREMENT, not part of the library: what would pre-split fp16 A planes give the dominant GEMM?  
//
// The product kernel (csrc/gemm_f16.hip, compiled INTO this program by the #include below) stages the fp32 A operand through
// registers: 2 global loads per thread and K-step, the two-term split, 2 paired LDS writes.  The variant here takes A the way B
// travels: as two scaled fp16 planes in the LDS image's own layout [term][K/8][M][8 over k], copied HBM -> LDS by LDS-DMA - no
// registers, no split, no LDS writes in the loop (and no row guard, no row-group mask: a forward GEMM of a tensor whose producer
// wrote the planes).  The planes are made here by the library's weight splitter applied to A, with A's own amax slot, so both
// kernels multiply the SAME halves: the outputs must be bit-identical, and are checked.
// Timed: the five layer shapes of the path at M token rows (default 114 688 = c2's decoder), two and three LDS stages, in interleaved
// rounds after a long warm-up (see main: timing the kernels one after the other misread them by 10 % either way).
//
// Build + run (on the GPU box, from the repo root):
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Inpvp_amd/csrc -Iinclude tools/gemm_aplanes_probe.hip -Lnpvp_amd -l:libnpvp_hip.so \
//         -Wl,-rpath,$PWD/npvp_amd -o /tmp/gemm_aplanes_probe && /tmp/gemm_aplanes_probe [rows]
#include "../npvp_amd/csrc/gemm_f16.hip"
#include <cstdio>
#include <vector>
extern "C" const char* npvp_last_error(void);

namespace npvp {

template <int TM, int TN, int WM, int WN, int NST>
__device__ __forceinline__ void gemm_f16_aplanes_body(const GemmParams& p, const char* a_pre, long long a_plane_bytes, char* lds,
                                                      const int bid, const int nwg) {
  constexpr int NW = WM * WN;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int KGS_A = BM * 16 + 64, A_PLANE = 2 * KGS_A, A_BYTES = 2 * A_PLANE;
  constexpr int KGS_B = BN * 16, B_PLANE = 2 * KGS_B, B_BYTES = 2 * B_PLANE;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int CPS_B = BN / 64, NCH_B = 4 * CPS_B, CPW_B = NCH_B / NW;
  constexpr int CPS_A = BM / 64, NCH_A = 4 * CPS_A, CPW_A = NCH_A / NW;
  static_assert(NCH_A % NW == 0 && NCH_B % NW == 0 && (NST == 2 || NST == 3), "chunk maps");

  int tile_m, tile_n;
  tile_of_block_unsplit(p, nwg, bid, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
  const int nk = p.K >> 4;
  const float sa = amax_scale(amax_slot_read(p.a_amax));

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

  const char* b_base = reinterpret_cast<const char*>(p.b_pre);
  unsigned int b_off[CPW_B], a_off[CPW_A];
  int b_dst[CPW_B], a_dst[CPW_A];
#pragma unroll
  for (int i = 0; i < CPW_B; ++i) {
    const int c = wave + NW * i;
    const int slab = c / CPS_B, part = c - slab * CPS_B, s = slab >> 1, kg = slab & 1;
    const int col = min(n0 + part * 64 + lane, p.N - 1);
    b_off[i] = (unsigned int)(((long long)s * (p.b_pre_plane >> 3) + (long long)kg * p.N + col) * 16);
    b_dst[i] = A_BYTES + s * B_PLANE + kg * KGS_B + part * 1024;
  }
#pragma unroll
  for (int i = 0; i < CPW_A; ++i) {
    const int c = wave + NW * i;
    const int slab = c / CPS_A, part = c - slab * CPS_A, s = slab >> 1, kg = slab & 1;
    const int row = min(m0 + part * 64 + lane, p.M - 1);
    a_off[i] = (unsigned int)((long long)s * a_plane_bytes + ((long long)kg * p.M + row) * 16);
    a_dst[i] = s * A_PLANE + kg * KGS_A + part * 1024;
  }
  const long long b_step = 32ll * p.N, a_step = 32ll * p.M;       // bytes per K-step (two k-groups of 16 bytes per row / column)
  const unsigned int lds_u32 = (unsigned int)(size_t)((lptr_t)lds);
  const int fa_off = h * KGS_A + (wm * TM * 32 + r) * 16;
  const int fb_off = A_BYTES + h * KGS_B + (wn * TN * 32 + r) * 16;

#define PROBE_DMA(ST, KT)                                                                                  \
  { const int kt_ = min((KT), nk - 1);                                                                     \
    const char* bb_ = b_base + (long long)kt_ * b_step; const char* ab_ = a_pre + (long long)kt_ * a_step; \
    _Pragma("unroll") for (int i_ = 0; i_ < CPW_B; ++i_) {                                                 \
      const unsigned int m0_ = lds_u32 + (unsigned int)((ST) - lds) + (unsigned int)b_dst[i_];             \
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0_), "v"(b_off[i_]), "s"(bb_) : "memory", "m0"); } \
    _Pragma("unroll") for (int i_ = 0; i_ < CPW_A; ++i_) {                                                 \
      const unsigned int m0_ = lds_u32 + (unsigned int)((ST) - lds) + (unsigned int)a_dst[i_];             \
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0_), "v"(a_off[i_]), "s"(ab_) : "memory", "m0"); } }

  PROBE_DMA(lds, 0)
  if constexpr (NST == 3) {
    PROBE_DMA(lds + STAGE, 1)
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(CPW_A + CPW_B) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();

  for (int kt = 0; kt < nk; ++kt) {
    const char* st_ = lds + (kt % NST) * STAGE;
    PROBE_DMA(lds + ((kt + NST - 1) % NST) * STAGE, kt + NST - 1)      // tile kt+1 (two stages) / kt+2 (three) -> the stage step kt-1 read
    f16x8 fb_[2][TN];
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
      for (int j_ = 0; j_ < TN; ++j_) fb_[s_][j_] = *reinterpret_cast<const f16x8*>(st_ + fb_off + s_ * B_PLANE + j_ * 512);
#pragma unroll
    for (int i_ = 0; i_ < TM; ++i_) {
      f16x8 fa_[2];
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) fa_[s_] = *reinterpret_cast<const f16x8*>(st_ + fa_off + s_ * A_PLANE + i_ * 512);
#pragma unroll
      for (int j_ = 0; j_ < TN; ++j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_[1], fb_[0][j_], acc[i_][j_], 0, 0, 0);
#pragma unroll
      for (int j_ = 0; j_ < TN; ++j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_[0], fb_[1][j_], acc[i_][j_], 0, 0, 0);
#pragma unroll
      for (int j_ = 0; j_ < TN; ++j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_[0], fb_[0][j_], acc[i_][j_], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NST == 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(CPW_A + CPW_B) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#undef PROBE_DMA

  // epilogue: the product kernel's (no rescue pass, no row statistics)
  const float ia = pow2_recip(sa), ib = pow2_recip(amax_scale(amax_slot_read(p.b_amax)));
  float alpha = p.alpha;
  {
    const float f = ia * ib, af = p.alpha * f, mag = fabsf(af);
    const bool fold = f >= 1.17549435e-38f && f <= 3.0e38f && ((mag >= 1.17549435e-38f && mag <= 3.0e38f) || p.alpha == 0.f);
    if (fold) alpha = af;
    else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int g = 0; g < 16; ++g) acc[i][j][g] = acc[i][j][g] * ia * ib;
    }
  }
  const int row_base = m0 + wm * TM * 32, col_base = n0 + wn * TN * 32;
  float cmax = 0.f;
  float* scr = reinterpret_cast<float*>(lds) + 4 + wave * EPI_FLOATS;
  const unsigned long long seed = (p.seed && p.drop.thresh) ? *p.seed : 0ull;      // (as in the product kernel: with a constant 0 the compiler drops the masked epilogue and the kernel is half the size)
  static_for<0, TM>([&](auto i) __attribute__((always_inline)) {
    const float4 rowsc = epilogue_row_scales(p, seed, row_base + i * 32, lane);
    static_for<0, TN>([&](auto j) __attribute__((always_inline)) {
      epilogue_tile(p, acc[i][j], row_base + i * 32, col_base + j * 32, lane, scr, 0, seed, cmax, rowsc, alpha);
    });
  });
}

// ---- second variant: the fp32 A operand ITSELF by LDS-DMA (no producer has to write planes), split at fragment-read time.
// A stage holds the raw fp32 tile: 1 KB chunks of 16 rows x 64 bytes (4 lanes fetch one row's 64 bytes, as the register path does:
// same coalescing), the 16-byte quads of a row XOR-swizzled by ((row >> 1) & 3) - chosen by WHICH quad a lane fetches, since the
// DMA writes lane l at byte 16 l - so that the fragment reads (lane = row, two ds_read_b128 per fragment) are conflict free.
// A wave splits the fragments it multiplies (16 elements per lane and K-step: 32 v_fma_mix + the row maxima), twice the split work
// of the register path in total (both column waves split the same rows) - vector-ALU instructions are free beside the MFMAs.
template <int TM, int TN, int WM, int WN, int NST>
__device__ __forceinline__ void probe_adma_body(const GemmParams& p, char* lds, const int bid, const int nwg) {
  constexpr int NW = WM * WN;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int A_BYTES = BM * 64;                                   // fp32: 16 k x 4 bytes per row
  constexpr int KGS_B = BN * 16, B_PLANE = 2 * KGS_B, B_BYTES = 2 * B_PLANE;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int CPS_B = BN / 64, NCH_B = 4 * CPS_B, CPW_B = NCH_B / NW;
  constexpr int NCH_A = BM / 16, CPW_A = NCH_A / NW;
  static_assert(NCH_A % NW == 0 && NCH_B % NW == 0 && (NST == 2 || NST == 3), "chunk maps");
  static_assert(NST * STAGE <= gemm_f16_lds_bytes<TM, TN, WM, WN, NST>(), "lds");

  int tile_m, tile_n;
  tile_of_block_unsplit(p, nwg, bid, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
  const int nk = p.K >> 4;
  const float sa = amax_scale(amax_slot_read(p.a_amax));
#ifdef PROBE_AMAX
  const unsigned int cpeek = amax_peek_block(p.c_amax);
#endif
#ifdef PROBE_GUARD
  __shared__ int guard_flags[NW];
  float rf[TM] = {1.f, 1.f};
  bool rescued = false;
#endif
  float rm[TM], sa_row[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) { rm[i] = 0.f; sa_row[i] = sa; if (p.adrop.thresh) sa_row[i] *= 2.f; }

  f32x16 acc[TM][TN];
#ifdef PROBE_PASSLOOP
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
#endif
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

  const char* b_base = reinterpret_cast<const char*>(p.b_pre);
  const char* a_base = reinterpret_cast<const char*>(p.A + (long long)m0 * p.lda);
  unsigned int b_off[CPW_B], a_off[CPW_A];
  int b_dst[CPW_B], a_dst[CPW_A];
#pragma unroll
  for (int i = 0; i < CPW_B; ++i) {
    const int c = wave + NW * i;
    const int slab = c / CPS_B, part = c - slab * CPS_B, s = slab >> 1, kg = slab & 1;
    const int col = min(n0 + part * 64 + lane, p.N - 1);
    b_off[i] = (unsigned int)(((long long)s * (p.b_pre_plane >> 3) + (long long)kg * p.N + col) * 16);
    b_dst[i] = A_BYTES + s * B_PLANE + kg * KGS_B + part * 1024;
  }
#pragma unroll
  for (int i = 0; i < CPW_A; ++i) {
    const int c = wave + NW * i;                                      // rows 16 c .. 16 c + 15 of the tile
    const int row = 16 * c + (lane >> 2), q = (lane & 3) ^ ((row >> 1) & 3);
    a_off[i] = (unsigned int)(((long long)(min(m0 + row, p.M - 1) - m0) * p.lda + 4 * q) * 4);
    a_dst[i] = c * 1024;
  }
  const long long b_step = 32ll * p.N;
  const unsigned int lds_u32 = (unsigned int)(size_t)((lptr_t)lds);
  int fa_off[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = wm * TM * 32 + i * 32 + r;
    fa_off[i] = row * 64 + (((2 * h) ^ ((row >> 1) & 3)) * 16);
  }
  const int fb_off = A_BYTES + h * KGS_B + (wn * TN * 32 + r) * 16;

#define PROBE_DMA2(ST, KT)                                                                                 \
  { const int kt_ = min((KT), nk - 1);                                                                     \
    const char* bb_ = b_base + (long long)kt_ * b_step; const char* ab_ = a_base + ((long long)kt_ << 6);  \
    _Pragma("unroll") for (int i_ = 0; i_ < CPW_B; ++i_) {                                                 \
      const unsigned int m0_ = lds_u32 + (unsigned int)((ST) - lds) + (unsigned int)b_dst[i_];             \
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0_), "v"(b_off[i_]), "s"(bb_) : "memory", "m0"); } \
    _Pragma("unroll") for (int i_ = 0; i_ < CPW_A; ++i_) {                                                 \
      const unsigned int m0_ = lds_u32 + (unsigned int)((ST) - lds) + (unsigned int)a_dst[i_];             \
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0_), "v"(a_off[i_]), "s"(ab_) : "memory", "m0"); } }

  PROBE_DMA2(lds, 0)
  if constexpr (NST == 3) {
    PROBE_DMA2(lds + STAGE, 1)
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(CPW_A + CPW_B) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();

  for (int kt = 0; kt < nk; ++kt) {
    const char* st_ = lds + (kt % NST) * STAGE;
    PROBE_DMA2(lds + ((kt + NST - 1) % NST) * STAGE, kt + NST - 1)
    f32x4 ra_[TM][2];
#pragma unroll
    for (int i_ = 0; i_ < TM; ++i_) {
      ra_[i_][0] = *reinterpret_cast<const f32x4*>(st_ + fa_off[i_]);
      ra_[i_][1] = *reinterpret_cast<const f32x4*>(st_ + (fa_off[i_] ^ 16));
    }
    f16x8 fb_[2][TN];
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
      for (int j_ = 0; j_ < TN; ++j_) fb_[s_][j_] = *reinterpret_cast<const f16x8*>(st_ + fb_off + s_ * B_PLANE + j_ * 512);
#pragma unroll
    for (int i_ = 0; i_ < TM; ++i_) {
      f16x4 h0, l0, h1, l1;
#ifdef PROBE_ROWMAX
      asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(rm[i_]) : "v"(ra_[i_][0][0]), "v"(ra_[i_][0][1]));
      asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(rm[i_]) : "v"(ra_[i_][0][2]), "v"(ra_[i_][0][3]));
      asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(rm[i_]) : "v"(ra_[i_][1][0]), "v"(ra_[i_][1][1]));
      asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(rm[i_]) : "v"(ra_[i_][1][2]), "v"(ra_[i_][1][3]));
#endif
#ifdef PROBE_ROWSCALE
      split_f16_scaled(ra_[i_][0], sa_row[i_], h0, l0);
      split_f16_scaled(ra_[i_][1], sa_row[i_], h1, l1);
#else
      split_f16_scaled(ra_[i_][0], sa, h0, l0);
      split_f16_scaled(ra_[i_][1], sa, h1, l1);
#endif
      const f16x8 fa_hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7), fa_lo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int j_ = 0; j_ < TN; ++j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_lo, fb_[0][j_], acc[i_][j_], 0, 0, 0);
#pragma unroll
      for (int j_ = 0; j_ < TN; ++j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_hi, fb_[1][j_], acc[i_][j_], 0, 0, 0);
#pragma unroll
      for (int j_ = 0; j_ < TN; ++j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_hi, fb_[0][j_], acc[i_][j_], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NST == 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(CPW_A + CPW_B) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#ifdef PROBE_PASSLOOP
#ifdef PROBE_GUARD
    if (pass == 1) break;
    const int et = (int)((__float_as_uint(sa) >> 23) & 0xffu);
    bool want = false;
    int d_[TM]; bool ok_[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      rm[i] = fmaxf(rm[i], __shfl_xor(rm[i], 32));
      const int e = (int)((__float_as_uint(rm[i]) >> 23) & 0xffu);
      d_[i] = 268 - et - e;
      ok_[i] = e >= 16 && e != 255 && d_[i] > 0 && d_[i] <= 120;
      want = want || (ok_[i] && d_[i] >= 18);
    }
    if (!block_any<NW>(want, guard_flags, wave)) break;
    const float isa = pow2_recip(sa);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const float g = sa_row[i] * isa;
      rf[i] = ok_[i] ? __uint_as_float((unsigned int)(127 - d_[i]) << 23) : 1.f;
      sa_row[i] = (ok_[i] ? amax_scale(rm[i]) : sa) * g;
    }
    rescued = true;
  }
#else
    float mx = fmaxf(rm[0], rm[1]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (pass == 1 || !(mx > 3.0e38f)) break;          // (never taken twice: the structure is what is measured)
    sa_row[0] *= 0.5f; sa_row[1] *= 0.5f;
  }
#endif
#endif
#undef PROBE_DMA2
  if (rm[0] + rm[1] == -1.f) acc[0][0][0] += sa_row[0];   // (keeps the row maxima alive)

  const float ia = pow2_recip(sa), ib = pow2_recip(amax_scale(amax_slot_read(p.b_amax)));
  float alpha = p.alpha;
  {
    const float f = ia * ib, af = p.alpha * f, mag = fabsf(af);
    const bool fold = f >= 1.17549435e-38f && f <= 3.0e38f && ((mag >= 1.17549435e-38f && mag <= 3.0e38f) || p.alpha == 0.f);
    if (fold) alpha = af;
    else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int g = 0; g < 16; ++g) acc[i][j][g] = acc[i][j][g] * ia * ib;
    }
  }
  const int row_base = m0 + wm * TM * 32, col_base = n0 + wn * TN * 32;
  float cmax = 0.f;
  float* scr = reinterpret_cast<float*>(lds) + 4 + wave * EPI_FLOATS;
#if defined(PROBE_GUARD) && defined(PROBE_PASSLOOP)
  if (rescued) {
    float* rowfac = reinterpret_cast<float*>(lds) + 4 + NW * EPI_FLOATS;
    if (h == 0 && wn == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i) rowfac[wm * TM * 32 + i * 32 + r] = rf[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const float f = rowfac[wm * TM * 32 + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * h];
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j][g] *= f;
      }
  }
#endif
  const unsigned long long seed = (p.seed && p.drop.thresh) ? *p.seed : 0ull;      // (as in the product kernel: with a constant 0 the compiler drops the masked epilogue and the kernel is half the size)
  static_for<0, TM>([&](auto i) __attribute__((always_inline)) {
    const float4 rowsc = epilogue_row_scales(p, seed, row_base + i * 32, lane);
    static_for<0, TN>([&](auto j) __attribute__((always_inline)) {
      epilogue_tile(p, acc[i][j], row_base + i * 32, col_base + j * 32, lane, scr, 0, seed, cmax, rowsc, alpha);
    });
  });
#ifdef PROBE_AMAX
  amax_slot_commit_block(p.c_amax, cmax, reinterpret_cast<float*>(lds), cpeek);
#endif
}

template <int NST>
__global__ __launch_bounds__(256, 2) void probe_adma_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) char lds[gemm_f16_lds_bytes<2, 4, 2, 2, NST>()];
  probe_adma_body<2, 4, 2, 2, NST>(p, lds, blockIdx.x, gridDim.x);
}

template <int NST>
__global__ __launch_bounds__(256, 2) void gemm_f16_aplanes_kernel(GemmParams p, const char* a_pre, long long a_plane_bytes) {
  __shared__ __attribute__((aligned(16))) char lds[gemm_f16_lds_bytes<2, 4, 2, 2, NST>()];
  gemm_f16_aplanes_body<2, 4, 2, 2, NST>(p, a_pre, a_plane_bytes, lds, blockIdx.x, gridDim.x);
}

__global__ void fill_kernel(float* x, long long n, unsigned int seed, float scale) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const unsigned int hsh = mix32((unsigned int)i * 0x9e3779b9u + seed);
    x[i] = scale * ((float)(hsh >> 8) * (1.f / 8388608.f) - 1.f) * (1.f + 3.f * (float)((hsh & 255u) == 0u));
  }
}

}  // namespace npvp

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

template <class F>
static float time_us(F&& f, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return 1e3f * ms / iters;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 114688;
  const int shapes[5][2] = {{512, 512}, {1024, 512}, {2048, 512}, {512, 2048}, {512, 1024}};       // (N, K) of the path's layers
  const int KMAX = 2048, NMAX = 2048;
  float *A, *W, *C0, *C1, *slotA, *slotW;
  void *PA, *PW;
  HIPCHECK(hipMalloc(&A, (size_t)M * KMAX * 4)); HIPCHECK(hipMalloc(&PA, (size_t)M * KMAX * 4));
  HIPCHECK(hipMalloc(&W, (size_t)NMAX * KMAX * 4)); HIPCHECK(hipMalloc(&PW, (size_t)NMAX * KMAX * 4));
  HIPCHECK(hipMalloc(&C0, (size_t)M * NMAX * 4)); HIPCHECK(hipMalloc(&C1, (size_t)M * NMAX * 4));
  HIPCHECK(hipMalloc(&slotA, AMAX_WORDS * AMAX_STRIDE * 4)); HIPCHECK(hipMalloc(&slotW, AMAX_WORDS * AMAX_STRIDE * 4));
  double tot[5] = {0, 0, 0, 0, 0}, flops = 0;
  for (int si = 0; si < 5; ++si) {
    const int N = shapes[si][0], K = shapes[si][1];
    npvp::fill_kernel<<<4096, 256>>>(A, (long long)M * K, 11u + si, 1.0f);
    npvp::fill_kernel<<<1024, 256>>>(W, (long long)N * K, 77u + si, 0.05f);
    if (npvp_split_weight_f16(W, K, N, K, PW, nullptr, slotW, 0) || npvp_split_weight_f16(A, K, M, K, PA, nullptr, slotA, 0)) {
      printf("split failed: %s\n", npvp_last_error()); return 2; }
    GemmParams p = {};
    p.A = A; p.lda = K; p.C = C0; p.ldc = N; p.M = M; p.N = N; p.K = K; p.alpha = 1.f; p.splits = 1;
    p.b_pre = PW; p.b_pre_plane = (long long)N * K; p.a_amax = slotA; p.b_amax = slotW;
    GemmParams q = p;
    if (!launch_gemm_f16(q, 0)) { printf("the product kernel did not take the shape\n"); return 2; }
    GemmParams v = p; v.C = C1;
    if (prep_gemm_f16(v) != 1) {        // (small outputs: the product takes smaller tiles; the variants are 128 x 256-tile kernels)
      v.tiles_m = (M + 127) / 128; v.tiles_n = (N + 255) / 256; v.colgroups = pick_colgroups((long long)N * K * 4, v.tiles_m, v.tiles_n); }
    const dim3 grid(v.tiles_m * v.tiles_n), block(256);
    const long long a_plane_bytes = (long long)M * K * 2;
    npvp::gemm_f16_aplanes_kernel<2><<<grid, block>>>(v, (const char*)PA, a_plane_bytes);
    HIPCHECK(hipDeviceSynchronize());
    // bit-identical?
    const long long RCHK = M >= 16384 ? 4096 : M / 4;
    std::vector<float> h0((size_t)RCHK * N), h1((size_t)RCHK * N);
    size_t bad = 0;
    for (long long r0 : {0ll, (long long)M / 2, (long long)M - RCHK}) {
      HIPCHECK(hipMemcpy(h0.data(), C0 + r0 * N, h0.size() * 4, hipMemcpyDeviceToHost));
      HIPCHECK(hipMemcpy(h1.data(), C1 + r0 * N, h1.size() * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < h0.size(); ++i) bad += h0[i] != h1[i];
    }
    // the fp32 operand by DMA, split at fragment time: bit-identical too?
    HIPCHECK(hipMemset(C1, 0, (size_t)M * N * 4));
    npvp::probe_adma_kernel<3><<<grid, block>>>(v);
    HIPCHECK(hipDeviceSynchronize());
    size_t bad2 = 0;
    for (long long r0 : {0ll, (long long)M / 2, (long long)M - RCHK}) {
      HIPCHECK(hipMemcpy(h0.data(), C0 + r0 * N, h0.size() * 4, hipMemcpyDeviceToHost));
      HIPCHECK(hipMemcpy(h1.data(), C1 + r0 * N, h1.size() * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < h0.size(); ++i) bad2 += h0[i] != h1[i];
    }
    // TIMING.  The first version of this program timed the kernels one after the other, the product kernel first - right after
    // the host had been copying results, i.e. on a GPU that had just idled: it read 10 % slow and every variant after it 10 % fast.
    // Now: a long warm-up, then three ROUNDS over all five kernels (10 launches each), the fastest round of each kernel counts.
    auto k0 = [&] { npvp::gemm_f16_kernel<2, 4, 2, 2, false><<<grid, block>>>(v); };            // the product's kernel (register path)
    auto k1 = [&] { npvp::gemm_f16_aplanes_kernel<2><<<grid, block>>>(v, (const char*)PA, a_plane_bytes); };
    auto k2 = [&] { npvp::gemm_f16_aplanes_kernel<3><<<grid, block>>>(v, (const char*)PA, a_plane_bytes); };
    auto k3 = [&] { npvp::probe_adma_kernel<2><<<grid, block>>>(v); };
    auto k4 = [&] { npvp::probe_adma_kernel<3><<<grid, block>>>(v); };
    for (int w = 0; w < 40; ++w) { k0(); k2(); }
    float best[5] = {1e30f, 1e30f, 1e30f, 1e30f, 1e30f};
    for (int round = 0; round < 3; ++round) {
      best[0] = fminf(best[0], time_us(k0, 10)); best[1] = fminf(best[1], time_us(k1, 10)); best[2] = fminf(best[2], time_us(k2, 10));
      best[3] = fminf(best[3], time_us(k3, 10)); best[4] = fminf(best[4], time_us(k4, 10));
    }
    const float t0 = best[0], t2 = best[1], t3 = best[2], t4 = best[3], t5 = best[4];
    const double fl = 2.0 * M * N * K;
    printf("R=%6d N=%5d K=%5d  product %7.1f us (%6.1f TF)   A planes by DMA, 2 stages %7.1f us (%6.1f TF)   3 stages %7.1f us (%6.1f TF)   "
           "mismatching outputs: %zu of %zu\n", M, N, K, t0, fl / t0 / 1e6, t2, fl / t2 / 1e6, t3, fl / t3 / 1e6, bad, 3 * h0.size());
    printf("                            fp32 A by DMA + split at fragment time, 2 stages %7.1f us (%6.1f TF)   3 stages %7.1f us (%6.1f TF)   "
           "mismatching outputs: %zu\n", t4, fl / t4 / 1e6, t5, fl / t5 / 1e6, bad2);
    tot[0] += t0; tot[1] += t2; tot[2] += t3; tot[3] += t4; tot[4] += t5; flops += fl;
  }
  printf("R=%6d all shapes: product %.1f TF, A planes 2 stages %.1f TF, 3 stages %.1f TF; fp32 A by DMA 2 stages %.1f TF, 3 stages %.1f TF\n", M,
         flops / tot[0] / 1e6, flops / tot[1] / 1e6, flops / tot[2] / 1e6, flops / tot[3] / 1e6, flops / tot[4] / 1e6);
  return 0;
}
