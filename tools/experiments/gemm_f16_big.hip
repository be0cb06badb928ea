// The 256-row-tile form of the two-term fp16 forward / dgrad GEMM (gemm_f16.hip has the arithmetic and the operand formats):
// 256 x 256 (or 256 x 128) output tile per workgroup, 4 waves of 128 x 128 (128 x 64), ONE wave per SIMD with the whole register
// file (256 accumulator registers + ~120 working ones), K-step 16, two LDS stages.
//
// Why: the 128 x 256 kernel is bound by instruction issue around its MFMAs (profiles/r03_pmc_gemm.md, r03_issue_cost_probe.txt) - per
// MFMA 2.4 VALU, 0.5 LDS reads, 0.17 LDS writes and 0.25 vector-memory instructions, ~27 issue cycles beside each 32-cycle MFMA,
// more than the 24 a wave can hide (MI355X_MICROARCH.md, "single-issue instructions hidden per MFMA gap").  The bytes per MFMA
// through the vector-memory and LDS paths are set by the tile shape: a 128 x 128 wave tile needs a third fewer fragment reads and
// staging instructions per MFMA than 64 x 128 (per K-step and wave: 48 MFMAs, 16 fragment reads, 8 LDS writes, 8 vector-memory
// instructions, ~60 VALU: ~17 issue cycles per MFMA), and with one wave per SIMD nothing is time-sliced: what does not fit the gap
// is exposed, so the step is laid out by hand in four sub-blocks of 12 MFMAs with the staging work spread over them.
#include "gemm.h"

namespace npvp {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_f16_big(const f32x4 v, f16x4& hi, f16x4& lo) {      // (gemm_f16.hip: 5 VALU per pair)
  const f32x2_t a = {v[0], v[1]}, b = {v[2], v[3]};
  const f16x2_t ha = __builtin_convertvector(a, f16x2_t), hb = __builtin_convertvector(b, f16x2_t);
  const f32x2_t ra = a - __builtin_convertvector(ha, f32x2_t), rb = b - __builtin_convertvector(hb, f32x2_t);
  const f16x2_t la = __builtin_convertvector(ra, f16x2_t), lb = __builtin_convertvector(rb, f16x2_t);
  hi[0] = ha[0]; hi[1] = ha[1]; hi[2] = hb[0]; hi[3] = hb[1];
  lo[0] = la[0]; lo[1] = la[1]; lo[2] = lb[0]; lo[3] = lb[1];
}

// TNW = 32-column blocks per wave: 4 -> 256 x 256 tile, 2 -> 256 x 128 tile
template <int TNW>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_f16_big_kernel(GemmParams p) {
  constexpr int TM = 4, NW = 4, BM = 256, BN = 64 * TNW;
  constexpr int KGS_A = BM * 16 + 64, A_PLANE = 2 * KGS_A, A_BYTES = 2 * A_PLANE;
  constexpr int KGS_B = BN * 16, B_PLANE = 2 * KGS_B, B_BYTES = 2 * B_PLANE;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int CPS = BN / 64, NCHUNK = 4 * CPS, CPW = NCHUNK / NW;        // LDS-DMA pieces (1 KB) of a B stage, per wave
  static_assert(NCHUNK % NW == 0 && (CPW == 4 || CPW == 2), "DMA pieces per wave");
  constexpr int EPI_BYTES = (4 + NW * EPI_FLOATS) * 4;
  __shared__ __attribute__((aligned(16))) char lds[2 * STAGE > EPI_BYTES ? 2 * STAGE : EPI_BYTES];

  int tile_m, tile_n;
  tile_of_block_unsplit(p, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int nk = p.K >> 4;
  const float sa = amax_scale(amax_slot_read(p.a_amax));
  const unsigned int cpeek = amax_peek_block(p.c_amax);

  f32x16 acc[TM][TNW];

  // A staging: thread (rl = t >> 2, quad = t & 3) stages 16 bytes (4 k) of rows rl + 64 pass, pass = 0..3
  const int quad = t & 3, rl = t >> 2;
  const char* a_base = reinterpret_cast<const char*>(p.A + (long long)m0 * p.lda);
  unsigned int a_off[4];
  float sap[4];
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    a_off[ps] = (unsigned int)(((long long)(min(m0 + rl + 64 * ps, p.M - 1) - m0) * p.lda + 4 * quad) * 4);
    sap[ps] = sa;
  }
  if (p.adrop.thresh) {
    const unsigned long long aseed = *p.seed;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) sap[ps] *= drop_spec_scale(p.adrop, aseed, min(m0 + rl + 64 * ps, p.M - 1), 0, 1);
  }
  const int a_dst = (quad >> 1) * KGS_A + (quad & 1) * 8 + rl * 16;
  const char* b_base = reinterpret_cast<const char*>(p.b_pre);
  unsigned int b_off[CPW];
  int b_dst[CPW];
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int c = wave + NW * i;
    const int slab = c / CPS, part = c - slab * CPS, s = slab >> 1, kg = slab & 1;
    const int col = min(n0 + part * 64 + lane, p.N - 1);
    b_off[i] = (unsigned int)(((long long)s * (p.b_pre_plane >> 3) + (long long)kg * p.N + col) * 16);
    b_dst[i] = A_BYTES + s * B_PLANE + kg * KGS_B + part * 1024;
  }
  const long long b_step = 32ll * p.N;
  const unsigned int lds_u32 = (unsigned int)(size_t)((lptr_t)lds);
  const int fa_off = h * KGS_A + (wm * TM * 32 + r) * 16;
  const int fb_off = A_BYTES + h * KGS_B + (wn * TNW * 32 + r) * 16;

  // two register sets of the A tile (4 row passes each): set 0 = ea, set 1 = eb
  f32x4 ea[4], eb[4];
#define NPVP_B_ALOAD1(R, PS, KT)                                                                           \
  { const char* ab_ = a_base + ((long long)min((KT), nk - 1) << 6);                                        \
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"((R)[PS]) : "v"(a_off[PS]), "s"(ab_) : "memory"); }
#define NPVP_B_ASTORE1(ST, R, PS)                                                                          \
  { f16x4 hi_, lo_; split_f16_big((R)[PS] * sap[PS], hi_, lo_);                                            \
    *reinterpret_cast<f16x4*>((ST) + a_dst + (PS) * 1024) = hi_;                                           \
    *reinterpret_cast<f16x4*>((ST) + A_PLANE + a_dst + (PS) * 1024) = lo_; }
#define NPVP_B_DMA1(ST, I, KT)                                                                             \
  { const char* bb_ = b_base + (long long)min((KT), nk - 1) * b_step;                                      \
    const unsigned int m0_ = lds_u32 + (unsigned int)((ST) - lds) + (unsigned int)b_dst[I];                \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"                           \
                 :: "s"(m0_), "v"(b_off[I]), "s"(bb_) : "memory", "m0"); }
  // (row pass ps of the A tile sits 64 rows = 1024 bytes further in each k-group slab)

  // prologue: tile 0 -> stage 0; A tiles 1 and 2 -> register sets; the first fragments -> registers
#pragma unroll
  for (int i = 0; i < CPW; ++i) NPVP_B_DMA1(lds, i, 0)
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) NPVP_B_ALOAD1(ea, ps, 0)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(ea[0]), "+v"(ea[1]), "+v"(ea[2]), "+v"(ea[3]) :: "memory");
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) NPVP_B_ASTORE1(lds, ea, ps)
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) NPVP_B_ALOAD1(ea, ps, 1)
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) NPVP_B_ALOAD1(eb, ps, 2)
  asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" : "+v"(ea[0]), "+v"(ea[1]), "+v"(ea[2]), "+v"(ea[3]) :: "memory");
  __builtin_amdgcn_s_barrier();

  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // fragment registers: the B fragments of a whole K-step (two sets: the step in flight and the next one) and the A fragments of
  // two consecutive row blocks
  f16x8 fbA[2][TNW], fbB[2][TNW], fa[2][2];
#define NPVP_B_READ_FB(FB, ST)                                                                             \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                         \
    _Pragma("unroll") for (int j_ = 0; j_ < TNW; ++j_)                                                     \
      (FB)[s_][j_] = *reinterpret_cast<const f16x8*>((ST) + fb_off + s_ * B_PLANE + j_ * 512);
#define NPVP_B_READ_FA(SLOT, ST, I)                                                                        \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                         \
    fa[SLOT][s_] = *reinterpret_cast<const f16x8*>((ST) + fa_off + s_ * A_PLANE + (I) * 512);
#define NPVP_B_MFMAS(I, SLOT, FB, FIRST)                                                                   \
  _Pragma("unroll") for (int j_ = 0; j_ < TNW; ++j_) acc[I][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[SLOT][1], (FB)[0][j_], (FIRST) ? zero16 : acc[I][j_], 0, 0, 0); \
  _Pragma("unroll") for (int j_ = 0; j_ < TNW; ++j_) acc[I][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[SLOT][0], (FB)[1][j_], acc[I][j_], 0, 0, 0); \
  _Pragma("unroll") for (int j_ = 0; j_ < TNW; ++j_) acc[I][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[SLOT][0], (FB)[0][j_], acc[I][j_], 0, 0, 0);
#define NPVP_B_FENCE __builtin_amdgcn_sched_barrier(0);
  NPVP_B_READ_FB(fbA, lds)
  NPVP_B_READ_FA(0, lds, 0)

  // One K-step = four sub-blocks (A row block i = 0..3, 3 * TNW MFMAs each), fenced against each other; inside a sub-block the
  // compiler interleaves the staging work of the step with the MFMAs:
  //   sub-block 0: MFMAs of row block 0 | reads of row block 1's A fragments; LDS-DMA pieces of B tile kt+1 (first half); split +
  //                LDS write of row passes 0, 1 of A tile kt+1
  //   sub-block 1: row block 1 | A fragments of block 2; DMA pieces (second half); row pass 2
  //   sub-block 2: row block 2 | A fragments of block 3; row pass 3; global loads of row passes 0, 1 of A tile kt+3
  //   -- "vmcnt(2) lgkmcnt(0)" + barrier: stage NXT is complete (the DMA pieces are older than the two loads just issued), and
  //      nobody reads stage CUR any more (its last fragments are in registers) --
  //   sub-block 3: row block 3 | ALL B fragments and the first A fragments of the NEXT step from stage NXT (second register set);
  //                global loads of row passes 2, 3
  // so the next step's first MFMA finds its operands in registers: the LDS latency behind the barrier is hidden by 12 MFMAs.
#define NPVP_B_STEP(KT, CUR, NXT, R, RN, FB, FBN, FIRST)                                                   \
  {                                                                                                        \
    const char* st_ = lds + (CUR) * STAGE;                                                                 \
    char* nx_ = lds + (NXT) * STAGE;                                                                       \
    NPVP_B_READ_FA(1, st_, 1)                                                                              \
    _Pragma("unroll") for (int c_ = 0; c_ < CPW / 2; ++c_) NPVP_B_DMA1(nx_, c_, (KT) + 1)                  \
    NPVP_B_ASTORE1(nx_, R, 0) NPVP_B_ASTORE1(nx_, R, 1)                                                    \
    NPVP_B_MFMAS(0, 0, FB, FIRST)                                                                          \
    NPVP_B_FENCE                                                                                           \
    NPVP_B_READ_FA(0, st_, 2)                                                                              \
    _Pragma("unroll") for (int c_ = 0; c_ < CPW / 2; ++c_) NPVP_B_DMA1(nx_, CPW / 2 + c_, (KT) + 1)        \
    NPVP_B_ASTORE1(nx_, R, 2)                                                                              \
    NPVP_B_MFMAS(1, 1, FB, FIRST)                                                                          \
    NPVP_B_FENCE                                                                                           \
    NPVP_B_READ_FA(1, st_, 3)                                                                              \
    NPVP_B_ASTORE1(nx_, R, 3)                                                                              \
    NPVP_B_ALOAD1(R, 0, (KT) + 3) NPVP_B_ALOAD1(R, 1, (KT) + 3)                                            \
    NPVP_B_MFMAS(2, 0, FB, FIRST)                                                                          \
    NPVP_B_FENCE                                                                                           \
    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" : "+v"((RN)[0]), "+v"((RN)[1]), "+v"((RN)[2]), "+v"((RN)[3]) :: "memory"); \
    __builtin_amdgcn_s_barrier();                                                                          \
    NPVP_B_FENCE                                                                                           \
    NPVP_B_READ_FB(FBN, nx_)                                                                               \
    NPVP_B_ALOAD1(R, 2, (KT) + 3) NPVP_B_ALOAD1(R, 3, (KT) + 3)                                            \
    NPVP_B_MFMAS(3, 1, FB, FIRST)                                                                          \
    NPVP_B_READ_FA(0, nx_, 0)                                                                              \
    NPVP_B_FENCE                                                                                           \
  }

  NPVP_B_STEP(0, 0, 1, ea, eb, fbA, fbB, true)
  int kt = 1;
  for (; kt + 1 < nk; kt += 2) {
    NPVP_B_STEP(kt, 1, 0, eb, ea, fbB, fbA, false)
    NPVP_B_STEP(kt + 1, 0, 1, ea, eb, fbA, fbB, false)
  }
  if (kt < nk) NPVP_B_STEP(kt, 1, 0, eb, ea, fbB, fbA, false)
  // (the clamped loads past the last tile; both register sets tied: see gemm_f16_kernel)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(ea[0]), "+v"(ea[1]), "+v"(ea[2]), "+v"(ea[3]), "+v"(eb[0]), "+v"(eb[1]), "+v"(eb[2]), "+v"(eb[3]) :: "memory");
  __builtin_amdgcn_s_barrier();
#undef NPVP_B_FENCE
#undef NPVP_B_MFMAS
#undef NPVP_B_READ_FA
#undef NPVP_B_READ_FB
#undef NPVP_B_STEP
#undef NPVP_B_DMA1
#undef NPVP_B_ASTORE1
#undef NPVP_B_ALOAD1

  const float ia = pow2_recip(sa), ib = pow2_recip(amax_scale(amax_slot_read(p.b_amax)));
  GemmParams q = p;
  {
    const float f = ia * ib, af = p.alpha * f, mag = fabsf(af);
    const bool fold = f >= 1.17549435e-38f && f <= 3.0e38f && ((mag >= 1.17549435e-38f && mag <= 3.0e38f) || p.alpha == 0.f);
    if (fold) q.alpha = af;
    else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j)
#pragma unroll
          for (int g = 0; g < 16; ++g) acc[i][j][g] = acc[i][j][g] * ia * ib;
    }
  }
  const int row_base = m0 + wm * TM * 32, col_base = n0 + wn * TNW * 32;
  float cmax = 0.f;
  float* scr = reinterpret_cast<float*>(lds) + 4 + wave * EPI_FLOATS;
  const unsigned long long seed = (q.seed && q.drop.thresh) ? *q.seed : 0ull;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TNW; ++j) epilogue_tile(q, acc[i][j], row_base + i * 32, col_base + j * 32, lane, scr, 0, seed, cmax);
  amax_slot_commit_block(p.c_amax, cmax, reinterpret_cast<float*>(lds), cpeek);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Wave-specialised form: 256 x 128 tile, EIGHT waves - waves 0..3 only multiply (128 x 64 each: 24 MFMAs and 12 fragment reads per
// K-step, nothing else), waves 4..7 only stage (global loads of the fp32 A tile, the fp16 split, LDS writes, the LDS-DMA of the B
// planes).  A SIMD holds one wave of each kind: the stager's vector-memory / VALU / LDS-write instructions issue in the 24 free
// cycles beside each of the multiplier's 32-cycle MFMAs and stall only the stager.  Three LDS stages: during step kt the stagers
// fill stage kt+2 while the multipliers take the fragments of tile kt+1 into a second register set and multiply tile kt from the
// first - one barrier per step, and no wave ever waits for a load it has just issued.
template <int DUMMY>
__global__ __launch_bounds__(512, 1) void gemm_f16_ws_kernel(GemmParams p) {
  constexpr int TM = 4, TNW = 2, BM = 256, BN = 128;
  constexpr int KGS_A = BM * 16 + 64, A_PLANE = 2 * KGS_A, A_BYTES = 2 * A_PLANE;
  constexpr int KGS_B = BN * 16, B_PLANE = 2 * KGS_B, B_BYTES = 2 * B_PLANE;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int CPS = BN / 64, NCHUNK = 4 * CPS, CPW = NCHUNK / 4;          // 8 LDS-DMA pieces per stage, 2 per staging wave
  __shared__ __attribute__((aligned(16))) char lds[3 * STAGE];
  static_assert(3 * STAGE >= (4 + 4 * EPI_FLOATS) * 4, "epilogue scratch");

  int tile_m, tile_n;
  tile_of_block_unsplit(p, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nk = p.K >> 4;
  const float sa = amax_scale(amax_slot_read(p.a_amax));
  const unsigned int cpeek = amax_peek_block(p.c_amax);

  if (wave >= 4) {
    // ------------------------------------------------------------------ staging waves
    const int lt = t - 256, lw = wave - 4;
    const int quad = lt & 3, rl = lt >> 2;
    const char* a_base = reinterpret_cast<const char*>(p.A + (long long)m0 * p.lda);
    unsigned int a_off[4];
    float sap[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      a_off[ps] = (unsigned int)(((long long)(min(m0 + rl + 64 * ps, p.M - 1) - m0) * p.lda + 4 * quad) * 4);
      sap[ps] = sa;
    }
    if (p.adrop.thresh) {
      const unsigned long long aseed = *p.seed;
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) sap[ps] *= drop_spec_scale(p.adrop, aseed, min(m0 + rl + 64 * ps, p.M - 1), 0, 1);
    }
    const int a_dst = (quad >> 1) * KGS_A + (quad & 1) * 8 + rl * 16;
    const char* b_base = reinterpret_cast<const char*>(p.b_pre);
    unsigned int b_off[CPW];
    int b_dst[CPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      const int c = lw + 4 * i;
      const int slab = c / CPS, part = c - slab * CPS, s = slab >> 1, kg = slab & 1;
      const int col = min(n0 + part * 64 + lane, p.N - 1);
      b_off[i] = (unsigned int)(((long long)s * (p.b_pre_plane >> 3) + (long long)kg * p.N + col) * 16);
      b_dst[i] = A_BYTES + s * B_PLANE + kg * KGS_B + part * 1024;
    }
    const long long b_step = 32ll * p.N;
    const unsigned int lds_u32 = (unsigned int)(size_t)((lptr_t)lds);
    f32x4 ea[4], eb[4];
#define NPVP_S_ALOAD(R, KT)                                                                                \
  { const char* ab_ = a_base + ((long long)min((KT), nk - 1) << 6);                                        \
    _Pragma("unroll") for (int ps_ = 0; ps_ < 4; ++ps_)                                                    \
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"((R)[ps_]) : "v"(a_off[ps_]), "s"(ab_) : "memory"); }
#define NPVP_S_ASTORE(SI, R)                                                                               \
  { char* st_ = lds + (SI) * STAGE;                                                                        \
    _Pragma("unroll") for (int ps_ = 0; ps_ < 4; ++ps_) {                                                  \
      f16x4 hi_, lo_; split_f16_big((R)[ps_] * sap[ps_], hi_, lo_);                                        \
      *reinterpret_cast<f16x4*>(st_ + a_dst + ps_ * 1024) = hi_;                                           \
      *reinterpret_cast<f16x4*>(st_ + A_PLANE + a_dst + ps_ * 1024) = lo_; } }
#define NPVP_S_DMA(SI, KT)                                                                                 \
  { const char* bb_ = b_base + (long long)min((KT), nk - 1) * b_step;                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < CPW; ++i_) {                                                   \
      const unsigned int m0_ = lds_u32 + (unsigned int)((SI) * STAGE) + (unsigned int)b_dst[i_];           \
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"                         \
                   :: "s"(m0_), "v"(b_off[i_]), "s"(bb_) : "memory", "m0"); } }
    // prologue: B tiles 0, 1, 2 -> stages 0, 1, 2 (LDS-DMA); A tiles 0, 1 -> stages 0, 1; A tiles 2 and 3 -> the register sets
    NPVP_S_DMA(0, 0)
    NPVP_S_DMA(1, 1)
    NPVP_S_ALOAD(ea, 0)
    NPVP_S_ALOAD(eb, 1)
    asm volatile("s_waitcnt vmcnt(4)" : "+v"(ea[0]), "+v"(ea[1]), "+v"(ea[2]), "+v"(ea[3]) :: "memory");
    NPVP_S_ASTORE(0, ea)
    NPVP_S_ALOAD(ea, 2)
    asm volatile("s_waitcnt vmcnt(4)" : "+v"(eb[0]), "+v"(eb[1]), "+v"(eb[2]), "+v"(eb[3]) :: "memory");
    NPVP_S_ASTORE(1, eb)
    NPVP_S_DMA(2, 2)
    NPVP_S_ALOAD(eb, 3)
    // (outstanding, oldest first: ea = A tile 2 (4), DMA of B tile 2 (2), eb = A tile 3 (4); everything older has landed)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                    // stage 0 is published
    __builtin_amdgcn_s_barrier();                                    // (the multipliers' first prefetch barrier: stage 1 is published)
    // step kt: B tile kt + 3 -> stage kt % 3 by LDS-DMA (that stage's fragments were taken a step ago; the pieces have a whole step to
    // land), A tile kt + 2 (the register set loaded two steps ago) -> stage (kt + 2) % 3, reload that set with A tile kt + 4.
    // Vector-memory order per step: 2 DMA pieces, 4 loads.  At the step's end the stage to publish needs the PREVIOUS step's pieces
    // and the next step needs the other register set: "vmcnt(6)" = everything but this step's 6 operations.
    int si = 2, sd = 0;
    for (int kt = 0; kt < nk; kt += 2) {
      NPVP_S_DMA(sd, kt + 3)
      NPVP_S_ASTORE(si, ea)
      NPVP_S_ALOAD(ea, kt + 4)
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" : "+v"(eb[0]), "+v"(eb[1]), "+v"(eb[2]), "+v"(eb[3]) :: "memory");
      __builtin_amdgcn_s_barrier();
      si = si == 2 ? 0 : si + 1; sd = sd == 2 ? 0 : sd + 1;
      if (kt + 1 >= nk) break;
      NPVP_S_DMA(sd, kt + 4)
      NPVP_S_ASTORE(si, eb)
      NPVP_S_ALOAD(eb, kt + 5)
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" : "+v"(ea[0]), "+v"(ea[1]), "+v"(ea[2]), "+v"(ea[3]) :: "memory");
      __builtin_amdgcn_s_barrier();
      si = si == 2 ? 0 : si + 1; sd = sd == 2 ? 0 : sd + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ea[0]), "+v"(ea[1]), "+v"(ea[2]), "+v"(ea[3]), "+v"(eb[0]), "+v"(eb[1]), "+v"(eb[2]), "+v"(eb[3]) :: "memory");
    __builtin_amdgcn_s_barrier();                                    // (before the multipliers reuse the stages as epilogue scratch)
#undef NPVP_S_DMA
#undef NPVP_S_ASTORE
#undef NPVP_S_ALOAD
    return;
  }

  // -------------------------------------------------------------------- multiplying waves
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int fa_off = h * KGS_A + (wm * TM * 32 + r) * 16;
  const int fb_off = A_BYTES + h * KGS_B + (wn * TNW * 32 + r) * 16;
  f32x16 acc[TM][TNW];
  f16x8 faA[TM][2], fbA[2][TNW], faB[TM][2], fbB[2][TNW];
#define NPVP_M_READ(FA, FB, SI)                                                                            \
  { const char* st_ = lds + (SI) * STAGE;                                                                  \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                       \
      _Pragma("unroll") for (int j_ = 0; j_ < TNW; ++j_)                                                   \
        (FB)[s_][j_] = *reinterpret_cast<const f16x8*>(st_ + fb_off + s_ * B_PLANE + j_ * 512);            \
    _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_)                                                      \
      _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                     \
        (FA)[i_][s_] = *reinterpret_cast<const f16x8*>(st_ + fa_off + s_ * A_PLANE + i_ * 512); }
#define NPVP_M_MUL(FA, FB, FIRST)                                                                          \
  _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) {                                                      \
    _Pragma("unroll") for (int j_ = 0; j_ < TNW; ++j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16((FA)[i_][1], (FB)[0][j_], (FIRST) ? zero16 : acc[i_][j_], 0, 0, 0); \
    _Pragma("unroll") for (int j_ = 0; j_ < TNW; ++j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16((FA)[i_][0], (FB)[1][j_], acc[i_][j_], 0, 0, 0); \
    _Pragma("unroll") for (int j_ = 0; j_ < TNW; ++j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16((FA)[i_][0], (FB)[0][j_], acc[i_][j_], 0, 0, 0); }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_s_barrier();                                      // stage 0 is published
  NPVP_M_READ(faA, fbA, 0)
  __builtin_amdgcn_s_barrier();                                      // stage 1 is published
  // step kt: read the fragments of tile kt + 1 from stage (kt + 1) % 3 into the other register set, multiply tile kt
  int sj = 1;
  for (int kt = 0; kt < nk; kt += 2) {
    NPVP_M_READ(faB, fbB, sj)
    if (kt == 0) { NPVP_M_MUL(faA, fbA, true) } else { NPVP_M_MUL(faA, fbA, false) }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    sj = sj == 2 ? 0 : sj + 1;
    if (kt + 1 >= nk) break;
    NPVP_M_READ(faA, fbA, sj)
    NPVP_M_MUL(faB, fbB, false)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    sj = sj == 2 ? 0 : sj + 1;
  }
  __builtin_amdgcn_s_barrier();                                      // (the stagers' trailing loads have landed: the stages are scratch now)
#undef NPVP_M_MUL
#undef NPVP_M_READ

  const float ia = pow2_recip(sa), ib = pow2_recip(amax_scale(amax_slot_read(p.b_amax)));
  GemmParams q = p;
  {
    const float f = ia * ib, af = p.alpha * f, mag = fabsf(af);
    const bool fold = f >= 1.17549435e-38f && f <= 3.0e38f && ((mag >= 1.17549435e-38f && mag <= 3.0e38f) || p.alpha == 0.f);
    if (fold) q.alpha = af;
    else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j)
#pragma unroll
          for (int g = 0; g < 16; ++g) acc[i][j][g] = acc[i][j][g] * ia * ib;
    }
  }
  const int row_base = m0 + wm * TM * 32, col_base = n0 + wn * TNW * 32;
  float cmax = 0.f;
  float* scr = reinterpret_cast<float*>(lds) + 4 + wave * EPI_FLOATS;
  const unsigned long long seed = (q.seed && q.drop.thresh) ? *q.seed : 0ull;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TNW; ++j) epilogue_tile(q, acc[i][j], row_base + i * 32, col_base + j * 32, lane, scr, 0, seed, cmax);
  // amax of the stored values: the four multiplying waves only (the stagers have left)
  if (p.c_amax) {
    cmax = wave_max(cmax);
    if (lane == 0) amax_word_raise(p.c_amax, blockIdx.x * 4u + wave, cmax, cpeek);
  }
}

// shapes this kernel takes (experiment switch NPVP_F16_BIG: 0 off, 1 = 256 x 256 tiles where they fill the chip)
bool launch_gemm_f16_big(GemmParams& p, hipStream_t stream, int mode) {
  if (p.rowstats || (p.N % 128) != 0 || p.M < 4096) return false;
  const int tm = (p.M + 255) / 256;
  const bool wide = mode != 2 && p.N % 256 == 0 && tm * (p.N / 256) >= 512;
  p.tiles_m = tm;
  p.colgroups = 1;
  if (mode == 3) {
    p.tiles_n = p.N / 128;
    hipLaunchKernelGGL((gemm_f16_ws_kernel<0>), dim3(p.tiles_m * p.tiles_n), dim3(512), 0, stream, p);
    return true;
  }
  if (wide) {
    p.tiles_n = p.N / 256;
    hipLaunchKernelGGL((gemm_f16_big_kernel<4>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, stream, p);
  } else {
    p.tiles_n = p.N / 128;
    hipLaunchKernelGGL((gemm_f16_big_kernel<2>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, stream, p);
  }
  return true;
}

}  // namespace npvp
