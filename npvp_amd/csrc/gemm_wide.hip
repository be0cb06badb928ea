// Wide-tile form of the split-precision GEMM for the large forward / dgrad shapes (token rows x 512..2048 columns):
//
//     C[M,N] = epilogue( A[M,K] (fp32, k contiguous) * B ),   B = pre-split bf16 planes [term][K/8][N][8 k]
//
// i.e. y = x W^T with the F planes of W and dx = dy W with the D planes of W (npvp_split_weight): the weights change
// once per optimiser step and are split once, so this kernel only splits the ACTIVATION operand.
//
// Why a second kernel.  In the 128 x 128 kernel (gemm.hip) a K-step of 24 MFMAs per wave carries the split VALU and the
// ds_write of BOTH operands, 12 fragment reads, a barrier - the matrix pipe ends up ~1/3 busy.  Here a workgroup is
// 4 waves (one per SIMD) on a 128 x 256 tile (waves 2 x 2, each 64 x 128 = 2 x 4 accumulators of 32 x 32), K-step 16,
// two LDS stages of 36.4 KB -> TWO independent workgroups per CU (2 waves per SIMD, 256 VGPRs each):
//   * B never touches a VGPR or the VALU: its planes are laid out in HBM exactly like the LDS image, so a K-step's B tile
//     is 6 slabs of 256 x 16 B that global_load_lds (LDS-DMA, 16 B per lane) copies straight into LDS;
//   * A: 2 float4 loads per thread and K-step (a row's 64 B by 4 lanes), split into 3 bf16 terms in registers, 6
//     ds_write_b64 - half the staging work per MFMA of the 128 x 128 kernel, and 48 MFMAs per wave and barrier
//     instead of 24;
//   * fragment reads per MFMA: 18 / 48 instead of 12 / 24;
//   * two independent workgroups per CU: one's barrier waits and fragment-read latency sit under the other's MFMAs
//     (measured and dropped: delaying the second-slot workgroups by half a tile so that the epilogues - a 128 KB C tile
//     through a 20 GB/s per-CU share of HBM write bandwidth - interleave: no gain; padding the weight-gradient kernel's
//     LDS so that only one of its workgroups fits a CU beside the critical path: step time unchanged, -10 % stand-alone).  (First version: ONE 8-wave
//     workgroup per CU on a 256 x 256 tile - per-tile fixed cost 32-38 us against 39 us per 512 of K: 120 TF at K = 512.)
// LDS image per operand and term: [k-group of 8][row][8 k] bf16, one conflict-free ds_read_b128 per MFMA fragment (lanes
// 0-31 = rows of k-group 0, lanes 32-63 = k-group 1).  The A k-groups are skewed by 64 B so that the 16-lane groups of a
// ds_write_b64 (4 rows x 4 k-quads) cover all 32 banks once.
#include "gemm.h"
#include <cstdlib>

namespace npvp {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int TM, int TN, int WM, int WN, bool ROWSTATS>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 2) void gemm_wide_kernel(GemmParams p) {
  constexpr int NW = WM * WN, THREADS = 64 * NW;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int APASS = BM / (THREADS / 4);           // float4 loads per thread and K-step (rows of THREADS/4 per pass)
  static_assert(APASS == 2 && BN % 64 == 0, "A staging is written for two row passes");
  constexpr int KGS_A = BM * 16 + 64, A_PLANE = 2 * KGS_A, A_BYTES = 3 * A_PLANE;
  constexpr int KGS_B = BN * 16, B_PLANE = 2 * KGS_B, B_BYTES = 3 * B_PLANE;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int CPS = BN / 64;                       // 1 KB glds chunks per (term, k-group) slab
  constexpr int NCHUNK = 6 * CPS, CPW = (NCHUNK + NW - 1) / NW;
  __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];

  int tile_m, tile_n;
  tile_of_block_unsplit(p, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);     // wave-uniform: SGPR
  const int wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
  const int nk = p.K >> 4;
  const unsigned int cpeek = amax_peek_block(p.c_amax);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

  // ---- A staging: thread -> (rows rl and rl + BM/2, k-quad); rows past the edge are clamped (they only feed output rows
  // that are never stored)
  const int quad = t & 3, rl = t >> 2;
  const float* a_src0 = p.A + (long long)min(m0 + rl, p.M - 1) * p.lda + 4 * quad;
  const float* a_src1 = p.A + (long long)min(m0 + rl + BM / 2, p.M - 1) * p.lda + 4 * quad;
  const int a_dst = (quad >> 1) * KGS_A + (quad & 1) * 8 + rl * 16;
  // ---- B staging: wave -> chunks c = wave + NW i of the step's 6 slabs
  const uint4* b_src[CPW];
  int b_dst[CPW];
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int c = wave + NW * i;
    const int slab = c / CPS, part = c - slab * CPS, s = slab >> 1, kg = slab & 1;
    const int col = min(n0 + part * 64 + lane, p.N - 1);
    b_src[i] = reinterpret_cast<const uint4*>(p.b_pre) + (long long)s * (p.b_pre_plane >> 3) + (long long)kg * p.N + col;
    b_dst[i] = A_BYTES + s * B_PLANE + kg * KGS_B + part * 1024;
  }
  const long long b_step = 2ll * p.N;                // uint4 per K-step (two k-groups)

  // ---- fragment slots of this lane
  const int fa_off = h * KGS_A + (wm * TM * 32 + r) * 16;
  const int fb_off = A_BYTES + h * KGS_B + (wn * TN * 32 + r) * 16;

  f32x4 ra0, ra1;
#define NPVP_W_ALOAD(KT)                                                                 \
  { const int k_ = min((KT), nk - 1) << 4; ra0 = *reinterpret_cast<const f32x4*>(a_src0 + k_); ra1 = *reinterpret_cast<const f32x4*>(a_src1 + k_); }
#define NPVP_W_ASTORE(ST, V, ROWOFF)                                                     \
  { f32x4 v_ = (V);                                                                      \
    _Pragma("unroll") for (int s_ = 0; s_ < 3; ++s_) {                                   \
      bf16x4 q_;                                                                         \
      q_[0] = (__bf16)v_[0]; q_[1] = (__bf16)v_[1]; q_[2] = (__bf16)v_[2]; q_[3] = (__bf16)v_[3]; \
      *reinterpret_cast<bf16x4*>((ST) + s_ * A_PLANE + a_dst + (ROWOFF)) = q_;           \
      v_[0] -= (float)q_[0]; v_[1] -= (float)q_[1]; v_[2] -= (float)q_[2]; v_[3] -= (float)q_[3]; \
    } }
#define NPVP_W_BLOAD(ST, KT)                                                             \
  { const long long ko_ = (long long)min((KT), nk - 1) * b_step;                         \
    _Pragma("unroll") for (int i_ = 0; i_ < CPW; ++i_)                                   \
      if (NCHUNK % NW == 0 || wave + NW * i_ < NCHUNK)                                                    \
        __builtin_amdgcn_global_load_lds((gptr_t)(b_src[i_] + ko_), (lptr_t)((ST) + b_dst[i_]), 16, 0, 0); }

  // prologue: tile 0 -> stage 0, A tile 1 -> registers
  NPVP_W_BLOAD(lds, 0)
  NPVP_W_ALOAD(0)
  NPVP_W_ASTORE(lds, ra0, 0)
  NPVP_W_ASTORE(lds, ra1, (BM / 2) * 16)
  NPVP_W_ALOAD(1)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // one K-step: MFMAs of tile KT on stage CUR; A tile KT+1 (registers) is split and written to stage NXT, B tile KT+1 is
  // DMA'd into NXT, A tile KT+2 is loaded into the registers; everything issued here has landed at the step's barrier.
#define NPVP_W_STEP(KT, CUR, NXT)                                                                          \
  {                                                                                                        \
    const char* st_ = lds + (CUR) * STAGE;                                                                 \
    char* nx_ = lds + (NXT) * STAGE;                                                                       \
    /* the registers of A tile KT+1 were loaded a step ago: take the compiler's wait for them HERE, before the LDS-DMA */ \
    /* is issued (beside an outstanding global_load_lds hipcc waits vmcnt(0) at the first use of any loaded register) */ \
    asm volatile("" : "+v"(ra0), "+v"(ra1));                                                               \
    NPVP_W_BLOAD(nx_, (KT) + 1)                                                                            \
    bf16x8 fb_[3][TN];                                                                                     \
    _Pragma("unroll") for (int s_ = 0; s_ < 3; ++s_)                                                       \
      _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_)                                                    \
        fb_[s_][j_] = *reinterpret_cast<const bf16x8*>(st_ + fb_off + s_ * B_PLANE + j_ * 512);            \
    _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) {                                                    \
      bf16x8 fa_[3];                                                                                       \
      _Pragma("unroll") for (int s_ = 0; s_ < 3; ++s_)                                                     \
        fa_[s_] = *reinterpret_cast<const bf16x8*>(st_ + fa_off + s_ * A_PLANE + i_ * 512);                \
      if (i_ == 0) NPVP_W_ASTORE(nx_, ra0, 0)                                                              \
      /* the loads of A tile KT+2 go out here, a quarter into the step (pinned: the scheduler would sink them to the */ \
      /* end of the step, right in front of the wait) */                                                   \
      if (i_ == 1) { NPVP_W_ASTORE(nx_, ra1, (BM / 2) * 16) NPVP_W_ALOAD((KT) + 2) __builtin_amdgcn_sched_barrier(0); } \
      /* smallest terms first */                                                                           \
      _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[1], fb_[1][j_], acc[i_][j_], 0, 0, 0); \
      _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[0], fb_[2][j_], acc[i_][j_], 0, 0, 0); \
      _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[2], fb_[0][j_], acc[i_][j_], 0, 0, 0); \
      _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[0], fb_[1][j_], acc[i_][j_], 0, 0, 0); \
      _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[1], fb_[0][j_], acc[i_][j_], 0, 0, 0); \
      _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[0], fb_[0][j_], acc[i_][j_], 0, 0, 0); \
    }                                                                                                      \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                            \
    __builtin_amdgcn_s_barrier();                                                                          \
  }

  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    NPVP_W_STEP(kt, 0, 1)
    NPVP_W_STEP(kt + 1, 1, 0)
  }
  if (kt < nk) NPVP_W_STEP(kt, 0, 1)
#undef NPVP_W_STEP
#undef NPVP_W_BLOAD
#undef NPVP_W_ASTORE
#undef NPVP_W_ALOAD

  const int row_base = m0 + wm * TM * 32, col_base = n0 + wn * TN * 32;
  float cmax = 0.f;
  float* scr = reinterpret_cast<float*>(lds) + 4 + wave * EPI_FLOATS;      // per-wave transposition scratch (gemm.h)
  if constexpr (ROWSTATS) {
    static_assert(!ROWSTATS || (TN % 2 == 0 && TM % 2 == 0), "frame statistics ride on 64 x 64 accumulator blocks");
#pragma unroll
    for (int i = 0; i < TM; i += 2)
#pragma unroll
      for (int j = 0; j < TN; j += 2)
        epilogue_rowstats_block(p, acc[i][j], acc[i][j + 1], acc[i + 1][j], acc[i + 1][j + 1], row_base + i * 32, col_base + j * 32, lane, scr, cmax, p.alpha);
  } else {
    const unsigned long long seed = (p.seed && p.drop.thresh) ? *p.seed : 0ull;
    static_for<0, TM>([&](auto i) __attribute__((always_inline)) {
      const float4 rowsc = epilogue_row_scales(p, seed, row_base + i * 32, lane);
      static_for<0, TN>([&](auto j) __attribute__((always_inline)) {
        epilogue_tile(p, acc[i][j], row_base + i * 32, col_base + j * 32, lane, scr, 0, seed, cmax, rowsc, p.alpha);
      });
    });
  }
  amax_slot_commit_block(p.c_amax, cmax, reinterpret_cast<float*>(lds), cpeek);      // (the stages are idle after the K loop's last barrier)
}

// =====================================================================================================
// Weight gradients on the same wave / tile geometry:  dW[M,N] = A^T B  with  A = dy [K][M], B = x [K][N]  (K = token
// rows, M = out features, N = in features; both operands "k-major": a K-step's tile is 16 full rows of each matrix).
//
// The MFMA fragment of lane (column c, k-half h) is 8 consecutive k of ONE column - strided in memory.  The 128 x 128
// kernel gathers it with 8 dword loads per (column, k-group); here the tiles are staged ROW-major, exactly as they lie in
// memory (float4 loads of 512 B / 1 KB rows, 6 per thread and K-step instead of 24 dword loads), split into three bf16
// planes [16 k][columns], and the fragments are read with ds_read_b64_tr_b16, the LDS transposing read: a 16-lane group
// reads a 4 k x 16 column block and each lane receives its column's 4 k (two reads per fragment).  Bank conflicts: a
// 32-lane half reads 4 k-rows x 64 B; rows are 256 / 512 B apart (the same banks), so the 64-B granule index of row k is
// XORed with k & 3 - on the ds_write side too - and the four rows land in the four bank quadrants.
// Split-K over blockIdx.y (each K-chunk owned by one XCD), partial tiles to the workspace, column sums of dy (the bias
// gradient) from the staging registers of the blocks in tile column 0.
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 lds_read_tr_pair(const char* a, int second_off) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + second_off));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 2) void gemm_wgrad_wide_kernel(GemmParams p) {
  constexpr int NW = WM * WN, THREADS = 64 * NW;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  static_assert(THREADS == 256 && BM == 128 && BN == 256, "staging maps are written for 256 threads on a 128 x 256 tile");
  constexpr int ROWA = BM * 2, ROWB = BN * 2;                       // bytes per k-row of a plane
  constexpr int A_PLANE = 16 * ROWA, B_PLANE = 16 * ROWB, A_BYTES = 3 * A_PLANE, B_BYTES = 3 * B_PLANE;
  constexpr int STAGE = A_BYTES + B_BYTES;
  __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];

  // ---- (K-chunk z, tile): with splits % 8 == 0 chunk z is owned by XCD z % 8 (its tiles share each row block in ONE L2)
  int z, tl;
  {
    const int tiles = gridDim.x;
    if (p.splits > 1 && (p.splits & 7) == 0) {
      const int lin = blockIdx.x + tiles * blockIdx.y, xcd = lin & 7, slot = lin >> 3;
      const int g = slot / tiles;
      tl = slot - g * tiles; z = g * 8 + xcd;
    } else { z = blockIdx.y; tl = blockIdx.x; }
  }
  const int tile_m = tl / p.tiles_n, tile_n = tl - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const float* A = p.A + (long long)z * p.K * p.lda;
  const float* B = p.B + (long long)z * p.K * p.ldb;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
  const int nk = p.K >> 4;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

  // ---- staging maps.  A: float4 f = t, t + 256 -> k-row t >> 5 (+8), column quad t & 31.  B: f = t + 256 i -> k-row
  // (t >> 6) + 4 i, column quad t & 63.  Columns past the edge are clamped (they feed outputs that are never stored).
  const int ka = t >> 5, cqa = t & 31, kb = t >> 6, cqb = t & 63;
  const float* a_src = A + (long long)ka * p.lda + min(m0 + 4 * cqa, p.M - 4);
  const float* b_src = B + (long long)kb * p.ldb + min(n0 + 4 * cqb, p.N - 4);
  const long long a_row8 = 8 * p.lda, b_row4 = 4 * p.ldb, a_step = 16 * p.lda, b_step = 16 * p.ldb;
  const int a_dst = ka * ROWA + ((8 * cqa) ^ ((ka & 3) << 6));
  const int b_dst = A_BYTES + kb * ROWB + ((8 * cqb) ^ ((kb & 3) << 6));
  // ---- fragment addresses (transposing reads): 16-lane group G -> k-half h = G >> 1, columns 16 (G & 1) ..+15; lane
  // 4 q + pp of the group supplies k-row 8 h + q, columns 4 pp ..+3
  const int q = (lane >> 2) & 3, pp = lane & 3, cg = 16 * ((lane >> 4) & 1) + 4 * pp;
  int fa[TM], fb[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) fa[i] = (8 * h + q) * ROWA + (((wm * TM * 32 + i * 32 + cg) * 2) ^ (q << 6));
#pragma unroll
  for (int j = 0; j < TN; ++j) fb[j] = A_BYTES + (8 * h + q) * ROWB + (((wn * TN * 32 + j * 32 + cg) * 2) ^ (q << 6));

  const bool want_cs = p.colsum && tile_n == 0;
  f32x4 cs = {0.f, 0.f, 0.f, 0.f};
  f32x4 ra[2], rb[4];
#define NPVP_G_LOAD(KT)                                                                                     \
  { const int kt_ = min((KT), nk - 1);                                                                      \
    const float* pa_ = a_src + (long long)kt_ * a_step; const float* pb_ = b_src + (long long)kt_ * b_step; \
    ra[0] = *reinterpret_cast<const f32x4*>(pa_); ra[1] = *reinterpret_cast<const f32x4*>(pa_ + a_row8);    \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) rb[i_] = *reinterpret_cast<const f32x4*>(pb_ + i_ * b_row4); }
#define NPVP_G_STORE(DST, V, PLANE)                                                                         \
  { f32x4 v_ = (V);                                                                                         \
    _Pragma("unroll") for (int s_ = 0; s_ < 3; ++s_) {                                                      \
      bf16x4 q_;                                                                                            \
      q_[0] = (__bf16)v_[0]; q_[1] = (__bf16)v_[1]; q_[2] = (__bf16)v_[2]; q_[3] = (__bf16)v_[3];           \
      *reinterpret_cast<bf16x4*>((DST) + s_ * (PLANE)) = q_;                                                \
      v_[0] -= (float)q_[0]; v_[1] -= (float)q_[1]; v_[2] -= (float)q_[2]; v_[3] -= (float)q_[3];           \
    } }
#define NPVP_G_STORE_A(ST) { NPVP_G_STORE((ST) + a_dst, ra[0], A_PLANE) NPVP_G_STORE((ST) + a_dst + 8 * ROWA, ra[1], A_PLANE) }
#define NPVP_G_STORE_B(ST, I) NPVP_G_STORE((ST) + b_dst + (I) * 4 * ROWB, rb[I], B_PLANE)

  NPVP_G_LOAD(0)
  if (want_cs) cs += ra[0] + ra[1];
  NPVP_G_STORE_A(lds)
  NPVP_G_STORE_B(lds, 0) NPVP_G_STORE_B(lds, 1) NPVP_G_STORE_B(lds, 2) NPVP_G_STORE_B(lds, 3)
  NPVP_G_LOAD(1)
  __syncthreads();

#define NPVP_G_STEP(KT, CUR, NXT)                                                                            \
  {                                                                                                          \
    const char* st_ = lds + (CUR) * STAGE;                                                                   \
    char* nx_ = lds + (NXT) * STAGE;                                                                         \
    if (want_cs && (KT) + 1 < nk) cs += ra[0] + ra[1];                                                       \
    bf16x8 fa_[TM][3];                                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_)                                                        \
      _Pragma("unroll") for (int s_ = 0; s_ < 3; ++s_) fa_[i_][s_] = lds_read_tr_pair(st_ + fa[i_] + s_ * A_PLANE, 4 * ROWA); \
    _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) {                                                      \
      bf16x8 fb_[3];                                                                                         \
      _Pragma("unroll") for (int s_ = 0; s_ < 3; ++s_) fb_[s_] = lds_read_tr_pair(st_ + fb[j_] + s_ * B_PLANE, 4 * ROWB); \
      if (j_ == 0) { NPVP_G_STORE_A(nx_) NPVP_G_STORE_B(nx_, 0) }                                            \
      if (j_ == 1) { NPVP_G_STORE_B(nx_, 1) NPVP_G_STORE_B(nx_, 2) NPVP_G_STORE_B(nx_, 3)                    \
                     NPVP_G_LOAD((KT) + 2) __builtin_amdgcn_sched_barrier(0); }                              \
      _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[i_][1], fb_[1], acc[i_][j_], 0, 0, 0); \
      _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[i_][0], fb_[2], acc[i_][j_], 0, 0, 0); \
      _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[i_][2], fb_[0], acc[i_][j_], 0, 0, 0); \
      _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[i_][0], fb_[1], acc[i_][j_], 0, 0, 0); \
      _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[i_][1], fb_[0], acc[i_][j_], 0, 0, 0); \
      _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[i_][0], fb_[0], acc[i_][j_], 0, 0, 0); \
    }                                                                                                        \
    __syncthreads();                                                                                         \
  }

  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    NPVP_G_STEP(kt, 0, 1)
    NPVP_G_STEP(kt + 1, 1, 0)
  }
  if (kt < nk) NPVP_G_STEP(kt, 0, 1)
#undef NPVP_G_STEP
#undef NPVP_G_STORE_B
#undef NPVP_G_STORE_A
#undef NPVP_G_STORE
#undef NPVP_G_LOAD

  if (want_cs) {          // the 8 threads with the same column quad (t & 31) hold partial sums: fixed-order reduction in LDS
    float* red = reinterpret_cast<float*>(lds);
    *reinterpret_cast<f32x4*>(red + (t >> 5) * 128 + cqa * 4) = cs;
    __syncthreads();
    if (t < 128 && m0 + t < p.M) {
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) sum += red[i * 128 + t];
      store_colsum(p, (long long)z * p.M + m0 + t, sum);
    }
    __syncthreads();                     // `red` is about to become the epilogue's scratch
  }
  float* scr = reinterpret_cast<float*>(lds) + wave * EPI_FLOATS;
  const unsigned long long seed = 0ull;
  const int row_base = m0 + wm * TM * 32, col_base = n0 + wn * TN * 32;
  float cmax = 0.f;
  const float4 no_rowsc = make_float4(1.f, 1.f, 1.f, 1.f);         // (a weight gradient's epilogue carries no mask)
  static_for<0, TM>([&](auto i) __attribute__((always_inline)) {
    static_for<0, TN>([&](auto j) __attribute__((always_inline)) {
      epilogue_tile(p, acc[i][j], row_base + i * 32, col_base + j * 32, lane, scr, z, seed, cmax, no_rowsc, p.alpha);
    });
  });
}

// split count of the wide weight-gradient kernel: ~512 workgroups (2 per CU), >= 16 K-steps per split
int wide_wgrad_splits(int M, int N, int K) {
  // 0 = not taken.  Below ~32 K token rows the 128 x 128 kernel (3 workgroups per CU, finer tiles) is as fast or faster:
  // measured 168 vs 166 TF at 20 480 rows, 144 vs 134 TF at 8 192 rows, 179 vs 189 TF at 114 688 rows.
  if ((K & 15) || M < 64 || N < 128 || K < 32768) return 0;
  const int tiles = ((M + 127) / 128) * ((N + 255) / 256);
  int s = (512 + tiles - 1) / tiles;        // (256 / 128 workgroups: c2 step 397 / 366 ms against 336 ms; 1024 / 2048: 335.4 / 338.2 against 332.0)
  const int maxs = K / 256;
  if (s > maxs) s = maxs;
  if (s > 64) s = 64;
  if (s >= 8) s &= ~7;                                         // multiples of 8: one K-chunk per XCD
  while (s > 1 && (K % (s * 16)) != 0) --s;
  return s < 1 ? 1 : s;
}

bool launch_gemm_wgrad_wide(GemmParams& p, int splits, hipStream_t stream) {
  p.tiles_m = (p.M + 127) / 128;
  p.tiles_n = (p.N + 255) / 256;
  dim3 grid(p.tiles_m * p.tiles_n, splits), block(256);
  NPVP_LAUNCH((gemm_wgrad_wide_kernel<2, 4, 2, 2>), grid, block, 0, stream, p);
  return true;
}

// Which kernel for an [M, N] output: estimated time = rounds over the CUs x relative cost of one round.  The wide kernel
// runs 2 workgroups of 128 x 256 per CU, the 128 x 128 kernel 3 per CU; eff = measured relative MFMA throughput on full rounds.
static bool wide_pays(int M, int N) {
  auto rounds = [](double tiles, double slots) { return (double)(long long)((tiles + slots - 1) / slots); };
  const double tw = ((M + 127) / 128) * (double)((N + 255) / 256), t128 = ((M + 127) / 128) * (double)((N + 127) / 128);
  const double cw = rounds(tw, 512) * 2.0 / 1.00;
  const double c128 = rounds(t128, 768) * 1.5 / 0.80;
  return cw <= c128;
}

// Which instantiation takes an [M, N] x K forward / dgrad problem with pre-split planes: 0 none (128 x 128 gemm_split_db_kernel),
// 1 = 128 x 256 tiles, 2 = 128 x 128 tiles of the same kernel for small outputs (the per-GPU shards of the multi-GPU configs:
// 8 192 token rows: 173 / 169 TF forward / dgrad against 155 / 153 TF of the older 128 x 128 kernel; 2 048 rows: 70 / 90
// against 58 / 81).  Also behind npvp_gemm_kernel_id.
static int wide_variant(int M, int N, int K) {
  if ((K & 15) || (N & 7) || M < 128) return 0;
  if (wide_pays(M, N)) return 1;
  // 128 x 128 tiles of this kernel run 2 workgroups per CU, the older 128 x 128 kernel 3: this one wins while all its tiles are
  // resident at once (<= 512: 6 144 rows 154 / 149 TF against 132 / 140), beyond that the other's third slot does (20 480 rows:
  // 179 against 182 TF)
  const int tiles128 = ((M + 127) / 128) * (N / 128);
  return (N % 128 == 0 && tiles128 <= 512) ? 2 : 0;
}
bool gemm_wide_takes(int M, int N, int K) { return wide_variant(M, N, K) != 0; }
int gemm_wide_variant(int M, int N, int K) { return wide_variant(M, N, K); }

bool launch_gemm_wide(GemmParams& p, hipStream_t stream) {
  if (!p.b_pre || p.splits != 1 || p.colsum || ((uintptr_t)p.b_pre & 15) != 0) return false;
  const int v = wide_variant(p.M, p.N, p.K);
  if (v == 0 || (p.rowstats && (p.N % 64 != 0 || p.M % 64 != 0))) return false;
  const int bn = v == 1 ? 256 : 128;
  p.tiles_m = (p.M + 127) / 128;
  p.tiles_n = (p.N + bn - 1) / bn;
  p.colgroups = pick_colgroups((long long)p.N * p.K * 6, p.tiles_m, p.tiles_n);
  dim3 grid(p.tiles_m * p.tiles_n), block(256);
  if (v == 1) {
    if (p.rowstats) NPVP_LAUNCH((gemm_wide_kernel<2, 4, 2, 2, true>), grid, block, 0, stream, p);
    else NPVP_LAUNCH((gemm_wide_kernel<2, 4, 2, 2, false>), grid, block, 0, stream, p);
  } else {
    if (p.rowstats) NPVP_LAUNCH((gemm_wide_kernel<2, 2, 2, 2, true>), grid, block, 0, stream, p);
    else NPVP_LAUNCH((gemm_wide_kernel<2, 2, 2, 2, false>), grid, block, 0, stream, p);
  }
  return true;
}

}  // namespace npvp
