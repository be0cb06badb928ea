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
//   * the two workgroups of a CU run out of phase: one's prologue, epilogue (a 128 KB C tile leaves through a 20 GB/s
//     per-CU share of HBM write bandwidth) and barrier waits sit under the other's MFMAs.  (First version: ONE 8-wave
//     workgroup per CU on a 256 x 256 tile - per-tile fixed cost 32-38 us against 39 us per 512 of K: 120 TF at K = 512.)
// LDS image per operand and term: [k-group of 8][row][8 k] bf16, one conflict-free ds_read_b128 per MFMA fragment (lanes
// 0-31 = rows of k-group 0, lanes 32-63 = k-group 1).  The A k-groups are skewed by 64 B so that the 16-lane groups of a
// ds_write_b64 (4 rows x 4 k-quads) cover all 32 banks once.
#include "gemm.h"

// measurement builds only (NPVP_HIPCC_EXTRA=-DNPVP_WIDE_ABL=n on the GPU box; results INVALID): 1 = epilogue without its
// global stores, 2 = no K loop
#ifndef NPVP_WIDE_ABL
#define NPVP_WIDE_ABL 0
#endif

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

  int kt = (NPVP_WIDE_ABL & 2) ? nk : 0;
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
#if defined(__HIP_DEVICE_COMPILE__)
  if (NPVP_WIDE_ABL & 1) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" :: "v"(acc[i][j]));
    return;
  }
#endif
  if constexpr (ROWSTATS) {
    static_assert(!ROWSTATS || (TN % 2 == 0 && TM % 2 == 0), "frame statistics ride on 64 x 64 accumulator blocks");
#pragma unroll
    for (int i = 0; i < TM; i += 2)
#pragma unroll
      for (int j = 0; j < TN; j += 2)
        epilogue_rowstats_block(p, acc[i][j], acc[i][j + 1], acc[i + 1][j], acc[i + 1][j + 1], row_base + i * 32, col_base + j * 32, r, h);
  } else {
    const unsigned long long seed = (p.seed && p.drop.thresh) ? *p.seed : 0ull;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) epilogue_tile(p, acc[i][j], row_base + i * 32, col_base + j * 32, r, h, 0, seed);
  }
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

bool launch_gemm_wide(GemmParams& p, hipStream_t stream) {
  if (!p.b_pre || p.splits != 1 || (p.K & 15) || (p.N & 7) || p.colsum || p.M < 128) return false;
  if (((uintptr_t)p.b_pre & 15) != 0) return false;
  if (p.rowstats && (p.N % 64 != 0 || p.M % 64 != 0)) return false;
  if (!wide_pays(p.M, p.N)) return false;
  p.tiles_m = (p.M + 127) / 128;
  p.tiles_n = (p.N + 255) / 256;
  p.colgroups = pick_colgroups((long long)p.N * p.K * 6, p.tiles_m, p.tiles_n);
  dim3 grid(p.tiles_m * p.tiles_n), block(256);
  if (p.rowstats) hipLaunchKernelGGL((gemm_wide_kernel<2, 4, 2, 2, true>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((gemm_wide_kernel<2, 4, 2, 2, false>), grid, block, 0, stream, p);
  return true;
}

}  // namespace npvp
