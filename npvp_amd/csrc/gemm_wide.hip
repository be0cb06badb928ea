// Wide-tile form of the split-precision GEMM for the large forward / dgrad shapes (token rows x 512..2048 columns):
//
//     C[M,N] = epilogue( A[M,K] (fp32, k contiguous) * B ),   B = pre-split bf16 planes [term][K/8][N][8 k]
//
// i.e. y = x W^T with the F planes of W and dx = dy W with the D planes of W (npvp_split_weight): the weights change
// once per optimiser step and are split once, so this kernel only splits the ACTIVATION operand.
//
// Why a second kernel.  In the 128 x 128 kernel (gemm.hip) a K-step of 24 MFMAs per wave carries the split VALU and the
// ds_write of BOTH operands, 12 fragment reads, a barrier - the matrix pipe ends up ~1/3 busy.  Here a workgroup is
// 8 waves (2 per SIMD) on a 256 x BN tile (BN = 256: waves 2 x 4, each 128 x 64 = 4 x 2 accumulators of 32 x 32;
// BN = 128: waves 4 x 2, each 64 x 64), K-step 16, two LDS stages:
//   * B never touches a VGPR or the VALU: its planes are laid out in HBM exactly like the LDS image, so a K-step's B tile
//     is 6 slabs of BN x 16 B that global_load_lds (LDS-DMA, 16 B per lane) copies straight into LDS;
//   * A: 2 float4 loads per thread and K-step (a row's 64 B by 4 lanes), split into 3 bf16 terms in registers, 6
//     ds_write_b64 - half the staging work per MFMA of the 128 x 128 kernel, and per wave 48 (BN = 256) MFMAs per
//     barrier instead of 24;
//   * fragment reads per MFMA: 18 / 48 (BN = 256) instead of 12 / 24.
// LDS image per operand and term: [k-group of 8][row][8 k] bf16, one conflict-free ds_read_b128 per MFMA fragment (lanes
// 0-31 = rows of k-group 0, lanes 32-63 = k-group 1).  The A k-groups are skewed by 64 B so that the 16-lane groups of a
// ds_write_b64 (4 rows x 4 k-quads) cover all 32 banks once.
#include "gemm.h"

namespace npvp {

constexpr int WIDE_THREADS = 512;

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int TM, int TN, int WM, int WN, bool ROWSTATS>
__global__ __launch_bounds__(WIDE_THREADS, 2) void gemm_wide_kernel(GemmParams p) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  static_assert(BM == 256 && WM * WN == 8, "8 waves on a 256-row tile");
  constexpr int KGS_A = BM * 16 + 64, A_PLANE = 2 * KGS_A, A_BYTES = 3 * A_PLANE;
  constexpr int KGS_B = BN * 16, B_PLANE = 2 * KGS_B, B_BYTES = 3 * B_PLANE;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int CPS = BN / 64;                       // 1 KB glds chunks per (term, k-group) slab
  constexpr int NCHUNK = 6 * CPS, CPW = (NCHUNK + 7) / 8;
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

  // ---- A staging: thread -> (rows rl and rl + 128, k-quad); rows past the edge are clamped (they only feed output rows
  // that are never stored)
  const int quad = t & 3, rl = t >> 2;
  const float* a_src0 = p.A + (long long)min(m0 + rl, p.M - 1) * p.lda + 4 * quad;
  const float* a_src1 = p.A + (long long)min(m0 + rl + 128, p.M - 1) * p.lda + 4 * quad;
  const int a_dst = (quad >> 1) * KGS_A + (quad & 1) * 8 + rl * 16;
  // ---- B staging: wave -> chunks c = wave + 8 i of the step's 6 slabs
  const uint4* b_src[CPW];
  int b_dst[CPW];
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int c = wave + 8 * i;
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
      if (wave + 8 * i_ < NCHUNK)                                                        \
        __builtin_amdgcn_global_load_lds((gptr_t)(b_src[i_] + ko_), (lptr_t)((ST) + b_dst[i_]), 16, 0, 0); }

  // prologue: tile 0 -> stage 0, A tile 1 -> registers
  NPVP_W_BLOAD(lds, 0)
  NPVP_W_ALOAD(0)
  NPVP_W_ASTORE(lds, ra0, 0)
  NPVP_W_ASTORE(lds, ra1, 128 * 16)
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
      if (i_ == 1) { NPVP_W_ASTORE(nx_, ra1, 128 * 16) NPVP_W_ALOAD((KT) + 2) __builtin_amdgcn_sched_barrier(0); } \
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
  if constexpr (ROWSTATS) {
    static_assert(!ROWSTATS || (TN == 2 && TM % 2 == 0), "frame statistics ride on 64 x 64 accumulator blocks");
#pragma unroll
    for (int i = 0; i < TM; i += 2)
      epilogue_rowstats_block(p, acc[i][0], acc[i][TN - 1], acc[i + 1][0], acc[i + 1][TN - 1], row_base + i * 32, col_base, r, h);
  } else {
    const unsigned long long seed = (p.seed && p.drop.thresh) ? *p.seed : 0ull;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) epilogue_tile(p, acc[i][j], row_base + i * 32, col_base + j * 32, r, h, 0, seed);
  }
}

// Which tiling for an [M, N] output: estimated time = rounds over the 256 CUs x relative cost of one round.  A wide
// workgroup owns its CU (148 KB / 98 KB... LDS); the 128 x 128 kernel runs 3 workgroups per CU, each at a third of the CU.
// eff = measured relative MFMA throughput of the three kernels on full rounds (256 x 256 = 1).
static int pick_wide(int M, int N) {
  const double CUS = 256.0;
  auto rounds = [&](double tiles, double per_cu) { return (double)(long long)((tiles + CUS * per_cu - 1) / (CUS * per_cu)); };
  const double t256 = ((M + 255) / 256) * (double)((N + 255) / 256), t128w = ((M + 255) / 256) * (double)((N + 127) / 128);
  const double t128 = ((M + 127) / 128) * (double)((N + 127) / 128);
  const double c256 = rounds(t256, 1) * 1.0 / 1.00;
  const double c128w = rounds(t128w, 1) * 0.5 / 0.90;
  const double c128 = rounds(t128, 3) * 0.75 / 0.72;
  if (c256 <= c128w && c256 <= c128) return 256;
  if (c128w <= c128) return 128;
  return 0;
}

bool launch_gemm_wide(GemmParams& p, hipStream_t stream) {
  if (!p.b_pre || p.splits != 1 || (p.K & 15) || (p.N & 7) || p.colsum || p.M < 256) return false;
  if (((uintptr_t)p.b_pre & 15) != 0) return false;
  int bn = pick_wide(p.M, p.N);
  if (p.rowstats && (p.N % 64 != 0 || p.M % 64 != 0)) return false;
  if (bn == 0) return false;
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + bn - 1) / bn;
  p.colgroups = pick_colgroups((long long)p.N * p.K * 6, p.tiles_m, p.tiles_n);
  dim3 grid(p.tiles_m * p.tiles_n), block(WIDE_THREADS);
  if (bn == 256) {
    if (p.rowstats) hipLaunchKernelGGL((gemm_wide_kernel<4, 2, 2, 4, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((gemm_wide_kernel<4, 2, 2, 4, false>), grid, block, 0, stream, p);
  } else {
    if (p.rowstats) hipLaunchKernelGGL((gemm_wide_kernel<2, 2, 4, 2, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((gemm_wide_kernel<2, 2, 4, 2, false>), grid, block, 0, stream, p);
  }
  return true;
}

}  // namespace npvp
