// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered
// fma chain) with fused bias / GELU / ReLU / dropout / residual / activation-gradient
// epilogues.  This is the kernel that carries >99 % of the predictor's FLOPs
// (SURVEY 2b K3-K7): every nn.Linear, 1x1 conv and attention in/out projection of
// ref/models/VidHRFormer.py:71-72,184-185,270,345,364 and their backward passes.
//
//   C[M,N] = epilogue( op(A)[M,K] * op(B)[K,N] )
//
// Operand storage (no transposes are ever materialised):
//   a_kc = 1 : A stored [M][K] (k contiguous)     a_kc = 0 : A stored [K][M]
//   b_kc = 1 : B stored [N][K] (k contiguous)     b_kc = 0 : B stored [K][N]
//   forward  y = x W^T          : a_kc=1, b_kc=1
//   dgrad    dx = dy W          : a_kc=1, b_kc=0
//   wgrad    dW = dy^T x        : a_kc=0, b_kc=0   (K = token rows, split-K)
//
// Tiling: 128x128x32 block, 256 threads = 4 waves (2x2), each wave a 64x64 tile as
// 2x2 MFMA 32x32 accumulators (64 acc VGPRs).  Operands are staged global -> VGPR
// (float4, prefetched one K-step ahead) -> LDS in K-major [k][row] layout so each
// MFMA operand is ONE conflict-free ds_read_b32 per lane (lane l reads row l&31 of
// k-row l>>5).  LDS is double buffered: one barrier per K-step.  blockIdx is remapped
// so that the 8 XCDs each walk a contiguous run of tiles (B = the weight stays in the
// XCD's L2; each A row-panel is fetched by one XCD).
#include "gemm.h"
#include <type_traits>

namespace npvp {

constexpr int BM = 128, BN = 128, BK = 32, GEMM_THREADS = 256;

// ---- operand staging: one 128 x 32 (rows x k) tile -----------------------------------
template <bool KC> struct Stager;

template <> struct Stager<true> {   // source [rows][K], k contiguous
  static constexpr int LD = 129;
  float4 r[4];
  __device__ __forceinline__ void load(const float* src, long long ld, int row0, int nrows, int k0) {
    const int t = threadIdx.x;
    const int kc = (t & 7) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row0 + (t >> 3) + 32 * i;
      r[i] = (row < nrows) ? ld4(src + (long long)row * ld + k0 + kc) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void store(float* s) const {
    const int t = threadIdx.x;
    const int kc = (t & 7) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (t >> 3) + 32 * i;
      s[(kc + 0) * LD + row] = r[i].x;
      s[(kc + 1) * LD + row] = r[i].y;
      s[(kc + 2) * LD + row] = r[i].z;
      s[(kc + 3) * LD + row] = r[i].w;
    }
  }
};

template <> struct Stager<false> {  // source [K][rows], rows contiguous
  static constexpr int LD = 132;
  float4 r[4];
  __device__ __forceinline__ void load(const float* src, long long ld, int row0, int nrows, int k0) {
    const int t = threadIdx.x;
    const int rc = (t & 31) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = (t >> 5) + 8 * i;
      r[i] = (row0 + rc < nrows) ? ld4(src + (long long)(k0 + k) * ld + row0 + rc) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void store(float* s) const {
    const int t = threadIdx.x;
    const int rc = (t & 31) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = (t >> 5) + 8 * i;
      st4(s + k * LD + rc, r[i]);
    }
  }
  // sum over this thread's 4 k of its 4 rows (columns of the [K][rows] operand)
  __device__ __forceinline__ void add_tile_sum(float4& a) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) { a.x += r[i].x; a.y += r[i].y; a.z += r[i].z; a.w += r[i].w; }
  }
};

// the four 32x32 accumulators of a wave's 64 x 64 block (128 x 128 tile, 2 x 2 waves); lds = the block's LDS (idle after the K
// loop's last barrier): per-wave transposition scratch of the epilogue (gemm.h)
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, const f32x16& acc00, const f32x16& acc01,
                                              const f32x16& acc10, const f32x16& acc11, int m0, int n0, int wm, int wn,
                                              float* lds, int z) {
  const unsigned long long seed = (p.seed && p.drop.thresh) ? *p.seed : 0ull;
  const int row0 = m0 + wm * 64, col0 = n0 + wn * 64, lane = threadIdx.x & 63;
  float* scr = lds + (threadIdx.x >> 6) * EPI_FLOATS;
  float cmax = 0.f;
  float4 rowsc = epilogue_row_scales(p, seed, row0, lane);
  epilogue_tile(p, acc00, row0, col0, lane, scr, z, seed, cmax, rowsc, p.alpha);
  epilogue_tile(p, acc01, row0, col0 + 32, lane, scr, z, seed, cmax, rowsc, p.alpha);
  rowsc = epilogue_row_scales(p, seed, row0 + 32, lane);
  epilogue_tile(p, acc10, row0 + 32, col0, lane, scr, z, seed, cmax, rowsc, p.alpha);
  epilogue_tile(p, acc11, row0 + 32, col0 + 32, lane, scr, z, seed, cmax, rowsc, p.alpha);
  if (p.splits == 1) amax_slot_commit(p.c_amax, cmax, 0u);        // (small shapes only: no peek)
}

// 128 x 128 tiles: unsplit launches use the XCD-aware remap of gemm.h; split-K launches (weight gradients) keep all tiles
// of one K-chunk on one XCD.
__device__ __forceinline__ void tile_of_block(const GemmParams& p, int& m0, int& n0, int& z) {
  if (p.splits > 1) {
    // split-K (weight gradients): all tiles of one K-chunk z read the same rows of dy and x.  Dealt round-robin, the tiles
    // of a chunk land on all 8 XCDs and every L2 fetches the whole chunk (rocprofv3 FETCH_SIZE 1.9x the operand bytes,
    // ~2.9 TB/s of fabric traffic that the HBM-bound kernels on the other stream compete with).  With splits % 8 == 0
    // chunk z is owned by XCD z % 8: its tiles march through K together and share each row block in ONE L2.
    const int tiles = gridDim.x;
    if ((p.splits & 7) == 0) {
      const int lin = blockIdx.x + tiles * blockIdx.y, xcd = lin & 7, slot = lin >> 3;
      const int g = slot / tiles, tl = slot - g * tiles;
      z = g * 8 + xcd;
      const int tile_m = tl / p.tiles_n;
      m0 = tile_m * BM; n0 = (tl - tile_m * p.tiles_n) * BN;
    } else {
      z = blockIdx.y;
      const int tile_m = blockIdx.x / p.tiles_n;
      m0 = tile_m * BM; n0 = (blockIdx.x - tile_m * p.tiles_n) * BN;
    }
    return;
  }
  z = 0;
  int tile_m, tile_n;
  tile_of_block_unsplit(p, tile_m, tile_n);
  m0 = tile_m * BM; n0 = tile_n * BN;
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_f32_kernel(GemmParams p) {
  constexpr int LDA = Stager<AKC>::LD, LDB = Stager<BKC>::LD;
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];

  int m0, n0;
  int z;
  tile_of_block(p, m0, n0, z);

  // split-K: this block reduces k in [z*K, (z+1)*K)
  const float* A = p.A + (AKC ? (long long)z * p.K : (long long)z * p.K * p.lda);
  const float* B = p.B + (BKC ? (long long)z * p.K : (long long)z * p.K * p.ldb);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;

  f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};

  Stager<AKC> sa; Stager<BKC> sb;
  const int nk = p.K / BK;
  const bool want_cs = !AKC && p.colsum && n0 == 0;
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  sa.load(A, p.lda, m0, p.M, 0);
  sb.load(B, p.ldb, n0, p.N, 0);
  if constexpr (!AKC) { if (want_cs) sa.add_tile_sum(cs); }
  sa.store(As[0]); sb.store(Bs[0]);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      sa.load(A, p.lda, m0, p.M, (kt + 1) * BK);
      sb.load(B, p.ldb, n0, p.N, (kt + 1) * BK);
    }
    const float* as = As[cur] + h * LDA + wm * 64 + r;
    const float* bs = Bs[cur] + h * LDB + wn * 64 + r;
    // operands of k-pair kk+2 are read from LDS before the four MFMAs of k-pair kk issue,
    // so the ~64-cycle ds_read latency hides under 256 cycles of matrix work
    float a0 = as[0], a1 = as[32], b0 = bs[0], b1 = bs[32];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
      if (kk + 2 < BK) {
        na0 = as[(kk + 2) * LDA]; na1 = as[(kk + 2) * LDA + 32];
        nb0 = bs[(kk + 2) * LDB]; nb1 = bs[(kk + 2) * LDB + 32];
      }
      acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
    if (kt + 1 < nk) {
      if constexpr (!AKC) { if (want_cs) sa.add_tile_sum(cs); }
      sa.store(As[cur ^ 1]); sb.store(Bs[cur ^ 1]);
    }
    __syncthreads();
  }
  if constexpr (!AKC) {
    if (want_cs) {            // 8 threads (t>>5) hold partial sums of the same 4 columns: fixed-order reduction through LDS
      float* red = As[0];
      st4(red + (threadIdx.x >> 5) * 128 + (threadIdx.x & 31) * 4, cs);
      __syncthreads();
      if (threadIdx.x < 128 && m0 + (int)threadIdx.x < p.M) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i * 128 + threadIdx.x];
        store_colsum(p, (long long)z * p.M + m0 + threadIdx.x, t);
      }
      __syncthreads();                   // `red` is about to become the epilogue's scratch
    }
  }
  static_assert(sizeof(As) >= 4 * EPI_FLOATS * 4, "scratch");
  gemm_epilogue(p, acc00, acc01, acc10, acc11, m0, n0, wm, wn, &As[0][0], z);
}

// =====================================================================================================
// Split-precision path.  Every fp32 operand x is split on the fly into NS bf16 terms
//     x = p0 + p1 (+ p2) + eps,   p0 = rne(x), p1 = rne(x - p0), p2 = rne(x - p0 - p1)
// and the product is accumulated in fp32 on v_mfma_f32_32x32x16_bf16 from the cross terms of weight >= 2^-8(NS-1):
//     NS = 2 ("bf16x3"): p0q0 + p0q1 + p1q0                      3 MFMAs / product, ~2^-16 relative error
//     NS = 3 ("bf16x6"): p0q0 + p0q1 + p1q0 + p0q2 + p2q0 + p1q1 6 MFMAs / product, ~2^-23: fp32-grade
// i.e. 16/3 = 5.3x resp. 16/6 = 2.7x the fp32-input MFMA rate, with bf16's full fp32 exponent range (no scaling
// pass, gradients of 1e-8 are safe - an fp16 split would need one).  LDS image per operand and per term:
// [k/8][row][8 k] bf16 - a row's 8 consecutive k are one 16-byte slot, rows contiguous - so the MFMA fragment of lane
// (row r, half h) is ONE conflict-free ds_read_b128.  (Earlier organisations of the same arithmetic - single LDS stage,
// producer/consumer waves, a two-term fp16 split - were measured and dropped; they live in the history of this file.)
constexpr int KG_STRIDE = 129 * 16;          // bytes between k-groups

#define NPVP_MFMA4(A0, A1, B0, B1)                                                  \
  acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc00, 0, 0, 0);          \
  acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B1, acc01, 0, 0, 0);          \
  acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B0, acc10, 0, 0, 0);          \
  acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, acc11, 0, 0, 0);


// -----------------------------------------------------------------------------------------------------
// Double-buffered, software-pipelined form of the split-precision GEMM ("db").  A single-stage kernel
// (every wave alternating stage / barrier / MFMA) left the matrix pipe at 42-51 % even with loads and staging
// removed: every wave serialised [stage -> barrier -> fragment reads -> 48 MFMAs -> barrier].  Here the K-step
// is 16 deep with TWO LDS stages (2 x 24.2 KB -> still 3 workgroups per CU): within one K-step a wave issues
// its fragment reads, then its 24 MFMAs with the split + ds_write of the NEXT tile (already in registers)
// placed in the MFMA shadows, then one barrier.  Global loads run two tiles ahead in a second register set.
constexpr int OPER16 = 2 * KG_STRIDE;        // one operand tile (128 rows x 16 k), one split term

template <int NS, bool KC> struct SplitStager16;

template <int NS> struct SplitStager16<NS, true> {    // [rows][K]: 16 lanes = 8 rows x 2 four-k halves of one k-group
  float4 r[2];
  __device__ __forceinline__ void load(const float* src, long long ld, int row0, int nrows, int k0, int t) {
    // rows past the edge are CLAMPED, not zero-filled: they only feed output rows/columns that are never stored,
    // and an unconditional load keeps the K-step one basic block (the scheduler can then interleave it with MFMAs)
    const int kof = ((t >> 4) & 1) * 8 + (t & 1) * 4, rl = (t >> 5) * 8 + ((t >> 1) & 7);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = min(row0 + rl + 64 * i, nrows - 1);
      r[i] = ld4(src + (long long)row * ld + k0 + kof);
    }
  }
  template <int I> __device__ __forceinline__ void store_part(char* base, int t) const {
    const int off = ((t >> 4) & 1) * KG_STRIDE + (t & 1) * 8 + ((t >> 5) * 8 + ((t >> 1) & 7) + 64 * I) * 16;
    float4 v = r[I];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      bf16x4 q;
      q[0] = (__bf16)v.x; q[1] = (__bf16)v.y; q[2] = (__bf16)v.z; q[3] = (__bf16)v.w;
      *reinterpret_cast<bf16x4*>(base + s * OPER16 + off) = q;
      v.x -= (float)q[0]; v.y -= (float)q[1]; v.z -= (float)q[2]; v.w -= (float)q[3];
    }
  }
  __device__ __forceinline__ float tile_sum() const { return 0.f; }
};

template <int NS> struct SplitStager16<NS, false> {   // [K][rows]: one row per lane, thread t>>7 takes k-group 0/1
  float r2[2][4];
  __device__ __forceinline__ void load(const float* src, long long ld, int row0, int nrows, int k0, int t) {
    const int row = min(row0 + (t & 127), nrows - 1), kg = t >> 7;        // clamped, see above
    const float* q = src + (long long)(k0 + kg * 8) * ld + row;
#pragma unroll
    for (int j = 0; j < 8; ++j) r2[j >> 2][j & 3] = q[(long long)j * ld];
  }
  template <int I> __device__ __forceinline__ void store_part(char* base, int t) const {
    // half I of this lane's 8 k (4 bf16 = 8 bytes per term); two halves make the 16-byte slot
    const int off = (t >> 7) * KG_STRIDE + (t & 127) * 16 + I * 8;
    float a0 = r2[I][0], a1 = r2[I][1], a2 = r2[I][2], a3 = r2[I][3];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      bf16x4 q;
      q[0] = (__bf16)a0; q[1] = (__bf16)a1; q[2] = (__bf16)a2; q[3] = (__bf16)a3;
      *reinterpret_cast<bf16x4*>(base + s * OPER16 + off) = q;
      a0 -= (float)q[0]; a1 -= (float)q[1]; a2 -= (float)q[2]; a3 -= (float)q[3];
    }
  }
  __device__ __forceinline__ float tile_sum() const {
    return r2[0][0] + r2[0][1] + r2[0][2] + r2[0][3] + r2[1][0] + r2[1][1] + r2[1][2] + r2[1][3];
  }
};

// B operand from PRE-SPLIT planes: weights change once per optimiser step but are staged by every tile of three
// GEMMs, so they are split once (split_weight_kernel) into NS bf16 planes laid out exactly like the LDS image,
// [term][K/8][N][8 k]: a tile's (term, k-group) slab is 128 rows x 16 B = 2 KB contiguous.  Staging B is then three
// coalesced 16-byte loads + three ds_write_b128 per thread and K-step - no conversion VALU at all.  (The K-step of
// the db kernel is issue bound: 72 MFMA + ~360 VALU + ~90 LDS/VMEM instructions per SIMD against 2304 MFMA cycles.)
template <int NS> struct PreStager16 {
  uint4 r[NS];
  __device__ __forceinline__ void load(const GemmParams& p, int n0, int k0, int t) {
    const int row = min(n0 + (t & 127), p.N - 1), kg = t >> 7;
    const uint4* base = reinterpret_cast<const uint4*>(p.b_pre) + (long long)(k0 / 8 + kg) * p.N + row;
#pragma unroll
    for (int s = 0; s < NS; ++s) r[s] = base[(long long)s * (p.b_pre_plane / 8)];
  }
  template <int I> __device__ __forceinline__ void store_part(char* base, int t) const {
    const int off = (t >> 7) * KG_STRIDE + (t & 127) * 16;
    if (I == 0) {
      *reinterpret_cast<uint4*>(base + off) = r[0];
      if (NS > 2) *reinterpret_cast<uint4*>(base + 2 * OPER16 + off) = r[NS - 1];
    } else {
      *reinterpret_cast<uint4*>(base + OPER16 + off) = r[1];
    }
  }
};

// w [N][K] (row stride ld) -> fwd planes F[term][K/8][N][8 over k]  and  dgrad planes D[term][N/8][K][8 over n]
__global__ void split_weight_fwd_kernel(const float* __restrict__ w, long long ld, int N, int K, __bf16* __restrict__ F) {
  const long long total = (long long)N * (K / 8);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i % N), kb = (int)(i / N);
    const float4 a = ld4(w + (long long)n * ld + kb * 8), b = ld4(w + (long long)n * ld + kb * 8 + 4);
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      bf16x8 q;
#pragma unroll
      for (int j = 0; j < 8; ++j) { q[j] = (__bf16)v[j]; v[j] -= (float)q[j]; }
      *reinterpret_cast<bf16x8*>(F + ((long long)s * N * K) + i * 8) = q;
    }
  }
}
__global__ void split_weight_dgrad_kernel(const float* __restrict__ w, long long ld, int N, int K, __bf16* __restrict__ D) {
  const long long total = (long long)(N / 8) * K;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % K), nb = (int)(i / K);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = w[(long long)(nb * 8 + j) * ld + k];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      bf16x8 q;
#pragma unroll
      for (int j = 0; j < 8; ++j) { q[j] = (__bf16)v[j]; v[j] -= (float)q[j]; }
      *reinterpret_cast<bf16x8*>(D + ((long long)s * N * K) + i * 8) = q;
    }
  }
}

template <int NS, bool AKC, bool BKC, bool BPRE, bool ROWSTATS = false>
__global__ __launch_bounds__(GEMM_THREADS, 3) void gemm_split_db_kernel(GemmParams p) {
  constexpr int STAGE = 2 * NS * OPER16;                      // [A term 0..NS-1 | B term 0..NS-1]
  constexpr int BK16 = 16;
  __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];
  int m0, n0;
  int z;
  tile_of_block(p, m0, n0, z);
  const float* A = p.A + (AKC ? (long long)z * p.K : (long long)z * p.K * p.lda);
  const float* B = p.B + (BKC ? (long long)z * p.K : (long long)z * p.K * p.ldb);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
  const int nk = p.K / BK16;
  const bool want_cs = !AKC && p.colsum && n0 == 0;
  float cs = 0.f;
  const int fa_off = h * KG_STRIDE + (wm * 64 + r) * 16;
  const int fb_off = NS * OPER16 + h * KG_STRIDE + (wn * 64 + r) * 16;

  SplitStager16<NS, AKC> a0s, a1s;       // two register sets: tiles kt+1 and kt+2
  typename std::conditional<BPRE, PreStager16<NS>, SplitStager16<NS, BKC>>::type b0s, b1s;
#define NPVP_BLOAD(ST, K0) { if constexpr (BPRE) ST.load(p, n0, (K0) + z * p.K, t); else ST.load(B, p.ldb, n0, p.N, (K0), t); }
  // prologue: tile 0 -> stage 0, tile 1 -> set 0
  a0s.load(A, p.lda, m0, p.M, 0, t); NPVP_BLOAD(b0s, 0)
  if constexpr (!AKC) { if (want_cs) cs += a0s.tile_sum(); }
  a0s.template store_part<0>(lds, t); a0s.template store_part<1>(lds, t);
  b0s.template store_part<0>(lds + NS * OPER16, t); b0s.template store_part<1>(lds + NS * OPER16, t);
  { const int k1 = min(BK16, p.K - BK16); a0s.load(A, p.lda, m0, p.M, k1, t); if constexpr (!BPRE) NPVP_BLOAD(b0s, k1) }
  __syncthreads();

#define NPVP_DB_STEP(KT, SA_CUR, SB_CUR, SA_NXT, SB_NXT)                                                        \
  {                                                                                                             \
    const char* st = lds + ((KT) & 1) * STAGE;                                                                  \
    char* nx = lds + (((KT) + 1) & 1) * STAGE;                                                                  \
    const int k2 = min(((KT) + 2) * BK16, p.K - BK16);     /* past the end: re-read the last tile, never consumed */ \
    SA_NXT.load(A, p.lda, m0, p.M, k2, t);                                                                      \
    /* pre-split B needs no conversion: ONE register set, tile kt+1 loaded at the top of step kt, stored at its end */ \
    if constexpr (BPRE) { NPVP_BLOAD(SB_CUR, min(((KT) + 1) * BK16, p.K - BK16)) } else { NPVP_BLOAD(SB_NXT, k2) }  \
    bf16x8 fa0[NS], fa1[NS], fb0[NS], fb1[NS];                                                                  \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                                            \
      fa0[s] = *reinterpret_cast<const bf16x8*>(st + fa_off + s * OPER16);                                      \
      fa1[s] = *reinterpret_cast<const bf16x8*>(st + fa_off + s * OPER16 + 32 * 16);                            \
      fb0[s] = *reinterpret_cast<const bf16x8*>(st + fb_off + s * OPER16);                                      \
      fb1[s] = *reinterpret_cast<const bf16x8*>(st + fb_off + s * OPER16 + 32 * 16);                            \
    }                                                                                                           \
    /* the stores below are unconditional: after the last tile they fill the stage nobody reads again */        \
    if constexpr (!AKC) { if (want_cs && (KT) + 1 < nk) cs += SA_CUR.tile_sum(); }                              \
    SA_CUR.template store_part<0>(nx, t);                                                                       \
    if (NS == 3) { NPVP_MFMA4(fa0[1], fa1[1], fb0[1], fb1[1]) }                                                 \
    SA_CUR.template store_part<1>(nx, t);                                                                       \
    if (NS == 3) { NPVP_MFMA4(fa0[0], fa1[0], fb0[2], fb1[2]) }                                                 \
    if constexpr (!BPRE) SB_CUR.template store_part<0>(nx + NS * OPER16, t);                                    \
    if (NS == 3) { NPVP_MFMA4(fa0[2], fa1[2], fb0[0], fb1[0]) }                                                 \
    if (NS != 3) { NPVP_MFMA4(fa0[0], fa1[0], fb0[1], fb1[1]) }                                                 \
    if constexpr (!BPRE) SB_CUR.template store_part<1>(nx + NS * OPER16, t);                                    \
    if (NS == 3) { NPVP_MFMA4(fa0[0], fa1[0], fb0[1], fb1[1]) }                                                 \
    NPVP_MFMA4(fa0[1], fa1[1], fb0[0], fb1[0])                                                                  \
    NPVP_MFMA4(fa0[0], fa1[0], fb0[0], fb1[0])                                                                  \
    if constexpr (BPRE) { SB_CUR.template store_part<0>(nx + NS * OPER16, t); SB_CUR.template store_part<1>(nx + NS * OPER16, t); } \
    __syncthreads();                                                                                            \
  }

  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    NPVP_DB_STEP(kt, a0s, b0s, a1s, b1s)
    NPVP_DB_STEP(kt + 1, a1s, b1s, a0s, b0s)
  }
  if (kt < nk) NPVP_DB_STEP(kt, a0s, b0s, a1s, b1s)
#undef NPVP_DB_STEP
#undef NPVP_BLOAD

  if constexpr (!AKC) {
    if (want_cs) {            // threads t and t+128 hold the two k-group halves of column t&127
      float* red = reinterpret_cast<float*>(lds);
      red[t] = cs;
      __syncthreads();
      if (t < 128 && m0 + t < p.M) store_colsum(p, (long long)z * p.M + m0 + t, red[t] + red[t + 128]);
      __syncthreads();                   // `red` is about to become the epilogue's scratch
    }
  }
  static_assert(2 * STAGE >= 4 * EPI_FLOATS * 4, "scratch");
  float* scr0 = reinterpret_cast<float*>(lds);
  if constexpr (ROWSTATS) {
    float cmax = 0.f;
    epilogue_rowstats_block(p, acc00, acc01, acc10, acc11, m0 + wm * 64, n0 + wn * 64, lane, scr0 + wave * EPI_FLOATS, cmax, p.alpha);
    amax_slot_commit(p.c_amax, cmax, 0u);
  } else gemm_epilogue(p, acc00, acc01, acc10, acc11, m0, n0, wm, wn, scr0, z);
}

// All registered weight views in ONE launch (after the optimiser step): desc[v] = {w, ld, N, K, F, D} as 64-bit words
// (npvp_split_weights_batched); blockIdx.y = view, blockIdx.x strides over its N*K/8 output slots of F, then of D.
struct SplitDesc { const float* w; long long ld; long long N; long long K; __bf16* F; __bf16* D; };
__global__ void split_weights_batched_kernel(const SplitDesc* __restrict__ desc) {
  const SplitDesc d = desc[blockIdx.y];
  const int N = (int)d.N, K = (int)d.K;
  const long long slots = (long long)N * K / 8, plane = (long long)N * K;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * slots; i += (long long)gridDim.x * blockDim.x) {
    float v[8];
    __bf16* out;
    if (i < slots) {                          // F[term][k/8][n][8 over k]
      const int n = (int)(i % N), kb = (int)(i / N);
      const float4 a = ld4(d.w + (long long)n * d.ld + kb * 8), b = ld4(d.w + (long long)n * d.ld + kb * 8 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      out = d.F + i * 8;
    } else {                                  // D[term][n/8][k][8 over n]
      const long long j = i - slots;
      const int k = (int)(j % K), nb = (int)(j / K);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = d.w[(long long)(nb * 8 + e) * d.ld + k];
      out = d.D + j * 8;
    }
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) {
      bf16x8 q;
#pragma unroll
      for (int e = 0; e < 8; ++e) { q[e] = (__bf16)v[e]; v[e] -= (float)q[e]; }
      *reinterpret_cast<bf16x8*>(out + (long long)s_ * plane) = q;
    }
  }
}

// sum split-K partial slabs: out[m][n] = alpha * sum_z ws[z][m][n]   (ldc-strided out); the same launch also finishes the
// column-sum partials of the bias gradient (cs_part [splits][M] -> cs_out [M], no alpha), which used to be a launch of its own
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int M, int N,
                                     long long ldc, int splits, float alpha, int accum, const float* __restrict__ cs_part,
                                     float* __restrict__ cs_out, float* c_amax) {
  // c_amax (nullable): the bound of the values stored to `out` - the split launch itself cannot commit it (its tiles are partial
  // sums), and a consumer that trusted an unraised slot would run at scale 1
  __shared__ float ared[4];
  const unsigned int cpeek = amax_peek_block(c_amax);
  float cmax = 0.f;
  const ReduceJob j = {ws, out, ldc, M, N, splits, accum, alpha, 0, cs_part, cs_out};
  splitk_reduce_body(j, blockIdx.x, gridDim.x, cmax);
  amax_slot_commit_block(c_amax, cmax, ared, cpeek);
}

static int pick_splits(int M, int N, int K) {
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  if (tiles >= 256 || K < 2048) return 1;
  // ~2 blocks per CU.  Weight gradients run on the low-priority gradient stream and fill gaps, so occupancy of their own
  // grid matters less than the partial-slab traffic (splits x M x N x 8 bytes written and re-read) that competes with
  // the HBM-bound kernels of the critical path: 512 blocks instead of 1024 measured -2.6 ms on a c1 step, 256: -0.9 ms.
  int s = (512 + tiles - 1) / tiles;
  const int maxs = K / (BK * 8);               // at least 8 K-steps per split
  if (s > maxs) s = maxs;
  if (s > 64) s = 64;
  while (s > 1 && (K % (s * BK)) != 0) --s;
  return s < 1 ? 1 : s;
}

}  // namespace npvp

using namespace npvp;

// split-K partial slabs [splits][M][N] followed by the column-sum partials [splits][M]: the larger of what the 128 x 128
// kernel and the wide weight-gradient kernel would use for this shape
extern "C" long long npvp_gemm_workspace_bytes(int M, int N, int K) {
  const int s = pick_splits(M, N, K), sw = wide_wgrad_splits(M, N, K), sh = f16_wgrad_splits(M, N, K);
  const int m = (s > sw ? s : sw) > sh ? (s > sw ? s : sw) : sh;
  return m > 1 ? ((long long)m * M * N + (long long)m * M) * 4 : 0;
}

// w [N][K] fp32 -> bf16 planes (3 terms each): F for y = x w^T (B operand [N][K]), D for dx = dy w (B operand [K][N]).
// Each output holds 3*N*K bf16.  Either may be null.
extern "C" int npvp_split_weight(const float* w, long long ld, int N, int K, void* F, void* D, hipStream_t stream) {
  NPVP_CHECK_ARG(N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0 && ld % 4 == 0, "split_weight: N, K must be multiples of 8");
  NPVP_CHECK_ARG(((uintptr_t)w % 16) == 0, "split_weight: w must be 16-byte aligned");
  const long long total = (long long)N * K / 8;
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  if (F) { NPVP_LAUNCH(split_weight_fwd_kernel, dim3(blocks), dim3(256), 0, stream, w, ld, N, K, (__bf16*)F); NPVP_CHECK_LAUNCH(); }
  if (D) { NPVP_LAUNCH(split_weight_dgrad_kernel, dim3(blocks), dim3(256), 0, stream, w, ld, N, K, (__bf16*)D); NPVP_CHECK_LAUNCH(); }
  return NPVP_OK;
}

// desc: DEVICE array of `count` records of six 64-bit words {w, ld, N, K, F, D} (pointers as integers); every view obeys the
// npvp_split_weight contract.  One launch re-splits every weight of the model after an optimiser step.
extern "C" int npvp_split_weights_batched(const void* desc, int count, hipStream_t stream) {
  NPVP_CHECK_ARG(desc && count > 0, "split_weights_batched: empty table");
  NPVP_LAUNCH(split_weights_batched_kernel, dim3(128, count), dim3(256), 0, stream, (const SplitDesc*)desc);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

// Which kernel npvp_gemm_f32 runs for a problem (a pure function of the shape; bench.py groups its live event-pair timings
// by it so that they line up with the per-kernel rows of a rocprofv3 trace): 0 gemm_f32_kernel, 1 gemm_split_db_kernel,
// 2 gemm_wide_kernel (128 x 256 tiles), 3 gemm_wgrad_wide_kernel, 4 gemm_wide_kernel's 128 x 128 instantiation.  has_planes = b_pre
// will be passed.
extern "C" int npvp_gemm_kernel_id(int a_kc, int b_kc, int M, int N, int K, int precision, int has_planes) {
  if (precision == 0) return 0;
  if (precision == 6) {
    if (a_kc && has_planes && gemm_f16_variant(M, N, K)) return gemm_f16_variant(M, N, K) == 1 ? 5 : 7;
    if (!a_kc && !b_kc && f16_wgrad_splits(M, N, K) > 0) return 6;
    return 1;
  }
  if (precision == 4 && a_kc && has_planes && pick_splits(M, N, K) == 1 && gemm_wide_takes(M, N, K))
    return gemm_wide_variant(M, N, K) == 1 ? 2 : 4;
  if (precision == 4 && !a_kc && !b_kc && wide_wgrad_splits(M, N, K) > 0) return 3;
  return 1;
}

// See include/npvp_hip.h for the contract.
extern "C" int npvp_gemm_f32(int a_kc, int b_kc, int M, int N, int K, const float* A, long long lda, const float* B,
                             long long ldb, float* C, long long ldc, const float* bias, int act, const float* aux_in,
                             float* aux_out, const float* residual, long long ldr, float drop_p, int drop_mode,
                             int drop_g1, int drop_g2, const unsigned long long* seed, unsigned int salt, float alpha,
                             int precision, float* colsum_a, const void* b_pre, int accumulate, float* rowstats,
                             const float* a_amax, const float* b_amax, float* c_amax, unsigned int* range_flag,
                             float adrop_p, int adrop_g1, int adrop_g2, unsigned int adrop_salt,
                             void* workspace, long long ws_bytes, hipStream_t stream) {
  NPVP_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: empty problem");
  NPVP_CHECK_ARG(precision == 0 || precision == 4 || precision == 5 || precision == 6,
                 "gemm: precision must be 0 (exact fp32-input MFMA), 4 (three-term bf16 split, 6 MFMAs per product: "
                 "fp32-grade), 5 (two-term bf16 split, 3 MFMAs per product, ~2^-16) or 6 (two-term fp16 split with amax-scaled "
                 "operands, 3 MFMAs per product: fp32-grade; shapes it does not take run as 4)");
  // precision 6 = fp16 two-term kernels where they apply (a row-major A with fp16 planes of B + both amax slots; weight
  // gradients with both amax slots), the three-term bf16 kernels WITHOUT planes everywhere else
  const bool want_h = precision == 6;
  if (want_h) precision = 4;
  NPVP_CHECK_ARG(K % BK == 0, "gemm: K must be a multiple of 32");
  NPVP_CHECK_ARG(M % 4 == 0 && N % 4 == 0, "gemm: M and N must be multiples of 4");
  NPVP_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0, "gemm: pointers must be 16-byte aligned");
  NPVP_CHECK_ARG(lda % 4 == 0 && ldb % 4 == 0, "gemm: lda/ldb must be multiples of 4 floats");
  NPVP_CHECK_ARG(ldc % 4 == 0 && (!residual || (ldr % 4 == 0 && ((uintptr_t)residual % 16) == 0)) && (!bias || ((uintptr_t)bias % 16) == 0) &&
                 (!aux_in || ((uintptr_t)aux_in % 16) == 0) && (!aux_out || ((uintptr_t)aux_out % 16) == 0),
                 "gemm: ldc / ldr must be multiples of 4 floats and bias / aux / residual 16-byte aligned (float4 epilogue)");
  NPVP_CHECK_ARG(!(a_kc == 0 && b_kc == 1), "gemm: (a_kc=0,b_kc=1) is not used by the path");
  NPVP_CHECK_ARG((act != 3 && act != 4) || aux_in, "gemm: act 3/4 need aux_in");
  NPVP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "gemm: dropout p out of range");
  NPVP_CHECK_ARG(drop_p == 0.f || seed, "gemm: dropout needs a device seed");
  NPVP_CHECK_ARG(!colsum_a || a_kc == 0, "gemm: colsum_a is the column sum of a [K][M] operand (a_kc = 0)");
  NPVP_CHECK_ARG(!b_pre || ((uintptr_t)b_pre % 16) == 0, "gemm: b_pre must be 16-byte aligned");

  GemmParams p;
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.residual = residual; p.aux_out = aux_out; p.aux_in = aux_in;
  p.seed = seed; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.M = M; p.N = N; p.K = K; p.act = act;
  p.drop = make_drop_spec(drop_p, salt, drop_mode, drop_g1, drop_g2);
  // row-group mask on operand A (see GemmParams::adrop): only the fp16 kernels implement it, and only with whole K-steps per group
  NPVP_CHECK_ARG(adrop_p >= 0.f && adrop_p < 0.5f, "gemm: adrop_p must be in [0, 0.5) (the masked operand must stay inside its amax scale)");
  NPVP_CHECK_ARG(adrop_p == 0.f || (seed && want_h && adrop_g1 > 0 && adrop_g2 > 0 && (a_kc || adrop_g1 % 16 == 0)),
                 "gemm: adrop needs precision 6, a device seed and (for a weight gradient) groups of a multiple of 16 rows");
  p.adrop = make_drop_spec(adrop_p, adrop_salt, 1, adrop_g1, adrop_g2);
  p.alpha = alpha;
  p.tiles_m = (M + BM - 1) / BM; p.tiles_n = (N + BN - 1) / BN;
  int splits = pick_splits(M, N, K);
  const bool plain = !bias && act == 0 && !aux_out && !residual && drop_p == 0.f;
  if (splits > 1 && (!plain || ws_bytes < npvp_gemm_workspace_bytes(M, N, K) || !workspace || (N % 4) != 0)) splits = 1;
  p.splits = splits;
  p.colsum = colsum_a;
  p.rowstats = rowstats;
  p.a_amax = a_amax; p.b_amax = b_amax; p.c_amax = c_amax; p.range_flag = range_flag;
  p.prev = ReduceJob{};
  NPVP_CHECK_ARG(!rowstats || (precision == 4 && a_kc && b_kc && M % 64 == 0 && N % 128 == 0 && act == 0 && !aux_out && !residual &&
                               drop_p == 0.f && !accumulate),
                 "gemm: rowstats needs the default (bf16x6) forward layout, M % 64 == 0, N % 128 == 0 and a bias-only epilogue");
  p.accum = accumulate ? 1 : 0;
  // pre-split B planes are only consumed by the bf16x6 kernels with a row-major A and an unsplit reduction
  const void* b_pre_arg = b_pre;
  p.b_pre = (precision == 4 && a_kc && splits == 1 && K % 16 == 0 && N % 8 == 0) ? b_pre : nullptr;
  p.b_pre_plane = (long long)N * K;
  b_pre = p.b_pre;

  if (want_h) {
    if (b_pre_arg && a_kc && K % 16 == 0 && N % 8 == 0 && a_amax && b_amax) {
      p.b_pre = b_pre_arg; p.splits = 1;
      if (launch_gemm_f16(p, stream)) {
        NPVP_CHECK_LAUNCH();
        return NPVP_OK;
      }
      p.splits = splits;
    }
    p.b_pre = b_pre = nullptr;               // fp16 planes are not what the bf16 kernels read
    if (!a_kc && !b_kc && a_amax && b_amax && !bias && act == 0 && !aux_out && !residual && drop_p == 0.f && !rowstats &&
        N % 4 == 0 && M % 4 == 0) {
      const int sh = f16_wgrad_splits(M, N, K);
      if (sh == 1 || (sh > 1 && workspace && ws_bytes >= ((long long)sh * M * N + (long long)sh * M) * 4)) {
        splits = sh; p.splits = sh;
        p.colgroups = 1;
        if (sh > 1) {
          p.K = K / sh; p.C = (float*)workspace; p.ldc = N;
          if (colsum_a) p.colsum = (float*)workspace + (long long)sh * M * N;
        }
        launch_gemm_wgrad_f16(p, sh, stream);
        NPVP_CHECK_LAUNCH();
        if (sh > 1) {
          const long long total4 = (long long)M * N / 4;
          int blocks = (int)((total4 + 255) / 256); if (blocks > 2048) blocks = 2048;
          const bool cs_here = colsum_a && M % 4 == 0 && ((uintptr_t)colsum_a % 16) == 0;       // (a float4-aligned bias gradient rides along)
          NPVP_LAUNCH(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)workspace, C, M, N, ldc,
                             sh, alpha, p.accum, cs_here ? (const float*)p.colsum : nullptr, cs_here ? colsum_a : nullptr, c_amax);
          NPVP_CHECK_LAUNCH();
          if (colsum_a && !cs_here && launch_sum_rows(p.colsum, colsum_a, sh, M, M, stream, p.accum)) {
            npvp_set_error("gemm: column-sum reduce launch failed");
            return NPVP_ERR_LAUNCH;
          }
        }
        return NPVP_OK;
      }
    }
  }

  NPVP_CHECK_ARG(adrop_p == 0.f, "gemm: adrop was requested but this shape does not run on the fp16 kernels (npvp_gemm_kernel_id tells)");

  // large forward / dgrad shapes: 256 x 256 tiles (gemm_wide.hip); it declines what it is not built for
  if (b_pre && launch_gemm_wide(p, stream)) {
    NPVP_CHECK_LAUNCH();
    return NPVP_OK;
  }

  // weight gradients (dW = dy^T x): 128 x 256 tiles, row-major staging + transposing LDS reads (gemm_wide.hip)
  bool wide_wgrad = false;
  if (precision == 4 && !a_kc && !b_kc && plain && !rowstats && N % 4 == 0 && M % 4 == 0) {
    const int sw = wide_wgrad_splits(M, N, K);
    if (sw == 1 || (sw > 1 && workspace && ws_bytes >= ((long long)sw * M * N + (long long)sw * M) * 4)) {
      wide_wgrad = true;
      splits = sw; p.splits = sw;
    }
  }

  p.colgroups = splits == 1 ? pick_colgroups((long long)N * K * 4, p.tiles_m, p.tiles_n) : 1;
  if (splits > 1) {
    p.K = K / splits; p.C = (float*)workspace; p.ldc = N;
    if (colsum_a) p.colsum = (float*)workspace + (long long)splits * M * N;
  }

  dim3 grid(p.tiles_m * p.tiles_n, splits), block(GEMM_THREADS);
  if (wide_wgrad) {
    launch_gemm_wgrad_wide(p, splits, stream);
  } else if (precision == 0) {
    if (a_kc && b_kc) NPVP_LAUNCH((gemm_f32_kernel<true, true>), grid, block, 0, stream, p);
    else if (a_kc && !b_kc) NPVP_LAUNCH((gemm_f32_kernel<true, false>), grid, block, 0, stream, p);
    else NPVP_LAUNCH((gemm_f32_kernel<false, false>), grid, block, 0, stream, p);
  } else if (precision == 4) {
    if (rowstats) NPVP_LAUNCH((gemm_split_db_kernel<3, true, true, false, true>), grid, block, 0, stream, p);
    else if (b_pre && a_kc) NPVP_LAUNCH((gemm_split_db_kernel<3, true, true, true>), grid, block, 0, stream, p);
    else if (a_kc && b_kc) NPVP_LAUNCH((gemm_split_db_kernel<3, true, true, false>), grid, block, 0, stream, p);
    else if (a_kc && !b_kc) NPVP_LAUNCH((gemm_split_db_kernel<3, true, false, false>), grid, block, 0, stream, p);
    else NPVP_LAUNCH((gemm_split_db_kernel<3, false, false, false>), grid, block, 0, stream, p);
  } else {
    if (a_kc && b_kc) NPVP_LAUNCH((gemm_split_db_kernel<2, true, true, false>), grid, block, 0, stream, p);
    else if (a_kc && !b_kc) NPVP_LAUNCH((gemm_split_db_kernel<2, true, false, false>), grid, block, 0, stream, p);
    else NPVP_LAUNCH((gemm_split_db_kernel<2, false, false, false>), grid, block, 0, stream, p);
  }
  NPVP_CHECK_LAUNCH();
  if (splits > 1) {
    const long long total4 = (long long)M * N / 4;
    int blocks = (int)((total4 + 255) / 256); if (blocks > 2048) blocks = 2048;
    const bool cs_here = colsum_a && M % 4 == 0 && ((uintptr_t)colsum_a % 16) == 0;
    NPVP_LAUNCH(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)workspace, C, M, N, ldc,
                       splits, alpha, p.accum, cs_here ? (const float*)p.colsum : nullptr, cs_here ? colsum_a : nullptr, c_amax);
    NPVP_CHECK_LAUNCH();
    if (colsum_a && !cs_here && launch_sum_rows(p.colsum, colsum_a, splits, M, M, stream, p.accum)) {
      npvp_set_error("gemm: column-sum reduce launch failed");
      return NPVP_ERR_LAUNCH;
    }
  }
  return NPVP_OK;
}
