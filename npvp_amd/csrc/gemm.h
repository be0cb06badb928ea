// Shared by the GEMM translation units (gemm.hip: fp32-MFMA triage kernel + 128x128 split kernel + C entry point;
// gemm_wide.hip: 256-row-tile split kernels for the large forward / dgrad / wgrad shapes).
#pragma once
#include <type_traits>
#include "common.h"

namespace npvp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// A split-K reduction still to do: out[m][n] (+)= alpha * sum_z ws[z][m][n] (and, with it, cs_out[m] (+)= sum_z cs_part[z][m], the
// bias gradient's column-sum partials).  Plain data: the host hands it from one weight-gradient launch to the next (gemm_f16.hip).
struct ReduceJob {
  const float* ws; float* out; long long ldc; int M, N, splits, accum; float alpha; int blocks; const float* cs_part; float* cs_out;
};

struct GemmParams {
  const float* A; const float* B; float* C;
  const float* bias;       // [N] or null
  const float* residual;   // [M][ldr] or null (added last)
  float* aux_out;          // [M][ldc] pre-activation copy (after bias) or null
  const float* aux_in;     // [M][ldc] for act 3/4 (activation gradients)
  float* rowstats;         // frame-statistics partials of the output (see epilogue_rowstats_block), or null
  float* colsum;           // a_kc==0 only: colsum[z][m] = sum over this split's k of A[k][m] (bias gradient), or null
  const unsigned long long* seed;  // device seed for dropout or null
  long long lda, ldb, ldc, ldr;
  int M, N, K;             // K = this launch's reduction length per split
  int act;                 // 0 none 1 gelu 2 relu 3 *gelu'(aux_in) 4 *relu'(aux_in)
  DropSpec drop;
  DropSpec adrop;          // fp16 kernels only: a ROW-GROUP mask (mode 1) on the rows of operand A (dgrad: A = dy [M][K]; weight
                           // gradient: A = dy [K][M]) - the backward of a DropPath site without a masked copy of dy; thresh 0 = off
  int tiles_m, tiles_n;
  int splits;              // >1: raw partial tiles go to C + z*M*ldc (workspace)
  float alpha;
  const void* b_pre;       // pre-split B planes (bf16, blocked [term][K/8][N][8]) or null: see split_weight kernels
  long long b_pre_plane;   // bf16 elements per term plane (= N*K)
  int colgroups;           // XCD tiling: 1 = every XCD sweeps all tile columns; G>1 = XCD x owns column group x%G (see tile_of_block)
  int accum;               // 1: C += result and colsum += sums (gradient accumulation into a live .grad slice)
  // fp16 two-term kernels (gemm_f16.hip): amax slots (32 floats whose maximum bounds |operand|) of A and B; any kernel:
  // c_amax (nullable) receives the bound of the values this launch stores to C (the next GEMM's a_amax)
  const float* a_amax; const float* b_amax; float* c_amax;
  unsigned int* range_flag;   // nullable; fp16 weight-gradient kernel: raised when a column of A lies 2^18 below its bound (see the kernel)
  ReduceJob prev;             // fp16 weight-gradient kernel: the PREVIOUS launch's split-K reduction, done by prev.blocks extra
                              // workgroups of this launch (grid.y = splits + extra rows); prev.blocks == 0: none
};

// the reduction, by `nblocks` blocks of 256 threads of which this is block `bid` (fixed summation order: bit-identical whoever runs it)
__device__ __forceinline__ void splitk_reduce_body(const ReduceJob& j, int bid, int nblocks, float& cmax) {
  const long long total4 = (long long)j.M * j.N / 4, cs4 = j.cs_out ? j.M / 4 : 0;
  for (long long i = (long long)bid * 256 + threadIdx.x; i < total4 + cs4; i += (long long)nblocks * 256) {
    if (i >= total4) {
      const long long e = (i - total4) * 4;
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int zz = 0; zz < j.splits; ++zz) {
        const float4 v = ld4(j.cs_part + (long long)zz * j.M + e);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      if (j.accum) { const float4 o = ld4(j.cs_out + e); s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
      st4(j.cs_out + e, s);
      continue;
    }
    const long long e = i * 4;
    const int m = (int)(e / j.N), n = (int)(e - (long long)m * j.N);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int zz = 0; zz < j.splits; ++zz) {
      const float4 v = ld4(j.ws + (long long)zz * j.M * j.N + e);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    s.x *= j.alpha; s.y *= j.alpha; s.z *= j.alpha; s.w *= j.alpha;
    if (j.accum) { const float4 o = ld4(j.out + (long long)m * j.ldc + n); s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    st4(j.out + (long long)m * j.ldc + n, s);
    cmax = amax4(cmax, s);
  }
}

__device__ __forceinline__ void store_colsum(const GemmParams& p, long long idx, float v) {
  p.colsum[idx] = (p.accum && p.splits == 1) ? p.colsum[idx] + v : v;      // split-K partials are summed (and accumulated) later
}

// ---- epilogue of ONE 32x32 accumulator whose top-left element is (row0, col0).
// The C/D map of the 32x32 MFMA is col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): a lane owns one COLUMN,
// so storing from that layout is 16 dword stores per tile, each 2 x 128 B per wave (and the loads of aux_in / residual /
// the accumulate target likewise) - a 128 x 256 tile left as 512 store instructions per workgroup, and with the stores
// removed the f16 kernel ran 19 % faster.  The tile is therefore TRANSPOSED through a per-wave LDS scratch (the operand
// stages are idle after the K loop): 16 ds_write_b32 + 4 ds_read_b128 turn it into the row-major map
//     lane l -> rows (l >> 3) + 8 k (k = 0..3), columns 4 (l & 7) .. + 3
// in which every option of the epilogue works on float4 and a wave-instruction moves 8 rows x 128 B = 1 KB.
// Order: *alpha +bias -> aux_out (pre-activation copy) -> act -> dropout -> +residual -> (+= C when accumulating).
// Every option is a wave-uniform flag tested ONCE per tile.  ldc, ldr % 4 == 0 and 16-byte aligned bases (checked on the host).
constexpr int EPI_LD = 36;                         // floats per scratch row (16-byte aligned rows)
constexpr int EPI_FLOATS = 32 * EPI_LD;            // per wave

__device__ __forceinline__ void epi_transpose(const f32x16& acc, float* scr, int lane, float4 (&v)[4]) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int g = 0; g < 16; ++g) scr[(((g & 3) + 8 * (g >> 2)) + 4 * h) * EPI_LD + r] = acc[g];
  // (one wave: its LDS operations execute in order, the reads below see the writes above and the next tile's writes cannot
  // overtake these reads)
  const int rr = lane >> 3, cq = lane & 7;
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4*>(scr + (rr + 8 * k) * EPI_LD + 4 * cq);
}

__device__ __forceinline__ float4 f4_mad(const float4& a, float s, const float4& b) {
  return make_float4(a.x * s + b.x, a.y * s + b.y, a.z * s + b.z, a.w * s + b.w);
}
__device__ __forceinline__ void f4_add(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// f(integral_constant<int, I>) for I = 0 .. N-1, unrolled by construction.  The epilogues index the accumulator tiles acc[i][j]
// with these: a `#pragma unroll` loop the optimizer declines ("unrolled size is too large": three epilogue variants per tile)
// turns acc[i][j] into a dynamically indexed array - 128 accumulator registers through scratch memory, 1.6x the launch time.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// the keep-scales of a row-group (DropPath) mask for the four rows a lane owns in every 32 x 32 sub-tile of one 32-row block:
// taken once per row block, not once per sub-tile and element
__device__ __forceinline__ float4 epilogue_row_scales(const GemmParams& p, unsigned long long seed, int row0, int lane) {
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
  if (p.drop.thresh && p.drop.mode != 0) {
    const int r = row0 + (lane >> 3);
    sc.x = drop_spec_scale(p.drop, seed, r, 0, 1); sc.y = drop_spec_scale(p.drop, seed, r + 8, 0, 1);
    sc.z = drop_spec_scale(p.drop, seed, r + 16, 0, 1); sc.w = drop_spec_scale(p.drop, seed, r + 24, 0, 1);
  }
  return sc;
}

template <bool CHECK>
__device__ __forceinline__ void epilogue_rows(const GemmParams& p, float4 (&v)[4], int row0, int col0, int lane, int z,
                                              unsigned long long seed, float& cmax, float4 rowsc, float alpha) {
  const int rb = row0 + (lane >> 3), col = col0 + 4 * (lane & 7);
  if (CHECK && col >= p.N) return;                 // (N % 4 == 0: a quad is inside or outside)
#define NPVP_ROW(k) (rb + 8 * (k))
#define NPVP_INB(k) (!CHECK || NPVP_ROW(k) < p.M)
  float* cp = p.C + (p.splits > 1 ? (long long)z * p.M * p.ldc : 0ll) + col;
  const long long ld = p.ldc;
  if (p.splits > 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) if (NPVP_INB(k)) st4(cp + NPVP_ROW(k) * ld, v[k]);
    return;
  }
  const float4 bv = p.bias ? ld4(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = f4_mad(v[k], alpha, bv);
  if (p.aux_out) {
#pragma unroll
    for (int k = 0; k < 4; ++k) if (NPVP_INB(k)) st4(p.aux_out + NPVP_ROW(k) * ld + col, v[k]);
  }
  if (p.act == 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = make_float4(gelu_f(v[k].x), gelu_f(v[k].y), gelu_f(v[k].z), gelu_f(v[k].w));
  } else if (p.act == 2) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = make_float4(fmaxf(v[k].x, 0.f), fmaxf(v[k].y, 0.f), fmaxf(v[k].z, 0.f), fmaxf(v[k].w, 0.f));
  } else if (p.act == 3 || p.act == 4) {
    float4 u[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = NPVP_INB(k) ? ld4(p.aux_in + NPVP_ROW(k) * ld + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.act == 3) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k].x *= gelu_grad_f(u[k].x); v[k].y *= gelu_grad_f(u[k].y); v[k].z *= gelu_grad_f(u[k].z); v[k].w *= gelu_grad_f(u[k].w);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k].x = u[k].x > 0.f ? v[k].x : 0.f; v[k].y = u[k].y > 0.f ? v[k].y : 0.f;
        v[k].z = u[k].z > 0.f ? v[k].z : 0.f; v[k].w = u[k].w > 0.f ? v[k].w : 0.f;
      }
    }
  }
  if (p.drop.thresh) {
    if (p.drop.mode == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned long long key = (unsigned long long)NPVP_ROW(k) * (unsigned long long)p.N + (unsigned long long)col;
        v[k].x *= drop_scale(seed, p.drop.salt, key + 0, p.drop.thresh, p.drop.inv_keep);
        v[k].y *= drop_scale(seed, p.drop.salt, key + 1, p.drop.thresh, p.drop.inv_keep);
        v[k].z *= drop_scale(seed, p.drop.salt, key + 2, p.drop.thresh, p.drop.inv_keep);
        v[k].w *= drop_scale(seed, p.drop.salt, key + 3, p.drop.thresh, p.drop.inv_keep);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float f = k == 0 ? rowsc.x : k == 1 ? rowsc.y : k == 2 ? rowsc.z : rowsc.w;
        v[k].x *= f; v[k].y *= f; v[k].z *= f; v[k].w *= f;
      }
    }
  }
  if (p.residual) {
#pragma unroll
    for (int k = 0; k < 4; ++k) if (NPVP_INB(k)) f4_add(v[k], ld4(p.residual + NPVP_ROW(k) * p.ldr + col));
  }
  if (p.accum) {
#pragma unroll
    for (int k = 0; k < 4; ++k) if (NPVP_INB(k)) f4_add(v[k], ld4(cp + NPVP_ROW(k) * ld));
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) if (NPVP_INB(k)) st4(cp + NPVP_ROW(k) * ld, v[k]);
  if (p.c_amax) {
#pragma unroll
    for (int k = 0; k < 4; ++k) if (NPVP_INB(k)) cmax = amax4(cmax, v[k]);
  }
#undef NPVP_INB
#undef NPVP_ROW
}

// the bias-only epilogue of a tile inside the matrix (the launches with frame statistics): nothing but C = v alpha + bias
__device__ __forceinline__ void epilogue_rows_bias(const GemmParams& p, float4 (&v)[4], int row0, int col0, int lane, float& cmax, float alpha) {
  const int col = col0 + 4 * (lane & 7);
  float* cp = p.C + (long long)(row0 + (lane >> 3)) * p.ldc + col;
  const long long ld8 = 8 * p.ldc;
  const float4 bv = p.bias ? ld4(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float4 x = f4_mad(v[k], alpha, bv);
    st4(cp + k * ld8, x);
    if (p.c_amax) cmax = amax4(cmax, x);
  }
}

// scr = THIS WAVE's EPI_FLOATS floats of LDS
__device__ __forceinline__ void epilogue_tile(const GemmParams& p, const f32x16& acc, int row0, int col0, int lane, float* scr, int z,
                                              unsigned long long seed, float& cmax, float4 rowsc, float alpha) {
  if (row0 >= p.M || col0 >= p.N) return;          // (wave-uniform: a tile entirely outside the matrix)
  float4 v[4];
  epi_transpose(acc, scr, lane, v);
  const bool inside = row0 + 32 <= p.M && col0 + 32 <= p.N;
  // fast path: a tile inside the matrix with a bias-only epilogue (most forward and all plain dgrad GEMMs): the lean store
  // loop instead of the general one (whose address arithmetic for the options it does not use cost ~100 us of a
  // [114 688 x 2048]-output launch)
  const bool simple = p.splits == 1 && !p.aux_out && p.act == 0 && !p.drop.thresh && !p.residual && !p.accum;
  if (simple && inside) epilogue_rows_bias(p, v, row0, col0, lane, cmax, alpha);
  else if (inside) epilogue_rows<false>(p, v, row0, col0, lane, z, seed, cmax, rowsc, alpha);
  else epilogue_rows<true>(p, v, row0, col0, lane, z, seed, cmax, rowsc, alpha);
}

// Epilogue of the forward GEMMs that feed a frame LayerNorm (MlpDWBN fc1 -> norm1, fc2 -> norm3): C = acc*alpha + bias,
// plus, per wave, the (mean, M2) of a 64 x 64 block of outputs held as 2 x 2 accumulators.  Token rows come in frames of
// 64 and the block's 64 rows are exactly one frame (row0 % 64 == 0, M % 64 == 0), so rowstats[frame][column block of 64]
// = (mean, M2) are the partials of the frame statistics (merged by frame_stats_finalize): the LayerNorm needs no pass
// over C.  Sums are taken (in the accumulator layout) about the lane's first value and combined across the wave in Chan's
// form; fixed order, deterministic.  M % 64 == 0 and N % 64 == 0 (checked by the launcher): the block is either inside
// the matrix or outside.  The stores go through the same transposed path as every other epilogue.
__device__ __forceinline__ void epilogue_rowstats_block(const GemmParams& p, const f32x16& a00, const f32x16& a01,
                                                        const f32x16& a10, const f32x16& a11, int row0, int col0, int lane,
                                                        float* scr, float& cmax, float alpha) {
  if (row0 >= p.M || col0 >= p.N) return;
  const int r = lane & 31;
  float shift = 0.f, s1 = 0.f, s2 = 0.f;
  // statistics first (accumulator layout, nothing but three running scalars live), then the four tiles leave one at a time:
  // interleaved by the scheduler the two loops kept four transposed tiles in flight and spilled 54 VGPRs to scratch
  // (a 1.42 x write amplification of the fc1 / fc2 GEMMs in the WRITE_SIZE counter)
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const f32x16& acc = tm == 0 ? (tn == 0 ? a00 : a01) : (tn == 0 ? a10 : a11);
      const float bv = p.bias ? p.bias[col0 + tn * 32 + r] : 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const float v = acc[g] * alpha + bv;
        if (tm == 0 && tn == 0 && g == 0) shift = v;
        const float d = v - shift;
        s1 += d; s2 += d * d;
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const f32x16& acc = tm == 0 ? (tn == 0 ? a00 : a01) : (tn == 0 ? a10 : a11);
      float4 v[4];
      epi_transpose(acc, scr, lane, v);
      epilogue_rows_bias(p, v, row0 + tm * 32, col0 + tn * 32, lane, cmax, alpha);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const float n = 64.f, m1 = s1 / n, mean_l = shift + m1, m2_l = s2 - s1 * m1;       // this lane's 64 values
  const float mean_w = wave_sum(mean_l) * (1.f / 64.f);
  const float dl = mean_l - mean_w;
  const float m2_w = wave_sum(m2_l + n * dl * dl);
  if ((lane & 63) == 0) {
    const long long frame = row0 >> 6;
    const int cb = col0 >> 6, ncb = p.N >> 6;
    p.rowstats[(frame * ncb + cb) * 2] = mean_w;
    p.rowstats[(frame * ncb + cb) * 2 + 1] = m2_w;
  }
}

// XCD-aware, bijective tile remap (cdna guide T1) for an unsplit launch of gridDim.x = tiles_m * tiles_n workgroups.
// Blocks are dealt round-robin over the 8 XCDs (b and b+8 share one), each XCD has a private 4 MiB L2.  Default
// (colgroups = 1): XCD x walks a contiguous run of tiles, tile_n fastest, so an A row-panel is fetched by one XCD and
// reused from its L2 across the tile columns - but then the XCD needs ALL of B resident, and the weights of the
// 512<->2048 layers do not fit next to the streaming A panels.  colgroups = G > 1: XCD x owns column group x % G (a B
// slice that stays L2 resident) and row group x / G; A panels are then read by G XCDs.  Placement only changes speed.
__device__ __forceinline__ void tile_of_block_unsplit(const GemmParams& p, const int nwg, const int bid, int& tile_m, int& tile_n) {
  const int xcd = bid & 7, loc = bid >> 3;
  if (p.colgroups > 1) {
    const int G = p.colgroups, tn_g = p.tiles_n / G, tm_g = p.tiles_m / (8 / G);
    const int lm = loc / tn_g;
    tile_m = (xcd / G) * tm_g + lm;
    tile_n = (xcd % G) * tn_g + (loc - lm * tn_g);
  } else {
    const int q = nwg >> 3, rr = nwg & 7;
    const int nid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
    tile_m = nid / p.tiles_n;
    tile_n = nid - tile_m * p.tiles_n;
  }
}

__device__ __forceinline__ void tile_of_block_unsplit(const GemmParams& p, int& tile_m, int& tile_n) {
  tile_of_block_unsplit(p, (int)gridDim.x, (int)blockIdx.x, tile_m, tile_n);
}

// smallest G in {1,2,4,8} with b_bytes / G <= 2 MB that tiles the grid evenly (see tile_of_block_unsplit)
inline int pick_colgroups(long long b_bytes, int tiles_m, int tiles_n) {
  for (int G = 1; G <= 8; G *= 2) {
    if (b_bytes / G > (2ll << 20)) continue;
    if (G > 1 && (tiles_n % G == 0) && (tiles_m % (8 / G) == 0) && ((tiles_m * tiles_n) % 8 == 0)) return G;
    break;
  }
  return 1;
}

// gemm_wide.hip: 128 x 256 tile, 4 waves, A = fp32 [M][K] split on the fly, B = pre-split planes.  Returns true if it
// took the launch (shape / operand requirements met), false if the caller should use the 128 x 128 kernel.
bool launch_gemm_wide(GemmParams& p, hipStream_t stream);
bool gemm_wide_takes(int M, int N, int K);
int gemm_wide_variant(int M, int N, int K);      // 0 not taken, 1 = 128 x 256 tiles, 2 = 128 x 128 tiles
// weight gradients (A = dy [K][M], B = x [K][N], both fp32): split count (0 = shape not taken) and launch; the caller sets
// p.K / p.C / p.colsum for the split exactly as for the 128 x 128 kernel and runs the split-K reductions afterwards.
int wide_wgrad_splits(int M, int N, int K);
bool launch_gemm_wgrad_wide(GemmParams& p, int splits, hipStream_t stream);
// gemm_f16.hip: the two-term fp16 forms of the three (precision 6)
int gemm_f16_variant(int M, int N, int K);       // 0 not taken, 1 = 128 x 256 tiles, 2 = 128 x 128 tiles
bool launch_gemm_f16(GemmParams& p, hipStream_t stream);
int f16_wgrad_splits(int M, int N, int K);
bool launch_gemm_wgrad_f16(GemmParams& p, int splits, hipStream_t stream);

}  // namespace npvp
