// Attention cores of the predictor: softmax(q k^T / sqrt(d) + mask) v
// per (group, head), fwd and bwd, over the canonical (B*T, H*W, C) layout with the
// reference's permutes turned into index math:
//   mode 0  spatial local window (ref/models/VidHRFormer.py:274-307,447-475): group = (frame, window),
//           members = the ws*ws tokens of the window.
//   mode 1  temporal / encoder-decoder (ref/models/VidHRFormer.py:94-107,217-221,229-239): group =
//           (sample, pixel), members = the T time-steps (stride H*W*ld between them); Tq != Tk for
//           the enc-dec case; mask_mode 1 = the encoder's "no query but the last sees the last
//           time-step" mask (:100-102).
// q/k/v/o carry their own row stride so that the q and k projections can live in one
// [rows, 2C] buffer.  Sequence lengths are <= 32 and d = 64, so the core is latency bound, not MFMA
// work: one wavefront per (group, head); the shipped kernels run every 16x16 product on
// v_mfma_f32_16x16x4_f32 (exact fp32) with operands taken straight from global memory
// (attn_*_mfma_kernel) or, for the backward of 17..32 query rows, from tiles staged once in LDS
// (attn_bwd_staged1_kernel); softmax in registers, attention dropout replayed in backward from a counter
// hash.  (The first-generation LDS kernels and the two-orientation staged backward are in the history of this
// file: removed in round 4.)  Algorithmic bytes: 4*C*4 B per token fwd (q,k,v read + o write).
#include "common.h"

namespace npvp {

constexpr int HD = 64;      // head dim
constexpr int LDT = 68;     // LDS row stride of a [rows][64] tile (16-B aligned, bank-skewed)

struct AttnParams {
  const float* q; const float* k; const float* v; const float* go;   // go = dO (bwd only)
  float* o;                                                          // fwd out
  float* dq; float* dk; float* dv;                                   // bwd out
  long long ld_q, ld_k, ld_v, ld_o;                                  // row strides (floats); d* use the same
  long long ld_dq, ld_dk, ld_dv;
  int mode, heads, L, S;
  int P, W, ws, nww, nwin;        // mode 0 geometry (P = H*W)
  int Tq, Tk;                     // mode 1
  int mask_mode;
  float scale;
  unsigned int drop_thresh; float drop_inv_keep; unsigned int salt;
  const unsigned long long* seed;
  long long total;                // groups * heads
  float* o_amax; float* dq_amax; float* dk_amax; float* dv_amax;     // nullable amax slots of the outputs (common.h)
};

// The token rows of one attention group, resolved ONCE per wave.  (Until round 4 every row of every tile load and store divided by
// the window / frame geometry again - 64-bit scalar and 32-bit vector division sequences, 200+ per wave: the T = 28 backward
// executed ~15 000 scalar instructions around its 320 MFMAs and its time followed them, not the bytes.)
//   mode 1 (temporal):  a = sample, b = pixel:            row(m) = (a * Tn + m) * P + b
//   mode 0 (windows):   a = first row of the window:      row(m) = a + (m / ws) * W + m % ws
// m / ws for the few dozen rows of a window is taken as (int)((m + 0.5) / ws) in fp32: (m + 0.5) / ws is never closer than
// 1 / (2 ws) to an integer, far above the rounding of the product.  Row indices fit 32 bits (attn_setup checks).
struct AttnRows { int mode, a, b, P, W, ws; float inv_ws; };
__device__ __forceinline__ AttnRows attn_rows(const AttnParams& p, unsigned int g) {
  AttnRows r;
  r.mode = p.mode; r.P = p.P; r.W = p.W; r.ws = p.ws; r.b = 0; r.inv_ws = 0.f;
  if (p.mode == 0) {
    const unsigned int f = g / (unsigned int)p.nwin, win = g - f * (unsigned int)p.nwin;
    const unsigned int qh = win / (unsigned int)p.nww, qw = win - qh * (unsigned int)p.nww;
    r.a = (int)(f * (unsigned int)p.P + qh * (unsigned int)(p.ws * p.W) + qw * (unsigned int)p.ws);
    r.inv_ws = 1.f / (float)p.ws;
  } else {
    const unsigned int n = g / (unsigned int)p.P;
    r.a = (int)n; r.b = (int)(g - n * (unsigned int)p.P);
  }
  return r;
}
__device__ __forceinline__ long long attn_row(const AttnRows& r, int m, int Tn) {
  if (r.mode == 0) {
    const int ph = (int)(((float)m + 0.5f) * r.inv_ws);
    return r.a + ph * (r.W - r.ws) + m;
  }
  return (r.a * Tn + m) * r.P + r.b;
}

// =====================================================================================================
// MFMA form (sequences up to 32 = one or two 16-row blocks per side): no LDS at all.
// The LDS kernels above spend their time on ds_reads (every lane re-reads whole K / V rows for its dot products:
// ~400 KB of LDS traffic per wave in backward, 21.6 KB of LDS per wave -> 6 waves per CU).  Here every 16x16 product
// runs on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate) with operands loaded from global memory
// straight into the MFMA operand layout.  With lane = 16*c + n the instruction takes A[m=n][k=c], B[k=c][col=n] and
// returns D[4c+i][n] in register i.  Two tricks make every load a coalesced float4 and remove all transposes:
//   * a reduction index may be permuted freely if both operands use the same permutation:
//       "R" tiles  X_R[s]      = X[n][16c + s]      (s = 0..15, one 64-byte chunk of row n)  for products over d,
//       "G" tiles  X_G[i].blk  = X[4c+i][4n + blk]  (four float4 of rows 4c..4c+3)           for products over rows;
//     the output column of the row-reducing products is likewise d = 4n + blk, so results leave as float4 stores;
//   * the score matrix is computed in BOTH orientations (16 extra MFMAs, 4 extra exps per lane):
//       orientation A: register i = S[query n][key 4c+i]   - feeds P V and dS K     (reduction over keys)
//       orientation B: register i = S[query 4c+i][key n]   - feeds P^T dO, dS^T Q   (reduction over queries)
//     so no 16x16 transpose is ever needed; the per-query softmax statistics move by three __shfl per query.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define NPVP_MFMA16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)

struct AttnTileR { float v[16]; };
struct AttnTileG { float4 v[4]; };

// blk = 16-row block of the sequence (sequences up to 32 = two blocks); rows past the end are clamped (finite data,
// their contributions are masked or never stored)
__device__ __forceinline__ void attn_load_r(AttnTileR& t, const float* src, long long ld, const AttnParams& p, const AttnRows& g,
                                            int nrows, int Tn, int head, int n, int c, int blk) {
  const int r = min(16 * blk + n, nrows - 1);
  const float* q = src + attn_row(g, r, Tn) * ld + head * HD + 16 * c;
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    const float4 x = ld4(q + 4 * s4);
    t.v[4 * s4 + 0] = x.x; t.v[4 * s4 + 1] = x.y; t.v[4 * s4 + 2] = x.z; t.v[4 * s4 + 3] = x.w;
  }
}
__device__ __forceinline__ void attn_load_g(AttnTileG& t, const float* src, long long ld, const AttnParams& p, const AttnRows& g,
                                            int nrows, int Tn, int head, int n, int c, int blk) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = min(16 * blk + 4 * c + i, nrows - 1);
    t.v[i] = ld4(src + attn_row(g, r, Tn) * ld + head * HD + 4 * n);
  }
}
// D rows 16 blk + 4c+i, columns d = 4n + b: one float4 per row
__device__ __forceinline__ void attn_store_d(const f32x4_t (&acc)[4], float* dst, long long ld, const AttnParams& p, const AttnRows& g,
                                             int nrows, int Tn, int head, int n, int c, int blk, float& am) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 16 * blk + 4 * c + i;
    if (r < nrows) {
      const float4 v = make_float4(acc[0][i], acc[1][i], acc[2][i], acc[3][i]);
      st4(dst + attn_row(g, r, Tn) * ld + head * HD + 4 * n, v);
      am = amax4(am, v);
    }
  }
}
__device__ __forceinline__ float blk_of(const float4& x, int b) { return b == 0 ? x.x : b == 1 ? x.y : b == 2 ? x.z : x.w; }

// acc[b] += sum_i A[i] (x) G[i].b : the row-reducing product (reduction index = 4c+i on both sides)
__device__ __forceinline__ void attn_mm_rows(f32x4_t (&acc)[4], const float (&a)[4], const AttnTileG& gt) {
#pragma unroll
  for (int b = 0; b < 4; ++b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[b] = NPVP_MFMA16(a[i], blk_of(gt.v[i], b), acc[b]);
  }
}
__device__ __forceinline__ void attn_zero(f32x4_t (&acc)[4]) {
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ f32x4_t attn_mm_d(const AttnTileR& a, const AttnTileR& b) {
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 16; ++s) acc = NPVP_MFMA16(a.v[s], b.v[s], acc);
  return acc;
}
__device__ __forceinline__ float quad_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float quad_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

// NQ / NK = number of 16-row blocks of the query / key sequence (1 or 2)
template <int NQ, int NK>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(AttnParams p) {
  const int lane = threadIdx.x & 63, n = lane & 15, c = lane >> 4;
  const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= p.total) return;                      // waves are independent: no barriers in this kernel
  const int head = (int)((unsigned int)wid % (unsigned int)p.heads);
  const AttnRows g = attn_rows(p, (unsigned int)wid / (unsigned int)p.heads);
  const int L = p.L, S = p.S;
  const int Tq = p.mode == 1 ? p.Tq : 0, Tk = p.mode == 1 ? p.Tk : 0;
  const unsigned long long seed = (p.seed && p.drop_thresh) ? *p.seed : 0ull;
  AttnTileR kr[NK];
  AttnTileG vg[NK];
  float am_o = 0.f;
  const unsigned int pk_o = amax_peek_wave(p.o_amax);
#pragma unroll
  for (int kb = 0; kb < NK; ++kb) {
    attn_load_r(kr[kb], p.k, p.ld_k, p, g, S, Tk, head, n, c, kb);
    attn_load_g(vg[kb], p.v, p.ld_v, p, g, S, Tk, head, n, c, kb);
  }
#pragma unroll
  for (int qb = 0; qb < NQ; ++qb) {
    AttnTileR qr;
    attn_load_r(qr, p.q, p.ld_q, p, g, L, Tq, head, n, c, qb);
    __builtin_amdgcn_sched_barrier(0);      // all tile loads in flight before the first MFMA waits for one of them
    const int q = 16 * qb + n, qn = min(q, L - 1);
    // orientation A: sc[kb][i] = S[query q][key 16kb + 4c+i]
    float sc[NK][4], mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NK; ++kb) {
      const f32x4_t sa = attn_mm_d(kr[kb], qr);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 16 * kb + 4 * c + i;
        float s = sa[i] * p.scale;
        if (j >= S || (p.mask_mode == 1 && j == S - 1 && q < L - 1)) s = -INFINITY;
        sc[kb][i] = s; mx = fmaxf(mx, s);
      }
    }
    mx = quad_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NK; ++kb)
#pragma unroll
      for (int i = 0; i < 4; ++i) { sc[kb][i] = (sc[kb][i] == -INFINITY) ? 0.f : __expf(sc[kb][i] - mx); sum += sc[kb][i]; }
    sum = quad_sum(sum);
    const float inv = 1.f / sum;
    f32x4_t o[4];
    attn_zero(o);
#pragma unroll
    for (int kb = 0; kb < NK; ++kb) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 16 * kb + 4 * c + i;
        sc[kb][i] *= inv;
        if (p.drop_thresh && j < S)
          sc[kb][i] *= drop_scale(seed, p.salt, ((unsigned long long)wid * L + qn) * S + j, p.drop_thresh, p.drop_inv_keep);
      }
      attn_mm_rows(o, sc[kb], vg[kb]);
    }
    attn_store_d(o, p.o, p.ld_o, p, g, L, Tq, head, n, c, qb, am_o);
  }
  amax_slot_commit(p.o_amax, am_o, pk_o);
}

template <int NQ, int NK>
__global__ __launch_bounds__(256) void attn_bwd_mfma_kernel(AttnParams p) {
  const int lane = threadIdx.x & 63, n = lane & 15, c = lane >> 4;
  const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= p.total) return;
  const int head = (int)((unsigned int)wid % (unsigned int)p.heads);
  const AttnRows g = attn_rows(p, (unsigned int)wid / (unsigned int)p.heads);
  const int L = p.L, S = p.S;
  const int Tq = p.mode == 1 ? p.Tq : 0, Tk = p.mode == 1 ? p.Tk : 0;
  const unsigned long long seed = (p.seed && p.drop_thresh) ? *p.seed : 0ull;
  AttnTileR kr[NK], vr[NK];
#pragma unroll
  for (int kb = 0; kb < NK; ++kb) {
    attn_load_r(kr[kb], p.k, p.ld_k, p, g, S, Tk, head, n, c, kb);
    attn_load_r(vr[kb], p.v, p.ld_v, p, g, S, Tk, head, n, c, kb);
  }
  // ---- orientation A per query block: softmax statistics of query 16qb+n (kept for phase B), dQ
  float mxs[NQ], invs[NQ], rss[NQ], am_q = 0.f, am_k = 0.f, am_v = 0.f;
  const unsigned int pk_q = amax_peek_wave(p.dq_amax), pk_k = amax_peek_wave(p.dk_amax), pk_v = amax_peek_wave(p.dv_amax);
  AttnTileR qr, gr;               // with a single query block the tiles of phase A are reused by phase B
#pragma unroll
  for (int qb = 0; qb < NQ; ++qb) {
    attn_load_r(qr, p.q, p.ld_q, p, g, L, Tq, head, n, c, qb);
    attn_load_r(gr, p.go, p.ld_o, p, g, L, Tq, head, n, c, qb);
    AttnTileG kg[NK];
#pragma unroll
    for (int kb = 0; kb < NK; ++kb) attn_load_g(kg[kb], p.k, p.ld_k, p, g, S, Tk, head, n, c, kb);
    __builtin_amdgcn_sched_barrier(0);      // all tile loads in flight before the first MFMA waits for one of them
    const int q = 16 * qb + n, qn = min(q, L - 1);
    float sc[NK][4], dp[NK][4], mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NK; ++kb) {
      const f32x4_t sa = attn_mm_d(kr[kb], qr), da = attn_mm_d(vr[kb], gr);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 16 * kb + 4 * c + i;
        float s = sa[i] * p.scale;
        if (j >= S || (p.mask_mode == 1 && j == S - 1 && q < L - 1)) s = -INFINITY;
        sc[kb][i] = s; dp[kb][i] = da[i]; mx = fmaxf(mx, s);
      }
    }
    mx = quad_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NK; ++kb)
#pragma unroll
      for (int i = 0; i < 4; ++i) { sc[kb][i] = (sc[kb][i] == -INFINITY) ? 0.f : __expf(sc[kb][i] - mx); sum += sc[kb][i]; }
    sum = quad_sum(sum);
    const float inv = 1.f / sum;
    float rs = 0.f;
#pragma unroll
    for (int kb = 0; kb < NK; ++kb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 16 * kb + 4 * c + i;
        float m = 1.f;
        if (p.drop_thresh && j < S)
          m = drop_scale(seed, p.salt, ((unsigned long long)wid * L + qn) * S + j, p.drop_thresh, p.drop_inv_keep);
        sc[kb][i] *= inv;
        dp[kb][i] *= m;
        rs += dp[kb][i] * sc[kb][i];
      }
    rs = quad_sum(rs);
    mxs[qb] = mx; invs[qb] = inv; rss[qb] = rs;
    f32x4_t dq[4];
    attn_zero(dq);
#pragma unroll
    for (int kb = 0; kb < NK; ++kb) {
      float ds[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) ds[i] = sc[kb][i] * (dp[kb][i] - rs) * p.scale;
      attn_mm_rows(dq, ds, kg[kb]);                              // dQ[q][d] += sum_j dS[q][j] K[j][d]
    }
    attn_store_d(dq, p.dq, p.ld_dq, p, g, L, Tq, head, n, c, qb, am_q);
  }
  // ---- orientation B per key block: register i <-> (query 16qb + 4c+i, key 16kb + n): dK, dV
#pragma unroll
  for (int kb = 0; kb < NK; ++kb) {
    f32x4_t dv[4], dk[4];
    attn_zero(dv); attn_zero(dk);
    const int key = 16 * kb + n;
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
      if constexpr (NQ > 1) {
        attn_load_r(qr, p.q, p.ld_q, p, g, L, Tq, head, n, c, qb);
        attn_load_r(gr, p.go, p.ld_o, p, g, L, Tq, head, n, c, qb);
      }
      AttnTileG gg, qg;
      attn_load_g(gg, p.go, p.ld_o, p, g, L, Tq, head, n, c, qb);
      attn_load_g(qg, p.q, p.ld_q, p, g, L, Tq, head, n, c, qb);
      __builtin_amdgcn_sched_barrier(0);
      const f32x4_t sb = attn_mm_d(qr, kr[kb]), db = attn_mm_d(gr, vr[kb]);
      float pd[4], ds[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ql = 4 * c + i, q = 16 * qb + ql;             // statistics of query q live in lane ql of block qb
        const float mxq = __shfl(mxs[qb], ql, 64), invq = __shfl(invs[qb], ql, 64), rsq = __shfl(rss[qb], ql, 64);
        const bool dead = key >= S || q >= L || (p.mask_mode == 1 && key == S - 1 && q < L - 1);
        const float pr = dead ? 0.f : __expf(sb[i] * p.scale - mxq) * invq;
        float m = 1.f;
        if (p.drop_thresh && !dead)
          m = drop_scale(seed, p.salt, ((unsigned long long)wid * L + q) * S + key, p.drop_thresh, p.drop_inv_keep);
        pd[i] = pr * m;
        ds[i] = pr * (db[i] * m - rsq) * p.scale;
      }
      attn_mm_rows(dv, pd, gg);                                  // dV[j][d] += sum_q Pd[q][j] dO[q][d]
      attn_mm_rows(dk, ds, qg);                                  // dK[j][d] += sum_q dS[q][j] Q[q][d]
    }
    attn_store_d(dv, p.dv, p.ld_dv, p, g, S, Tk, head, n, c, kb, am_v);
    attn_store_d(dk, p.dk, p.ld_dk, p, g, S, Tk, head, n, c, kb, am_k);
  }
  amax_slot_commit(p.dq_amax, am_q, pk_q);
  amax_slot_commit(p.dk_amax, am_k, pk_k);
  amax_slot_commit(p.dv_amax, am_v, pk_v);
}

// ---- backward for query sequences of 17..32 rows (the c2 decoder: T = 28): Q, dO and K staged ONCE in LDS.
// attn_bwd_mfma_kernel<2,NK> needs each of Q / dO / K in both operand layouts and in both phases: ten tile loads in seven
// dependent load phases at 248 VGPRs (2 waves / SIMD) - 1.0 ms for 1.64 GB at c2 (1.6 TB/s).  Here one wavefront = one
// workgroup issues ALL its global loads up front (each tile once, 4 rows x 256 B per instruction), parks Q / dO / K in its own
// (2L + S) x 272 B of LDS and takes every "R" and "G" operand tile from there (row stride 68 floats: both read patterns are
// bank-conflict free); V is only ever needed as "R" tiles and stays in registers.  No barrier: the wave is alone in its
// workgroup and LDS operations of one wave execute in order.  22.8 KB of LDS at T = 28 -> 7 waves per CU.
template <int NIT>
__device__ __forceinline__ void attn_stage_load(float4 (&r)[NIT], const float* src, long long ld, const AttnParams& p, const AttnRows& g,
                                                int nrows, int Tn, int head, int lane) {
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = 4 * it + (lane >> 4);
    if (row < nrows) r[it] = ld4(src + attn_row(g, row, Tn) * ld + head * HD + (lane & 15) * 4);
  }
}
template <int NIT>
__device__ __forceinline__ void attn_stage_store(float* dst, const float4 (&r)[NIT], int nrows, int lane) {
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = 4 * it + (lane >> 4);
    if (row < nrows) st4(dst + row * LDT + (lane & 15) * 4, r[it]);
  }
}
__device__ __forceinline__ void attn_lds_r(AttnTileR& t, const float* xs, int nrows, int n, int c, int blk) {
  const float* q = xs + min(16 * blk + n, nrows - 1) * LDT + 16 * c;
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    const float4 x = ld4(q + 4 * s4);
    t.v[4 * s4 + 0] = x.x; t.v[4 * s4 + 1] = x.y; t.v[4 * s4 + 2] = x.z; t.v[4 * s4 + 3] = x.w;
  }
}
__device__ __forceinline__ void attn_lds_g(AttnTileG& t, const float* xs, int nrows, int n, int c, int blk) {
#pragma unroll
  for (int i = 0; i < 4; ++i) t.v[i] = ld4(xs + min(16 * blk + 4 * c + i, nrows - 1) * LDT + 4 * n);
}

// ---- ONE score orientation.  The row-reducing products P^T dO and dS^T Q want P / dS with the QUERY index in registers, the
// softmax statistics and dQ want queries on lanes: the first staged kernel (round 2, removed in round 4) therefore evaluated
// S = Q K^T and dP = dO V^T twice.  Here they are evaluated once, in the query-on-lanes
// orientation, and the two 16 x 16 blocks a (query block, key block) pair produces - P (dropout mask applied) and dS - are
// transposed through a 2 x 16 x 20-float LDS scratch (1 ds_write_b128 + 4 ds_read_b32 per lane and matrix).  448 -> 320
// MFMAs per wave, and the softmax exponentials and the dropout hash - half of the staged kernel's VALU work - are evaluated
// once per element instead of twice.  dV / dK accumulate over the query blocks in registers and leave at the end.
constexpr int LDX = 20;      // row stride of the transposition scratch (16-B aligned rows)

template <int NQ, int NK>
__global__ __launch_bounds__(64, 2) void attn_bwd_staged1_kernel(AttnParams p) {      // (2 waves / SIMD: <= 256 registers, VGPR + AGPR)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x, n = lane & 15, c = lane >> 4;
  const long long wid = blockIdx.x;
  const int head = (int)((unsigned int)wid % (unsigned int)p.heads);
  const AttnRows g = attn_rows(p, (unsigned int)wid / (unsigned int)p.heads);
  const int L = p.L, S = p.S;
  const int Tq = p.mode == 1 ? p.Tq : 0, Tk = p.mode == 1 ? p.Tk : 0;
  const unsigned long long seed = (p.seed && p.drop_thresh) ? *p.seed : 0ull;
  float* Qs = smem;
  float* Gs = Qs + L * LDT;
  float* Ks = Gs + L * LDT;
  float* Xp = Ks + S * LDT;          // [16][LDX] P (masked) of the current block pair, query-major
  float* Xd = Xp + 16 * LDX;         // [16][LDX] dS
  AttnTileR vr[NK];
  {
    float4 sq[4 * NQ], sg[4 * NQ], sk[4 * NK];
    attn_stage_load<4 * NK>(sk, p.k, p.ld_k, p, g, S, Tk, head, lane);
    attn_stage_load<4 * NQ>(sq, p.q, p.ld_q, p, g, L, Tq, head, lane);
    attn_stage_load<4 * NQ>(sg, p.go, p.ld_o, p, g, L, Tq, head, lane);
#pragma unroll
    for (int kb = 0; kb < NK; ++kb) attn_load_r(vr[kb], p.v, p.ld_v, p, g, S, Tk, head, n, c, kb);
    __builtin_amdgcn_sched_barrier(0);      // every global load of this wave is in flight before the first wait
    attn_stage_store<4 * NK>(Ks, sk, S, lane);
    attn_stage_store<4 * NQ>(Qs, sq, L, lane);
    attn_stage_store<4 * NQ>(Gs, sg, L, lane);
  }
  __syncthreads();                          // one wave per workgroup: orders the LDS writes before the reads below
  float am_q = 0.f, am_k = 0.f, am_v = 0.f;
  const unsigned int pk_q = amax_peek_wave(p.dq_amax), pk_k = amax_peek_wave(p.dk_amax), pk_v = amax_peek_wave(p.dv_amax);
  f32x4_t dv[NK][4], dk[NK][4];
#pragma unroll
  for (int kb = 0; kb < NK; ++kb) { attn_zero(dv[kb]); attn_zero(dk[kb]); }
#pragma unroll
  for (int qb = 0; qb < NQ; ++qb) {
    AttnTileR qr, gr;
    attn_lds_r(qr, Qs, L, n, c, qb);
    attn_lds_r(gr, Gs, L, n, c, qb);
    const int q = 16 * qb + n, qn = min(q, L - 1);
    float sc[NK][4], dp[NK][4], mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NK; ++kb) {
      AttnTileR kr;
      attn_lds_r(kr, Ks, S, n, c, kb);
      const f32x4_t sa = attn_mm_d(kr, qr), da = attn_mm_d(vr[kb], gr);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 16 * kb + 4 * c + i;
        float s = sa[i] * p.scale;
        if (j >= S || (p.mask_mode == 1 && j == S - 1 && q < L - 1)) s = -INFINITY;
        sc[kb][i] = s; dp[kb][i] = da[i]; mx = fmaxf(mx, s);
      }
    }
    mx = quad_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NK; ++kb)
#pragma unroll
      for (int i = 0; i < 4; ++i) { sc[kb][i] = (sc[kb][i] == -INFINITY) ? 0.f : __expf(sc[kb][i] - mx); sum += sc[kb][i]; }
    sum = quad_sum(sum);
    const float inv = 1.f / sum;
    float rs = 0.f, pm[NK][4];
#pragma unroll
    for (int kb = 0; kb < NK; ++kb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 16 * kb + 4 * c + i;
        float m = 1.f;
        if (p.drop_thresh && j < S)
          m = drop_scale(seed, p.salt, ((unsigned long long)wid * L + qn) * S + j, p.drop_thresh, p.drop_inv_keep);
        sc[kb][i] *= inv;
        pm[kb][i] = sc[kb][i] * m;              // P with the dropout mask: what multiplies dO in dV
        dp[kb][i] *= m;
        rs += dp[kb][i] * sc[kb][i];
      }
    rs = quad_sum(rs);
    const bool live = q < L;                    // rows past the end (clamped loads) must not reach dK / dV
    f32x4_t dq[4];
    attn_zero(dq);
    AttnTileG gg, qg;
    attn_lds_g(gg, Gs, L, n, c, qb);
    attn_lds_g(qg, Qs, L, n, c, qb);
#pragma unroll
    for (int kb = 0; kb < NK; ++kb) {
      float ds[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) ds[i] = sc[kb][i] * (dp[kb][i] - rs) * p.scale;
      AttnTileG kg;
      attn_lds_g(kg, Ks, S, n, c, kb);
      attn_mm_rows(dq, ds, kg);                                  // dQ[q][d] += sum_j dS[q][j] K[j][d]
      // transpose the pair's P and dS blocks: lane (n, c) holds M[query n][key 4c+i], dK / dV want M[query 4c+i][key n]
      st4(Xp + n * LDX + 4 * c, live ? make_float4(pm[kb][0], pm[kb][1], pm[kb][2], pm[kb][3]) : make_float4(0.f, 0.f, 0.f, 0.f));
      st4(Xd + n * LDX + 4 * c, live ? make_float4(ds[0], ds[1], ds[2], ds[3]) : make_float4(0.f, 0.f, 0.f, 0.f));
      float pT[4], dT[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { pT[i] = Xp[(4 * c + i) * LDX + n]; dT[i] = Xd[(4 * c + i) * LDX + n]; }
      attn_mm_rows(dv[kb], pT, gg);                              // dV[j][d] += sum_q Pd[q][j] dO[q][d]
      attn_mm_rows(dk[kb], dT, qg);                              // dK[j][d] += sum_q dS[q][j] Q[q][d]
    }
    attn_store_d(dq, p.dq, p.ld_dq, p, g, L, Tq, head, n, c, qb, am_q);
  }
#pragma unroll
  for (int kb = 0; kb < NK; ++kb) {
    attn_store_d(dv[kb], p.dv, p.ld_dv, p, g, S, Tk, head, n, c, kb, am_v);
    attn_store_d(dk[kb], p.dk, p.ld_dk, p, g, S, Tk, head, n, c, kb, am_k);
  }
  amax_slot_commit(p.dq_amax, am_q, pk_q);
  amax_slot_commit(p.dk_amax, am_k, pk_k);
  amax_slot_commit(p.dv_amax, am_v, pk_v);
}

// =====================================================================================================
// Generic form: sequence lengths 33 .. 128 (round 5).  The reference has no limit on the sequence length (ref
// VidHRFormer.py:94-107: nn.MultiheadAttention over T frames); every shipped configuration has T <= 30 and runs on the MFMA
// kernels above, whose operand layouts are written for one or two 16-row blocks per side.  Longer sequences - a clip of 40 frames,
// an 8 x 8 window - take these kernels instead of being refused: one workgroup per (group, head), K / V (backward: Q, K, V, dO) of
// that head staged in LDS, one thread per query row (backward: per query row, then per key row), scalar fp32 arithmetic, the
// softmax statistics recomputed per pass instead of stored.  Same mask function, same dropout keys, same amax commits as the MFMA
// kernels.  Correct, deterministic, not fast (O(L S d) scalar multiply-adds per pass): a correctness route, not a tuned one.
constexpr int GEN_MAX = 128, GEN_THREADS = 128;

__device__ __forceinline__ void gen_stage(float* dst, const float* src, long long ld, const AttnParams& p, const AttnRows& g, int rows_n,
                                          int Tn, int head) {
  for (int i = threadIdx.x; i < rows_n * (HD / 4); i += GEN_THREADS) {
    const int j = i / (HD / 4), c4 = i - j * (HD / 4);
    *reinterpret_cast<float4*>(dst + j * HD + 4 * c4) = ld4(src + attn_row(g, j, Tn) * ld + head * HD + 4 * c4);
  }
}
__device__ __forceinline__ float gen_dot(const float* a, const float* b) {        // a in registers / LDS, b in LDS: 64-term dot product
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    const float4 x = *reinterpret_cast<const float4*>(a + d), y = *reinterpret_cast<const float4*>(b + d);
    s += x.x * y.x; s += x.y * y.y; s += x.z * y.z; s += x.w * y.w;
  }
  return s;
}

__global__ __launch_bounds__(GEN_THREADS) void attn_fwd_generic_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  const long long wid = blockIdx.x;
  const int head = (int)((unsigned int)wid % (unsigned int)p.heads);
  const AttnRows g = attn_rows(p, (unsigned int)wid / (unsigned int)p.heads);
  const int L = p.L, S = p.S;
  const int Tq = p.mode == 1 ? p.Tq : 0, Tk = p.mode == 1 ? p.Tk : 0;
  const unsigned long long seed = (p.seed && p.drop_thresh) ? *p.seed : 0ull;
  float* Ks = gsm; float* Vs = gsm + S * HD;
  gen_stage(Ks, p.k, p.ld_k, p, g, S, Tk, head);
  gen_stage(Vs, p.v, p.ld_v, p, g, S, Tk, head);
  __syncthreads();
  float am_o = 0.f;
  const unsigned int pk_o = amax_peek_wave(p.o_amax);
  for (int q = threadIdx.x; q < L; q += GEN_THREADS) {
    float qr[HD];
    const float* qp = p.q + attn_row(g, q, Tq) * p.ld_q + head * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4) { const float4 t = ld4(qp + d); qr[d] = t.x; qr[d + 1] = t.y; qr[d + 2] = t.z; qr[d + 3] = t.w; }
    float mx = -INFINITY;
    for (int j = 0; j < S; ++j) {
      if (p.mask_mode == 1 && j == S - 1 && q < L - 1) continue;
      mx = fmaxf(mx, gen_dot(qr, Ks + j * HD) * p.scale);
    }
    float sum = 0.f;
    for (int j = 0; j < S; ++j) {
      if (p.mask_mode == 1 && j == S - 1 && q < L - 1) continue;
      sum += __expf(gen_dot(qr, Ks + j * HD) * p.scale - mx);
    }
    const float inv = 1.f / sum;
    float o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    for (int j = 0; j < S; ++j) {
      if (p.mask_mode == 1 && j == S - 1 && q < L - 1) continue;
      float pj = __expf(gen_dot(qr, Ks + j * HD) * p.scale - mx) * inv;
      if (p.drop_thresh) pj *= drop_scale(seed, p.salt, ((unsigned long long)wid * L + q) * S + j, p.drop_thresh, p.drop_inv_keep);
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] += pj * Vs[j * HD + d];
    }
    float* op = p.o + attn_row(g, q, Tq) * p.ld_o + head * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      st4(op + d, make_float4(o[d], o[d + 1], o[d + 2], o[d + 3]));
      am_o = fmaxf(fmaxf(am_o, fmaxf(fabsf(o[d]), fabsf(o[d + 1]))), fmaxf(fabsf(o[d + 2]), fabsf(o[d + 3])));
    }
  }
  amax_slot_commit(p.o_amax, am_o, pk_o);
}

__global__ __launch_bounds__(GEN_THREADS) void attn_bwd_generic_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  const long long wid = blockIdx.x;
  const int head = (int)((unsigned int)wid % (unsigned int)p.heads);
  const AttnRows g = attn_rows(p, (unsigned int)wid / (unsigned int)p.heads);
  const int L = p.L, S = p.S;
  const int Tq = p.mode == 1 ? p.Tq : 0, Tk = p.mode == 1 ? p.Tk : 0;
  const unsigned long long seed = (p.seed && p.drop_thresh) ? *p.seed : 0ull;
  float* Qs = gsm; float* Gs = Qs + L * HD; float* Ks = Gs + L * HD; float* Vs = Ks + S * HD; float* st = Vs + S * HD;   // st [L][3]
  gen_stage(Qs, p.q, p.ld_q, p, g, L, Tq, head);
  gen_stage(Gs, p.go, p.ld_o, p, g, L, Tq, head);
  gen_stage(Ks, p.k, p.ld_k, p, g, S, Tk, head);
  gen_stage(Vs, p.v, p.ld_v, p, g, S, Tk, head);
  __syncthreads();
  float am_q = 0.f, am_k = 0.f, am_v = 0.f;
  const unsigned int pk_q = amax_peek_wave(p.dq_amax), pk_k = amax_peek_wave(p.dk_amax), pk_v = amax_peek_wave(p.dv_amax);
  auto dead = [&](int q, int j) { return p.mask_mode == 1 && j == S - 1 && q < L - 1; };
  auto mask = [&](int q, int j) {
    return p.drop_thresh ? drop_scale(seed, p.salt, ((unsigned long long)wid * L + q) * S + j, p.drop_thresh, p.drop_inv_keep) : 1.f;
  };
  // phase A: one thread per query row - softmax statistics, delta = sum_j P dP, dQ = scale * sum_j dS K
  for (int q = threadIdx.x; q < L; q += GEN_THREADS) {
    const float* qr = Qs + q * HD; const float* gr = Gs + q * HD;
    float mx = -INFINITY;
    for (int j = 0; j < S; ++j) if (!dead(q, j)) mx = fmaxf(mx, gen_dot(qr, Ks + j * HD) * p.scale);
    float sum = 0.f;
    for (int j = 0; j < S; ++j) if (!dead(q, j)) sum += __expf(gen_dot(qr, Ks + j * HD) * p.scale - mx);
    const float inv = 1.f / sum;
    float delta = 0.f;
    for (int j = 0; j < S; ++j) {
      if (dead(q, j)) continue;
      const float pj = __expf(gen_dot(qr, Ks + j * HD) * p.scale - mx) * inv;
      delta += pj * gen_dot(gr, Vs + j * HD) * mask(q, j);
    }
    float dq[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) dq[d] = 0.f;
    for (int j = 0; j < S; ++j) {
      if (dead(q, j)) continue;
      const float pj = __expf(gen_dot(qr, Ks + j * HD) * p.scale - mx) * inv;
      const float ds = pj * (gen_dot(gr, Vs + j * HD) * mask(q, j) - delta) * p.scale;
#pragma unroll
      for (int d = 0; d < HD; ++d) dq[d] += ds * Ks[j * HD + d];
    }
    st[q * 3] = mx; st[q * 3 + 1] = inv; st[q * 3 + 2] = delta;
    float* dp = p.dq + attn_row(g, q, Tq) * p.ld_dq + head * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      st4(dp + d, make_float4(dq[d], dq[d + 1], dq[d + 2], dq[d + 3]));
      am_q = fmaxf(fmaxf(am_q, fmaxf(fabsf(dq[d]), fabsf(dq[d + 1]))), fmaxf(fabsf(dq[d + 2]), fabsf(dq[d + 3])));
    }
  }
  __syncthreads();
  // phase B: one thread per key row - dV = sum_q (P mask) dO, dK = scale * sum_q dS Q
  for (int j = threadIdx.x; j < S; j += GEN_THREADS) {
    const float* kr = Ks + j * HD; const float* vr = Vs + j * HD;
    float dk[HD], dv[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
    for (int q = 0; q < L; ++q) {
      if (dead(q, j)) continue;
      const float pj = __expf(gen_dot(Qs + q * HD, kr) * p.scale - st[q * 3]) * st[q * 3 + 1];
      const float m = mask(q, j);
      const float ds = pj * (gen_dot(Gs + q * HD, vr) * m - st[q * 3 + 2]) * p.scale, pd = pj * m;
#pragma unroll
      for (int d = 0; d < HD; ++d) { dv[d] += pd * Gs[q * HD + d]; dk[d] += ds * Qs[q * HD + d]; }
    }
    float* kp = p.dk + attn_row(g, j, Tk) * p.ld_dk + head * HD;
    float* vp = p.dv + attn_row(g, j, Tk) * p.ld_dv + head * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      st4(kp + d, make_float4(dk[d], dk[d + 1], dk[d + 2], dk[d + 3]));
      st4(vp + d, make_float4(dv[d], dv[d + 1], dv[d + 2], dv[d + 3]));
      am_k = fmaxf(fmaxf(am_k, fmaxf(fabsf(dk[d]), fabsf(dk[d + 1]))), fmaxf(fabsf(dk[d + 2]), fabsf(dk[d + 3])));
      am_v = fmaxf(fmaxf(am_v, fmaxf(fabsf(dv[d]), fabsf(dv[d + 1]))), fmaxf(fabsf(dv[d + 2]), fabsf(dv[d + 3])));
    }
  }
  amax_slot_commit(p.dq_amax, am_q, pk_q);
  amax_slot_commit(p.dk_amax, am_k, pk_k);
  amax_slot_commit(p.dv_amax, am_v, pk_v);
}

static int attn_setup(AttnParams& p, int mode, int heads, int head_dim, int frames_or_N, int P, int W, int ws, int Tq,
                      int Tk, int mask_mode, float drop_p, const unsigned long long* seed, unsigned int salt, bool bwd) {
  if (head_dim != HD) { npvp_set_error("attn: head_dim must be 64"); return NPVP_ERR_ARG; }
  if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !seed)) { npvp_set_error("attn: bad dropout arguments"); return NPVP_ERR_ARG; }
  p.mode = mode; p.heads = heads; p.P = P; p.W = W; p.ws = ws; p.Tq = Tq; p.Tk = Tk; p.mask_mode = mask_mode;
  long long groups;
  if (mode == 0) {
    if (ws <= 0 || W % ws != 0 || P % W != 0 || (P / W) % ws != 0) { npvp_set_error("attn: window must tile the grid"); return NPVP_ERR_ARG; }
    p.nww = W / ws; p.nwin = p.nww * ((P / W) / ws); p.L = p.S = ws * ws;
    groups = (long long)frames_or_N * p.nwin;
  } else {
    p.nww = p.nwin = 1; p.L = Tq; p.S = Tk;
    groups = (long long)frames_or_N * P;
  }
  if (p.L < 1 || p.S < 1 || p.L > GEN_MAX || p.S > GEN_MAX) { npvp_set_error("attn: sequence length must be in [1,128]"); return NPVP_ERR_ARG; }
  p.scale = 0.125f;   // 1/sqrt(64)
  p.drop_thresh = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
  p.drop_inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  p.salt = salt; p.seed = seed;
  p.total = groups * heads;
  // the kernels resolve (group, head) and token rows in 32-bit arithmetic (AttnRows)
  const long long rows = mode == 0 ? (long long)frames_or_N * P : (long long)frames_or_N * P * (Tq > Tk ? Tq : Tk);
  if (p.total >= (1ll << 31) || rows >= (1ll << 31)) { npvp_set_error("attn: too many token rows / (group, head) pairs for one launch"); return NPVP_ERR_ARG; }
  (void)bwd;
  return NPVP_OK;
}

}  // namespace npvp

using namespace npvp;

// mode 0: dim0 = frames (N*T); mode 1: dim0 = N.  See include/npvp_hip.h.
extern "C" int npvp_attn_fwd(const float* q, long long ld_q, const float* k, long long ld_k, const float* v, long long ld_v,
                             float* o, long long ld_o, int mode, int dim0, int P, int W, int ws, int Tq, int Tk, int heads,
                             int head_dim, int mask_mode, float drop_p, const unsigned long long* seed, unsigned int salt,
                             float* o_amax, hipStream_t stream) {
  AttnParams p = {};
  const int rc = attn_setup(p, mode, heads, head_dim, dim0, P, W, ws, Tq, Tk, mask_mode, drop_p, seed, salt, false);
  if (rc) return rc;
  p.o_amax = o_amax;
  NPVP_CHECK_ARG(dim0 > 0, "attn: empty batch");
  NPVP_CHECK_ARG(ld_q % 4 == 0 && ld_k % 4 == 0 && ld_v % 4 == 0 && ld_o % 4 == 0, "attn: row strides must be multiples of 4");
  p.q = q; p.k = k; p.v = v; p.o = o; p.ld_q = ld_q; p.ld_k = ld_k; p.ld_v = ld_v; p.ld_o = ld_o;
  const dim3 mg((unsigned)((p.total + 3) / 4)), mb(256);
  const int nq = (p.L + 15) / 16, nk = (p.S + 15) / 16;
  if (nq > 2 || nk > 2) {                  // 33 .. 128: the generic kernels (K, V of one head in LDS)
    const size_t lds = (size_t)2 * p.S * HD * sizeof(float);
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)attn_fwd_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      npvp_set_error("attn: could not reserve LDS for the generic kernel"); return NPVP_ERR_LAUNCH;
    }
    NPVP_LAUNCH(attn_fwd_generic_kernel, dim3((unsigned)p.total), dim3(GEN_THREADS), lds, stream, p);
    NPVP_CHECK_LAUNCH();
    return NPVP_OK;
  }
  if (nq == 1 && nk == 1) NPVP_LAUNCH((attn_fwd_mfma_kernel<1, 1>), mg, mb, 0, stream, p);
  else if (nq == 1) NPVP_LAUNCH((attn_fwd_mfma_kernel<1, 2>), mg, mb, 0, stream, p);
  else if (nk == 1) NPVP_LAUNCH((attn_fwd_mfma_kernel<2, 1>), mg, mb, 0, stream, p);
  else NPVP_LAUNCH((attn_fwd_mfma_kernel<2, 2>), mg, mb, 0, stream, p);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

extern "C" int npvp_attn_bwd(const float* q, long long ld_q, const float* k, long long ld_k, const float* v, long long ld_v,
                             const float* go, long long ld_o, float* dq, long long ld_dq, float* dk, long long ld_dk,
                             float* dv, long long ld_dv, int mode, int dim0, int P, int W, int ws, int Tq, int Tk,
                             int heads, int head_dim, int mask_mode, float drop_p, const unsigned long long* seed,
                             unsigned int salt, float* dq_amax, float* dk_amax, float* dv_amax, hipStream_t stream) {
  AttnParams p = {};
  const int rc = attn_setup(p, mode, heads, head_dim, dim0, P, W, ws, Tq, Tk, mask_mode, drop_p, seed, salt, true);
  if (rc) return rc;
  p.dq_amax = dq_amax; p.dk_amax = dk_amax; p.dv_amax = dv_amax;
  NPVP_CHECK_ARG(dim0 > 0, "attn_bwd: empty batch");
  NPVP_CHECK_ARG(ld_q % 4 == 0 && ld_k % 4 == 0 && ld_v % 4 == 0 && ld_o % 4 == 0 && ld_dq % 4 == 0 && ld_dk % 4 == 0 &&
                     ld_dv % 4 == 0, "attn_bwd: row strides must be multiples of 4");
  p.q = q; p.k = k; p.v = v; p.go = go; p.dq = dq; p.dk = dk; p.dv = dv;
  p.ld_q = ld_q; p.ld_k = ld_k; p.ld_v = ld_v; p.ld_o = ld_o; p.ld_dq = ld_dq; p.ld_dk = ld_dk; p.ld_dv = ld_dv;
  const size_t staged1_lds = (size_t)(2 * p.L + p.S) * LDT * sizeof(float) + 2 * 16 * LDX * sizeof(float);
  const dim3 mg((unsigned)((p.total + 3) / 4)), mb(256);
  const int nq = (p.L + 15) / 16, nk = (p.S + 15) / 16;
  if (nq > 2 || nk > 2) {                  // 33 .. 128: the generic kernels (Q, dO, K, V of one head + the row statistics in LDS)
    const size_t lds = ((size_t)2 * (p.L + p.S) * HD + 3 * (size_t)p.L) * sizeof(float);
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)attn_bwd_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      npvp_set_error("attn_bwd: could not reserve LDS for the generic kernel"); return NPVP_ERR_LAUNCH;
    }
    NPVP_LAUNCH(attn_bwd_generic_kernel, dim3((unsigned)p.total), dim3(GEN_THREADS), lds, stream, p);
    NPVP_CHECK_LAUNCH();
    return NPVP_OK;
  }
  // up to 16 query rows: operands straight from global memory; 17 .. 32: Q / dO / K staged once in LDS, one score orientation
  // (tools/attn_bench.py at the c2 size: T = 28 478 us, 28 x 2 205 us, T = 18 293 us - the two-orientation kernel that used to
  // take 17 .. 24 rows needed 382 us there once the address arithmetic was out of the way, and is gone)
  if (nq == 1 && nk == 1) NPVP_LAUNCH((attn_bwd_mfma_kernel<1, 1>), mg, mb, 0, stream, p);
  else if (nq == 1) NPVP_LAUNCH((attn_bwd_mfma_kernel<1, 2>), mg, mb, 0, stream, p);
  else if (nk == 1) NPVP_LAUNCH((attn_bwd_staged1_kernel<2, 1>), dim3((unsigned)p.total), dim3(64), staged1_lds, stream, p);
  else NPVP_LAUNCH((attn_bwd_staged1_kernel<2, 2>), dim3((unsigned)p.total), dim3(64), staged1_lds, stream, p);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}
