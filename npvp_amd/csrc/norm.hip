// HBM-bound normalisation kernels of the predictor:
//   * token LayerNorm(C) fwd/bwd, one wavefront per row, registers only
//     (ref/models/VidHRFormer.py:65-66,69,77,175-195; shared final norm :47-48,150-151, + relu_ :159)
//   * per-frame statistics (GroupNorm(1,C) of PosFeatFuser, ref/models/submodules.py:427,446, and
//     LayerNorm((Ch,H,W)) of MlpDWBN, ref/models/VidHRFormer.py:348,361,367)
//   * PosFeatFuser apply  y = xhat*(1+gamma)+beta with the decoder's `+query_evt` folded in
//     (ref/models/submodules.py:449-452, ref/models/VidHRFormer.py:211,236)
//   * frame-LN + per-element affine + GELU + dropout (+ residual, drop-path) apply
// and their backward passes.  Algorithmic bytes: one read + one write of the activation per
// pass (8 B/element); statistics passes re-read a frame that is L2 resident.
#include "common.h"
#include <cstring>
#include <cstdlib>

namespace npvp {

// ------------------------------------------------------------------ token LayerNorm
template <int NV>   // C = NV*256
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, long long rows,
                                                     float eps, int relu, float* __restrict__ amax) {
  constexpr int C = NV * 256;
  __shared__ float ared[4];
  const unsigned int peek = amax_peek_block(amax);
  const int lane = threadIdx.x & 63;
  float4 ww[NV], bb[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { ww[i] = ld4(w + (i * 64 + lane) * 4); bb[i] = ld4(b + (i * 64 + lane) * 4); }
  float am = 0.f;
  // one wave per row, a few rows per wave (grid capped by the launcher): the parameters are loaded once per wave and the
  // amax commit at the end is one per block
  for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i] = ld4(x + row * C + (i * 64 + lane) * 4);
      s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mu = wave_sum(s) * (1.f / C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float a = v[i].x - mu, b2 = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
      q += a * a + b2 * b2 + c * c + d * d;
    }
    const float rs = rsqrtf(wave_sum(q) * (1.f / C) + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c0 = (i * 64 + lane) * 4;
      float4 o;
      o.x = (v[i].x - mu) * rs * ww[i].x + bb[i].x;
      o.y = (v[i].y - mu) * rs * ww[i].y + bb[i].y;
      o.z = (v[i].z - mu) * rs * ww[i].z + bb[i].z;
      o.w = (v[i].w - mu) * rs * ww[i].w + bb[i].w;
      if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      st4(y + row * C + c0, o);
      am = amax4(am, o);
    }
    if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
  }
  amax_slot_commit_block(amax, am, ared, peek);
}

// ---- the decoder's FINAL LayerNorm + ReLU writing the reference's (N,T,C,H,W) tensor directly (SURVEY 2b K9;
// ref/models/VidHRFormer.py:150-159: norm, relu_, permute(0,1,4,2,3)).  Block = one frame of P = 64 token rows: the
// statistics of the 64 rows first (one wave per row, as ln_fwd_kernel), then 128-channel slabs are normalised into an LDS
// tile [128 c][64 p] (row stride 65: both the channel-major writes and the pixel-major reads are conflict free) and leave as
// 256-B rows of the NCHW frame.  One kernel instead of LayerNorm + an LDS-tiled transpose: 8 B/elem instead of 16.
template <int C>
__global__ __launch_bounds__(256) void ln_nchw_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ out,
                                                          float* __restrict__ mean, float* __restrict__ rstd, float eps, int relu) {
  constexpr int NV = C / 256, P = 64;
  __shared__ float tile[128][P + 1];
  __shared__ float smu[P], srs[P];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long f = blockIdx.x;
  const float* xf = x + f * P * C;
  for (int i = 0; i < 16; ++i) {
    const int p = wave * 16 + i;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) { v[k] = ld4(xf + p * C + (k * 64 + lane) * 4); s += v[k].x + v[k].y + v[k].z + v[k].w; }
    const float mu = wave_sum(s) * (1.f / C);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const float a0 = v[k].x - mu, a1 = v[k].y - mu, a2 = v[k].z - mu, a3 = v[k].w - mu;
      q += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
    }
    const float rs = rsqrtf(wave_sum(q) * (1.f / C) + eps);
    if (lane == 0) { smu[p] = mu; srs[p] = rs; mean[f * P + p] = mu; rstd[f * P + p] = rs; }
  }
  __syncthreads();
  float* of = out + f * C * P;
  for (int s0 = 0; s0 < C; s0 += 128) {
    const float w0 = w[s0 + lane], w1 = w[s0 + 64 + lane], b0 = b[s0 + lane], b1 = b[s0 + 64 + lane];
    for (int i = 0; i < 16; ++i) {
      const int p = wave * 16 + i;
      const float mu = smu[p], rs = srs[p];
      float y0 = (xf[p * C + s0 + lane] - mu) * rs * w0 + b0, y1 = (xf[p * C + s0 + 64 + lane] - mu) * rs * w1 + b1;
      if (relu) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); }
      tile[lane][p] = y0; tile[64 + lane][p] = y1;
    }
    __syncthreads();
    for (int j = 0; j < 32; ++j) { const int cl = wave * 32 + j; of[(long long)(s0 + cl) * P + lane] = tile[cl][lane]; }
    __syncthreads();
  }
}

// backward: dx per row; per-block partial dw/db in `partial[blockIdx][2][C]`
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ w, const float* __restrict__ b,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     float* __restrict__ dx, float* __restrict__ partial, long long rows,
                                                     int relu, const float* __restrict__ dres, float* __restrict__ amax) {
  constexpr int C = NV * 256;
  __shared__ float red[4][2 * C];
  __shared__ float ared[4];
  const unsigned int peek = amax_peek_block(amax);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 ww[NV], bb[NV], aw[NV], ab[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c0 = (i * 64 + lane) * 4;
    ww[i] = ld4(w + c0);
    bb[i] = ld4(b + c0);
    aw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float am = 0.f;
  for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    float4 xh[NV], g[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c0 = (i * 64 + lane) * 4;
      const float4 xv = ld4(x + row * C + c0);
      float4 d = ld4(dy + row * C + c0);
      xh[i].x = (xv.x - mu) * rs; xh[i].y = (xv.y - mu) * rs; xh[i].z = (xv.z - mu) * rs; xh[i].w = (xv.w - mu) * rs;
      if (relu) {
        if (xh[i].x * ww[i].x + bb[i].x <= 0.f) d.x = 0.f;
        if (xh[i].y * ww[i].y + bb[i].y <= 0.f) d.y = 0.f;
        if (xh[i].z * ww[i].z + bb[i].z <= 0.f) d.z = 0.f;
        if (xh[i].w * ww[i].w + bb[i].w <= 0.f) d.w = 0.f;
      }
      aw[i].x += d.x * xh[i].x; aw[i].y += d.y * xh[i].y; aw[i].z += d.z * xh[i].z; aw[i].w += d.w * xh[i].w;
      ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
      g[i].x = d.x * ww[i].x; g[i].y = d.y * ww[i].y; g[i].z = d.z * ww[i].z; g[i].w = d.w * ww[i].w;
      s1 += g[i].x + g[i].y + g[i].z + g[i].w;
      s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
    }
    s1 = wave_sum(s1) * (1.f / C);
    s2 = wave_sum(s2) * (1.f / C);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c0 = (i * 64 + lane) * 4;
      float4 o;
      o.x = rs * (g[i].x - s1 - xh[i].x * s2);
      o.y = rs * (g[i].y - s1 - xh[i].y * s2);
      o.z = rs * (g[i].z - s1 - xh[i].z * s2);
      o.w = rs * (g[i].w - s1 - xh[i].w * s2);
      if (dres) { const float4 e = ld4(dres + row * C + c0); o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w; }
      st4(dx + row * C + c0, o);
      am = amax4(am, o);
    }
  }
  amax_slot_commit_block(amax, am, ared, peek);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c0 = (i * 64 + lane) * 4;
    st4(&red[wave][c0], aw[i]);
    st4(&red[wave][C + c0], ab[i]);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * C; c += 256)
    partial[(long long)blockIdx.x * 2 * C + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

// out[c] = sum_b in[b*stride + c]   (second stage of the column reductions).  A block owns 64 columns and
// splits the nb partial rows over blockDim/64 row lanes (coalesced 256-B segments, LDS tree at the end).
template <int CW>     // columns per block: 64 (wide outputs) or 16 (few columns, many partial rows: more blocks, more row lanes)
__global__ __launch_bounds__(1024) void sum_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int nb,
                                                        int stride, int ncols, int accum, float* __restrict__ out_b,
                                                        int split) {
  __shared__ float red[1024];
  const int cx = threadIdx.x % CW, rl = threadIdx.x / CW, nrl = blockDim.x / CW;
  const int c = blockIdx.x * CW + cx;
  float s = 0.f;
  if (c < ncols)
    for (int b = rl; b < nb; b += nrl) s += in[(long long)b * stride + c];
  red[rl * CW + cx] = s;
  __syncthreads();
  if (rl == 0 && c < ncols) {
    float t = 0.f;
    for (int i = 0; i < nrl; ++i) t += red[i * CW + cx];
    float* o = (out_b && c >= split) ? out_b + (c - split) : out + c;
    *o = accum ? *o + t : t;
  }
}

// WIDE sets: few partial rows over very many columns (the frame-LN parameter gradients: 16 - 32 rows x 262 144 columns, 17 - 34 MB
// per layer).  One float per thread and row lane read them at 0.4 - 0.7 TB/s (profiles/r05_streams_c4shard_graph.md: 1.0 ms of a
// replayed 8-clip step for 0.67 GB); here a thread owns FOUR consecutive columns and walks the rows itself: float4 loads, eight in
// flight, no LDS.  Rows are added in order 0 .. nb-1 - the same order wherever this function is called from.
__host__ __device__ constexpr int sum_rows_wide_cols() { return 4096; }          // columns per 1024-thread block
__device__ __forceinline__ void sum_rows_wide(const float* __restrict__ in, float* __restrict__ out, float* __restrict__ out_b, int nb,
                                              int stride, int ncols, int split, int accum, int c) {
  if (c >= ncols) return;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* q = in + c;
#pragma unroll 8
  for (int r = 0; r < nb; ++r) {
    const float4 v = ld4(q + (long long)r * stride);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  float* o = (out_b && c >= split) ? out_b + (c - split) : out + c;       // (split % 4 == 0: the four columns go to one buffer)
  if (accum) { o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w; }
  else { o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; }
}
__global__ __launch_bounds__(1024) void sum_rows_wide_kernel(const float* __restrict__ in, float* __restrict__ out, int nb, int stride,
                                                             int ncols, int accum, float* __restrict__ out_b, int split) {
  sum_rows_wide(in, out, out_b, nb, stride, ncols, split, accum, (blockIdx.x * 1024 + threadIdx.x) * 4);
}

// ------------------------------------------------------------------ per-frame statistics
// One 512-thread block per frame; frame f = n*T + t; u = x[f] (+ add[n]).  Two-pass
// (mean, then centred second moment): the frame (128 KiB .. 512 KiB) is L2 resident.
__global__ __launch_bounds__(512) void frame_stats_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                          float* __restrict__ mean, float* __restrict__ rstd, int T,
                                                          int per_frame, float eps) {
  // ONE pass over the frame (the first version read it twice: mean, then centred second moment).  The sums are taken
  // about a shift = mean of the frame's first 2048 elements, so S2 - S1^2/n has no cancellation to speak of: the shift is
  // within a fraction of a standard deviation of the mean, and even 10 sigma off would cost 1e-5 relative in the variance.
  __shared__ float red[8];
  const int f = blockIdx.x;
  const float* xp = x + (long long)f * per_frame;
  const float* ap = add ? add + (long long)(f / T) * per_frame : nullptr;
  const int e0 = threadIdx.x * 4;
  float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e0 < per_frame) {
    v0 = ld4(xp + e0);
    if (ap) { const float4 a = ld4(ap + e0); v0.x += a.x; v0.y += a.y; v0.z += a.z; v0.w += a.w; }
  }
  const int n0 = per_frame < 2048 ? per_frame : 2048;
  const float shift = block_sum<8>(v0.x + v0.y + v0.z + v0.w, red) / n0;
  float s1 = 0.f, s2 = 0.f;
  for (int e = e0; e < per_frame; e += 512 * 4) {
    float4 v = v0;
    if (e != e0) {
      v = ld4(xp + e);
      if (ap) { const float4 a = ld4(ap + e); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
    }
    const float a0 = v.x - shift, a1 = v.y - shift, a2 = v.z - shift, a3 = v.w - shift;
    s1 += (a0 + a1) + (a2 + a3);
    s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
  }
  const float m1 = block_sum<8>(s1, red) / per_frame;
  const float m2 = block_sum<8>(s2, red) / per_frame;
  if (threadIdx.x == 0) { mean[f] = shift + m1; rstd[f] = rsqrtf(fmaxf(m2 - m1 * m1, 0.f) + eps); }
}

// ------------------------------------------------------------------ PosFeatFuser apply
// y[f,e] = (x[f,e] + add[n,e] - mean[f]) * rstd[f] * (1 + gamma[t,e]) + beta[t,e]
__global__ void posfuse_apply_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                     const float* __restrict__ beta, const float* __restrict__ gamma,
                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                     float* __restrict__ y, int T, int per_frame, long long total4, float* __restrict__ amax) {
  const int pf4 = per_frame / 4;
  __shared__ float ared[16];
  const unsigned int peek = amax_peek_block(amax);
  float am = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const long long f = i / pf4;
    const int e = (int)(i - f * pf4) * 4;
    const int n = (int)(f / T), t = (int)(f - (long long)n * T);
    float4 v = ld4(x + f * per_frame + e);
    if (add) { const float4 a = ld4(add + (long long)n * per_frame + e); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
    const float mu = mean[f], rs = rstd[f];
    const float4 bt = ld4(beta + (long long)t * per_frame + e);
    float4 o;
    o.x = (v.x - mu) * rs; o.y = (v.y - mu) * rs; o.z = (v.z - mu) * rs; o.w = (v.w - mu) * rs;
    if (gamma) {
      const float4 g = ld4(gamma + (long long)t * per_frame + e);
      o.x *= 1.f + g.x; o.y *= 1.f + g.y; o.z *= 1.f + g.z; o.w *= 1.f + g.w;
    }
    o.x += bt.x; o.y += bt.y; o.z += bt.z; o.w += bt.w;
    st4(y + f * per_frame + e, o);
    am = amax4(am, o);
  }
  amax_slot_commit_block(amax, am, ared, peek);
}

// The positional fuse forward in ONE pass for frames that fit a block's registers (8 x 8 x 512 floats = 32 per thread of 1024):
// load the frame (+ add), exact two-pass statistics on the registers, apply, store.  One read of x instead of two and one launch
// instead of two (frame_stats_kernel + posfuse_apply_kernel).
template <int NV4>
__global__ __launch_bounds__(1024) void posfuse_fwd_frame_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                                 const float* __restrict__ beta, const float* __restrict__ gamma,
                                                                 float* __restrict__ y, float* __restrict__ mean,
                                                                 float* __restrict__ rstd, int T, float eps, float* __restrict__ amax) {
  constexpr int PF = NV4 * 4096;
  __shared__ float red[16];
  __shared__ float ared[16];
  const unsigned int peek = amax_peek_block(amax);
  const int f = blockIdx.x, n = f / T, t = f - n * T;
  const float* xf = x + (long long)f * PF;
  const float* af = add ? add + (long long)n * PF : nullptr;
  float4 v[NV4];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NV4; ++k) {
    const int e = (k * 1024 + threadIdx.x) * 4;
    v[k] = ld4(xf + e);
    if (af) { const float4 a = ld4(af + e); v[k].x += a.x; v[k].y += a.y; v[k].z += a.z; v[k].w += a.w; }
    s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  }
  const float mu = block_sum<16>(s, red) * (1.f / PF);
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NV4; ++k) {
    const float a0 = v[k].x - mu, a1 = v[k].y - mu, a2 = v[k].z - mu, a3 = v[k].w - mu;
    q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
  }
  const float rs = rsqrtf(block_sum<16>(q, red) * (1.f / PF) + eps);
  if (threadIdx.x == 0) { mean[f] = mu; rstd[f] = rs; }
  float am = 0.f;
  const float* bt = beta + (long long)t * PF;
  const float* gt = gamma ? gamma + (long long)t * PF : nullptr;
  float* yf = y + (long long)f * PF;
#pragma unroll
  for (int k = 0; k < NV4; ++k) {
    const int e = (k * 1024 + threadIdx.x) * 4;
    float4 o;
    o.x = (v[k].x - mu) * rs; o.y = (v[k].y - mu) * rs; o.z = (v[k].z - mu) * rs; o.w = (v[k].w - mu) * rs;
    if (gt) { const float4 g = ld4(gt + e); o.x *= 1.f + g.x; o.y *= 1.f + g.y; o.z *= 1.f + g.z; o.w *= 1.f + g.w; }
    const float4 b = ld4(bt + e);
    o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
    st4(yf + e, o);
    am = amax4(am, o);
  }
  amax_slot_commit_block(amax, am, ared, peek);
}

// Token LayerNorm AND the positional fuse of its output in one kernel, for 64-token frames of 512 channels (the pre-norm of every
// attention sub-layer: ref/models/VidHRFormer.py:87,94,210,217,235): block = one frame, wave = 4 token rows (a row is 64 lanes x 2
// float4, as in ln_fwd_kernel), the frame stays in registers from the load of x to the store of the fused tensor.  Outputs: x1 =
// LN(x) (the v projection's input, saved for backward), its row statistics, fused = GN(x1 + add) (1 + gamma) + beta, the frame
// statistics, and both amax bounds.  One read of x and one launch instead of LayerNorm + a second pass over x1.
__global__ __launch_bounds__(1024) void ln_posfuse_fwd_frame_kernel(const float* __restrict__ x, const float* __restrict__ lw,
                                                                    const float* __restrict__ lb, float ln_eps, float* __restrict__ y1,
                                                                    float* __restrict__ ln_mean, float* __restrict__ ln_rstd,
                                                                    const float* __restrict__ add, const float* __restrict__ beta,
                                                                    const float* __restrict__ gamma, float* __restrict__ fused,
                                                                    float* __restrict__ pf_mean, float* __restrict__ pf_rstd, int T,
                                                                    float pf_eps, float* __restrict__ amax1, float* __restrict__ amax2) {
  constexpr int C = 512, P = 64, PF = P * C;
  __shared__ float red[16];
  __shared__ float ared1[16];
  __shared__ float ared2[16];
  const unsigned int peek1 = amax_peek_block(amax1), peek2 = amax_peek_block(amax2);
  const int f = blockIdx.x, n = f / T, t = f - n * T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 ww[2], bb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { ww[i] = ld4(lw + (i * 64 + lane) * 4); bb[i] = ld4(lb + (i * 64 + lane) * 4); }
  float4 v[4][2];
  float am1 = 0.f, su = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = wave * 4 + r;
    const long long base = (long long)f * PF + (long long)row * C;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) { v[r][i] = ld4(x + base + (i * 64 + lane) * 4); s += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w); }
    const float mu = wave_sum(s) * (1.f / C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float a = v[r][i].x - mu, b2 = v[r][i].y - mu, c = v[r][i].z - mu, d = v[r][i].w - mu;
      q += a * a + b2 * b2 + c * c + d * d;
    }
    const float rs = rsqrtf(wave_sum(q) * (1.f / C) + ln_eps);
    if (lane == 0) { ln_mean[(long long)f * P + row] = mu; ln_rstd[(long long)f * P + row] = rs; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c0 = (i * 64 + lane) * 4;
      float4 o;
      o.x = (v[r][i].x - mu) * rs * ww[i].x + bb[i].x; o.y = (v[r][i].y - mu) * rs * ww[i].y + bb[i].y;
      o.z = (v[r][i].z - mu) * rs * ww[i].z + bb[i].z; o.w = (v[r][i].w - mu) * rs * ww[i].w + bb[i].w;
      st4(y1 + base + c0, o);
      am1 = amax4(am1, o);
      if (add) { const float4 a = ld4(add + (long long)n * PF + row * C + c0); o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w; }
      v[r][i] = o;                                   // u = LN(x) + add: what the positional fuse normalises
      su += (o.x + o.y) + (o.z + o.w);
    }
  }
  amax_slot_commit_block(amax1, am1, ared1, peek1);
  const float mu = block_sum<16>(su, red) * (1.f / PF);
  float q = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float a0 = v[r][i].x - mu, a1 = v[r][i].y - mu, a2 = v[r][i].z - mu, a3 = v[r][i].w - mu;
      q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
  const float rs = rsqrtf(block_sum<16>(q, red) * (1.f / PF) + pf_eps);
  if (threadIdx.x == 0) { pf_mean[f] = mu; pf_rstd[f] = rs; }
  float am2 = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = (wave * 4 + r) * C + (i * 64 + lane) * 4;
      float4 o;
      o.x = (v[r][i].x - mu) * rs; o.y = (v[r][i].y - mu) * rs; o.z = (v[r][i].z - mu) * rs; o.w = (v[r][i].w - mu) * rs;
      if (gamma) { const float4 g = ld4(gamma + (long long)t * PF + e); o.x *= 1.f + g.x; o.y *= 1.f + g.y; o.z *= 1.f + g.z; o.w *= 1.f + g.w; }
      const float4 b = ld4(beta + (long long)t * PF + e);
      o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
      st4(fused + (long long)f * PF + e, o);
      am2 = amax4(am2, o);
    }
  amax_slot_commit_block(amax2, am2, ared2, peek2);
}

// backward statistics: s1[f] = mean(g), s2[f] = mean(g * uhat), g = dy * (1 + gamma)
__global__ __launch_bounds__(512) void posfuse_bwd_stats_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ add,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, float* __restrict__ s1o,
                                                                float* __restrict__ s2o, int T, int per_frame) {
  __shared__ float red[8];
  const int f = blockIdx.x, n = f / T, t = f - n * T;
  const float mu = mean[f], rs = rstd[f];
  float s1 = 0.f, s2 = 0.f;
  for (int e = threadIdx.x * 4; e < per_frame; e += 512 * 4) {
    float4 v = ld4(x + (long long)f * per_frame + e);
    if (add) { const float4 a = ld4(add + (long long)n * per_frame + e); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
    float4 g = ld4(dy + (long long)f * per_frame + e);
    if (gamma) {
      const float4 gm = ld4(gamma + (long long)t * per_frame + e);
      g.x *= 1.f + gm.x; g.y *= 1.f + gm.y; g.z *= 1.f + gm.z; g.w *= 1.f + gm.w;
    }
    s1 += g.x + g.y + g.z + g.w;
    s2 += g.x * (v.x - mu) * rs + g.y * (v.y - mu) * rs + g.z * (v.z - mu) * rs + g.w * (v.w - mu) * rs;
  }
  s1 = block_sum<8>(s1, red) / per_frame;
  s2 = block_sum<8>(s2, red) / per_frame;
  if (threadIdx.x == 0) { s1o[f] = s1; s2o[f] = s2; }
}

// du = rstd * (g - s1 - uhat*s2); optionally dyxh = dy * uhat (for d gamma)
__global__ void posfuse_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                         const float* __restrict__ add, const float* __restrict__ gamma,
                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                         const float* __restrict__ s1, const float* __restrict__ s2,
                                         float* __restrict__ du, float* __restrict__ dyxh, int T, int per_frame,
                                         long long total4) {
  const int pf4 = per_frame / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const long long f = i / pf4;
    const int e = (int)(i - f * pf4) * 4;
    const int n = (int)(f / T), t = (int)(f - (long long)n * T);
    float4 v = ld4(x + f * per_frame + e);
    if (add) { const float4 a = ld4(add + (long long)n * per_frame + e); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
    const float mu = mean[f], rs = rstd[f], a1 = s1[f], a2 = s2[f];
    float4 uh;
    uh.x = (v.x - mu) * rs; uh.y = (v.y - mu) * rs; uh.z = (v.z - mu) * rs; uh.w = (v.w - mu) * rs;
    const float4 d = ld4(dy + f * per_frame + e);
    float4 g = d;
    if (gamma) {
      const float4 gm = ld4(gamma + (long long)t * per_frame + e);
      g.x *= 1.f + gm.x; g.y *= 1.f + gm.y; g.z *= 1.f + gm.z; g.w *= 1.f + gm.w;
    }
    float4 o;
    o.x = rs * (g.x - a1 - uh.x * a2); o.y = rs * (g.y - a1 - uh.y * a2);
    o.z = rs * (g.z - a1 - uh.z * a2); o.w = rs * (g.w - a1 - uh.w * a2);
    st4(du + f * per_frame + e, o);
    if (dyxh) st4(dyxh + f * per_frame + e, make_float4(d.x * uh.x, d.y * uh.y, d.z * uh.z, d.w * uh.w));
  }
}

// The same apply pass with the batch loop INSIDE the thread: a thread owns one float4 of (t, e) and walks the N samples, so the
// parameter gradients d beta[t,e] = sum_n dy and d gamma[t,e] = sum_n dy * uhat are plain register sums in a fixed order - no
// dy*uhat tensor, no second read of dy, no reduction launches (per positional fuse: 3 passes over an [R, 512] tensor and 2
// launches less).  A block sits inside one t (per_frame / 4 is a multiple of the block size): the frame scalars are uniform.
__global__ __launch_bounds__(256) void posfuse_bwd_apply_nsum_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                     const float* __restrict__ add, const float* __restrict__ gamma,
                                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                     const float* __restrict__ s1, const float* __restrict__ s2,
                                                                     float* __restrict__ du, float* __restrict__ dbeta,
                                                                     float* __restrict__ dgamma, int N, int T, int per_frame, int accum) {
  const int pf4 = per_frame / 4;
  const int t = blockIdx.x / (pf4 / 256), e = ((blockIdx.x % (pf4 / 256)) * 256 + threadIdx.x) * 4;
  float4 gm = make_float4(1.f, 1.f, 1.f, 1.f);
  if (gamma) { const float4 g0 = ld4(gamma + (long long)t * per_frame + e); gm.x += g0.x; gm.y += g0.y; gm.z += g0.z; gm.w += g0.w; }
  float4 sb = make_float4(0.f, 0.f, 0.f, 0.f), sg = sb;
#pragma unroll 4
  for (int n = 0; n < N; ++n) {
    const long long f = (long long)n * T + t;
    float4 v = ld4(x + f * per_frame + e);
    if (add) { const float4 a = ld4(add + (long long)n * per_frame + e); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
    const float mu = mean[f], rs = rstd[f], a1 = s1[f], a2 = s2[f];
    float4 uh;
    uh.x = (v.x - mu) * rs; uh.y = (v.y - mu) * rs; uh.z = (v.z - mu) * rs; uh.w = (v.w - mu) * rs;
    const float4 d = ld4(dy + f * per_frame + e);
    float4 o;
    o.x = rs * (d.x * gm.x - a1 - uh.x * a2); o.y = rs * (d.y * gm.y - a1 - uh.y * a2);
    o.z = rs * (d.z * gm.z - a1 - uh.z * a2); o.w = rs * (d.w * gm.w - a1 - uh.w * a2);
    st4(du + f * per_frame + e, o);
    sb.x += d.x; sb.y += d.y; sb.z += d.z; sb.w += d.w;
    sg.x += d.x * uh.x; sg.y += d.y * uh.y; sg.z += d.z * uh.z; sg.w += d.w * uh.w;
  }
  if (dbeta) {
    if (accum) { const float4 o = ld4(dbeta + (long long)t * per_frame + e); sb.x += o.x; sb.y += o.y; sb.z += o.z; sb.w += o.w; }
    st4(dbeta + (long long)t * per_frame + e, sb);
  }
  if (dgamma) {
    if (accum) { const float4 o = ld4(dgamma + (long long)t * per_frame + e); sg.x += o.x; sg.y += o.y; sg.z += o.z; sg.w += o.w; }
    st4(dgamma + (long long)t * per_frame + e, sg);
  }
}

// ------------------------------------------------------------------ PosFeatFuser, param_free_norm_type = 'instance'
// InstanceNorm2d(affine=False) over the P = H*W pixels of every (frame, channel) (ref/models/submodules.py:427-431: the
// branch no shipped config takes), then xhat (1 + gamma) + beta as the 'layer' form.  Channels-last: thread = one channel of
// one frame, its P <= 64 pixels in registers (consecutive threads = consecutive channels: every load is coalesced); one
// pass forward, one pass backward.  mean / rstd are [frames][C].
constexpr int PFI_MAXP = 64;
__global__ __launch_bounds__(256) void posfuse_inst_fwd_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                               const float* __restrict__ beta, const float* __restrict__ gamma,
                                                               float* __restrict__ y, float* __restrict__ mean,
                                                               float* __restrict__ rstd, int T, int P, int C, float eps,
                                                               float* __restrict__ amax) {
  __shared__ float ared[4];
  const unsigned int peek = amax_peek_block(amax);
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const long long f = blockIdx.y;
  float am = 0.f;
  if (c < C) {
    const int n = (int)(f / T), t = (int)(f - (long long)n * T);
    const float* xp = x + f * P * C + c;
    const float* ap = add ? add + (long long)n * P * C + c : nullptr;
    float u[PFI_MAXP], s = 0.f;
#pragma unroll
    for (int p = 0; p < PFI_MAXP; ++p) if (p < P) { u[p] = xp[(long long)p * C] + (ap ? ap[(long long)p * C] : 0.f); s += u[p]; }
    const float mu = s / P;
    float q = 0.f;
#pragma unroll
    for (int p = 0; p < PFI_MAXP; ++p) if (p < P) { const float d = u[p] - mu; q += d * d; }
    const float rs = rsqrtf(q / P + eps);
    mean[f * C + c] = mu; rstd[f * C + c] = rs;
    const float* bp = beta + (long long)t * P * C + c;
    const float* gp = gamma ? gamma + (long long)t * P * C + c : nullptr;
#pragma unroll
    for (int p = 0; p < PFI_MAXP; ++p) if (p < P) {
      float o = (u[p] - mu) * rs;
      if (gp) o *= 1.f + gp[(long long)p * C];
      o += bp[(long long)p * C];
      y[f * P * C + (long long)p * C + c] = o;
      am = fmaxf(am, fabsf(o));
    }
  }
  amax_slot_commit_block(amax, am, ared, peek);
}

// du = rstd (g - mean_p g - uhat mean_p (g uhat)), g = dy (1 + gamma); dyxh (nullable) = dy * uhat
__global__ __launch_bounds__(256) void posfuse_inst_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ add, const float* __restrict__ gamma,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               float* __restrict__ du, float* __restrict__ dyxh, int T, int P,
                                                               int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const long long f = blockIdx.y;
  if (c >= C) return;
  const int n = (int)(f / T), t = (int)(f - (long long)n * T);
  const long long base = f * P * C + c;
  const float* ap = add ? add + (long long)n * P * C + c : nullptr;
  const float* gp = gamma ? gamma + (long long)t * P * C + c : nullptr;
  const float mu = mean[f * C + c], rs = rstd[f * C + c];
  float g[PFI_MAXP], uh[PFI_MAXP], s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int p = 0; p < PFI_MAXP; ++p) if (p < P) {
    const float d = dy[base + (long long)p * C];
    uh[p] = (x[base + (long long)p * C] + (ap ? ap[(long long)p * C] : 0.f) - mu) * rs;
    if (dyxh) dyxh[base + (long long)p * C] = d * uh[p];
    g[p] = gp ? d * (1.f + gp[(long long)p * C]) : d;
    s1 += g[p]; s2 += g[p] * uh[p];
  }
  s1 /= P; s2 /= P;
#pragma unroll
  for (int p = 0; p < PFI_MAXP; ++p) if (p < P) du[base + (long long)p * C] = rs * (g[p] - s1 - uh[p] * s2);
}

// ------------------------------------------------------------------ frame-LN + affine + GELU (+dropout, residual, drop-path)
// out = res + dp[n] * drop( gelu( (h-mean)*rstd*w[e] + b[e] ) )
struct FlnParams {
  const float* h; const float* mean; const float* rstd; const float* w; const float* b;
  const float* res;                 // nullable residual, same shape as h
  const unsigned long long* seed;
  unsigned int drop_thresh; float drop_inv_keep; unsigned int salt;       // elementwise dropout
  unsigned int dp_thresh; float dp_inv_keep; unsigned int dp_salt; int frames_per_sample;   // per-sample drop-path
  int per_frame; long long total4;
};

// keep-scale of an element = (its frame's DropPath decision) x (its own dropout decision).  The frame's factor is taken ONCE per
// frame (fln_frame_scale) and handed to the per-element function: evaluated per element it was a scalar division and a hash
// per element - in the forward-with-partials kernel more scalar instructions than the kernel had vector ones.
__device__ __forceinline__ float fln_frame_scale(const FlnParams& p, unsigned long long seed, long long f) {
  if (!p.dp_thresh) return 1.f;
  return drop_scale(seed, p.dp_salt, (unsigned long long)((unsigned int)f / (unsigned int)p.frames_per_sample), p.dp_thresh, p.dp_inv_keep);
}
__device__ __forceinline__ float fln_scale(const FlnParams& p, unsigned long long seed, float frame_scale, long long gidx) {
  if (!p.drop_thresh) return frame_scale;
  return drop_scale(seed, p.salt, (unsigned long long)gidx, p.drop_thresh, p.drop_inv_keep) * frame_scale;
}

__global__ void frameln_act_fwd_kernel(FlnParams p, float* __restrict__ out, float* __restrict__ amax) {
  const int pf4 = p.per_frame / 4;
  __shared__ float ared[16];
  const unsigned int peek = amax_peek_block(amax);
  float am = 0.f;
  const unsigned long long seed = (p.seed && (p.drop_thresh || p.dp_thresh)) ? *p.seed : 0ull;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.total4; i += (long long)gridDim.x * blockDim.x) {
    const long long f = i / pf4;
    const int e = (int)(i - f * pf4) * 4;
    const float mu = p.mean[f], rs = p.rstd[f];
    const float fsc = fln_frame_scale(p, seed, f);
    const float4 v = ld4(p.h + f * p.per_frame + e), ww = ld4(p.w + e), bb = ld4(p.b + e);
    const long long g0 = f * p.per_frame + e;
    float4 o;
    o.x = gelu_f((v.x - mu) * rs * ww.x + bb.x) * fln_scale(p, seed, fsc, g0 + 0);
    o.y = gelu_f((v.y - mu) * rs * ww.y + bb.y) * fln_scale(p, seed, fsc, g0 + 1);
    o.z = gelu_f((v.z - mu) * rs * ww.z + bb.z) * fln_scale(p, seed, fsc, g0 + 2);
    o.w = gelu_f((v.w - mu) * rs * ww.w + bb.w) * fln_scale(p, seed, fsc, g0 + 3);
    if (p.res) { const float4 r = ld4(p.res + g0); o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
    st4(out + g0, o);
    am = amax4(am, o);
  }
  amax_slot_commit_block(amax, am, ared, peek);
}

// The same with the frame statistics merged in the kernel from the producer's partials (no finalize launch): a block handles
// 4096 consecutive elements of ONE frame (per_frame % 4096 == 0), merges the frame's J partial pairs, and the frame's first
// block leaves mean / rstd for backward.
__global__ __launch_bounds__(256) void frameln_act_fwd_parts_kernel(FlnParams p, const float* __restrict__ part, int J, float nb,
                                                                    float eps, float* __restrict__ mean_out,
                                                                    float* __restrict__ rstd_out, float* __restrict__ out,
                                                                    float* __restrict__ amax) {
  __shared__ float ared[16];
  const unsigned int peek = amax_peek_block(amax);
  const int bpf = p.per_frame / 4096;
  const long long f = blockIdx.x / bpf;
  const int e0 = (blockIdx.x - (int)f * bpf) * 4096;
  const unsigned long long seed = (p.seed && (p.drop_thresh || p.dp_thresh)) ? *p.seed : 0ull;
  float mu, rs;
  frame_stats_merge(part, f, J, nb, eps, mu, rs);
  if (e0 == 0 && threadIdx.x == 0) { mean_out[f] = mu; rstd_out[f] = rs; }
  const float fsc = fln_frame_scale(p, seed, f);
  float am = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = e0 + (k * 256 + threadIdx.x) * 4;
    const long long g0 = f * p.per_frame + e;
    const float4 v = ld4(p.h + g0), ww = ld4(p.w + e), bb = ld4(p.b + e);
    float4 o;
    o.x = gelu_f((v.x - mu) * rs * ww.x + bb.x) * fln_scale(p, seed, fsc, g0 + 0);
    o.y = gelu_f((v.y - mu) * rs * ww.y + bb.y) * fln_scale(p, seed, fsc, g0 + 1);
    o.z = gelu_f((v.z - mu) * rs * ww.z + bb.z) * fln_scale(p, seed, fsc, g0 + 2);
    o.w = gelu_f((v.w - mu) * rs * ww.w + bb.w) * fln_scale(p, seed, fsc, g0 + 3);
    if (p.res) { const float4 r = ld4(p.res + g0); o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
    st4(out + g0, o);
    am = amax4(am, o);
  }
  amax_slot_commit_block(amax, am, ared, peek);
}

// dy_ln = dout * scale * gelu'(y);  g = dy_ln * w;  s1 = mean(g), s2 = mean(g*hhat)
constexpr int FLN_PARTS = 4;     // blocks per frame in the backward statistics pass
__global__ __launch_bounds__(512) void frameln_act_bwd_stats_kernel(FlnParams p, const float* __restrict__ dout,
                                                                    float* __restrict__ psum) {
  // grid (frames, FLN_PARTS): one block per frame left 64 of the 256 CUs with two 1-MB streams and the rest with one
  // (320 frames at c1) at 8 waves per CU; a quarter frame per block balances the device and lets the GELU' / dropout-hash
  // VALU work overlap the loads.  psum[f][part] = (sum g, sum g*hhat), summed in fixed order by the consumer.
  __shared__ float red[8];
  const long long f = blockIdx.x;
  const unsigned long long seed = (p.seed && (p.drop_thresh || p.dp_thresh)) ? *p.seed : 0ull;
  const float mu = p.mean[f], rs = p.rstd[f];
  const float fsc = fln_frame_scale(p, seed, f);
  float s1 = 0.f, s2 = 0.f;
  const int span = p.per_frame / FLN_PARTS, e0 = blockIdx.y * span, e1 = blockIdx.y == FLN_PARTS - 1 ? p.per_frame : e0 + span;
  for (int e = e0 + threadIdx.x * 4; e < e1; e += 512 * 4) {
    const long long g0 = f * p.per_frame + e;
    const float4 v = ld4(p.h + g0), ww = ld4(p.w + e), bb = ld4(p.b + e), d = ld4(dout + g0);
    const float hx = (v.x - mu) * rs, hy = (v.y - mu) * rs, hz = (v.z - mu) * rs, hw = (v.w - mu) * rs;
    const float gx = d.x * fln_scale(p, seed, fsc, g0 + 0) * gelu_grad_f(hx * ww.x + bb.x) * ww.x;
    const float gy = d.y * fln_scale(p, seed, fsc, g0 + 1) * gelu_grad_f(hy * ww.y + bb.y) * ww.y;
    const float gz = d.z * fln_scale(p, seed, fsc, g0 + 2) * gelu_grad_f(hz * ww.z + bb.z) * ww.z;
    const float gw = d.w * fln_scale(p, seed, fsc, g0 + 3) * gelu_grad_f(hw * ww.w + bb.w) * ww.w;
    s1 += gx + gy + gz + gw;
    s2 += gx * hx + gy * hy + gz * hz + gw * hw;
  }
  s1 = block_sum<8>(s1, red);
  s2 = block_sum<8>(s2, red);
  if (threadIdx.x == 0) { psum[(f * FLN_PARTS + blockIdx.y) * 2] = s1; psum[(f * FLN_PARTS + blockIdx.y) * 2 + 1] = s2; }
}

// One pass for BOTH the input gradient and the parameter gradients: thread = 4 consecutive elements e of the frame,
// loop over the frames of a chunk:  dh[f][e] = rstd (g - s1[f] - hhat s2[f]),  dw[e] += dy_ln*hhat,  db[e] += dy_ln.
// grid.x covers e (float4), grid.y = frame chunks whose partial sums go to part[chunk][2][per_frame]
// (summed by sum_rows_kernel).  Saves the separate apply pass of the first version (a full re-read of dout and h).
__global__ void frameln_act_bwd_fused_kernel(FlnParams p, const float* __restrict__ dout, const float* __restrict__ psum,
                                             float* __restrict__ dh, float* __restrict__ part, int frames,
                                             int frames_per_chunk, int nparts, float* __restrict__ amax) {
  const int e = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  __shared__ float ared[16];
  const unsigned int peek = amax_peek_block(amax);
  float am = 0.f;
  if (e >= p.per_frame) { amax_slot_commit_block(amax, am, ared, peek); return; }      // (every thread joins the block's commit)
  const unsigned long long seed = (p.seed && (p.drop_thresh || p.dp_thresh)) ? *p.seed : 0ull;
  const float4 ww = ld4(p.w + e), bb = ld4(p.b + e);
  float4 aw = make_float4(0.f, 0.f, 0.f, 0.f), ab = make_float4(0.f, 0.f, 0.f, 0.f);
  const int f0 = blockIdx.y * frames_per_chunk;
  const int f1 = min(frames, f0 + frames_per_chunk);
  for (long long f = f0; f < f1; ++f) {
    const long long g0 = f * p.per_frame + e;
    const float mu = p.mean[f], rs = p.rstd[f];
    const float fsc = fln_frame_scale(p, seed, f);
    float a1 = 0.f, a2 = 0.f;              // s1 = mean(g), s2 = mean(g*hhat) from the FLN_PARTS partial sums, fixed order
    for (int j = 0; j < nparts; ++j) { a1 += psum[(f * nparts + j) * 2]; a2 += psum[(f * nparts + j) * 2 + 1]; }
    a1 /= p.per_frame; a2 /= p.per_frame;
    const float4 v = ld4(p.h + g0), d = ld4(dout + g0);
    const float hx = (v.x - mu) * rs, hy = (v.y - mu) * rs, hz = (v.z - mu) * rs, hw = (v.w - mu) * rs;
    const float dx_ = d.x * fln_scale(p, seed, fsc, g0 + 0) * gelu_grad_f(hx * ww.x + bb.x);
    const float dy_ = d.y * fln_scale(p, seed, fsc, g0 + 1) * gelu_grad_f(hy * ww.y + bb.y);
    const float dz_ = d.z * fln_scale(p, seed, fsc, g0 + 2) * gelu_grad_f(hz * ww.z + bb.z);
    const float dw_ = d.w * fln_scale(p, seed, fsc, g0 + 3) * gelu_grad_f(hw * ww.w + bb.w);
    aw.x += dx_ * hx; aw.y += dy_ * hy; aw.z += dz_ * hz; aw.w += dw_ * hw;
    ab.x += dx_; ab.y += dy_; ab.z += dz_; ab.w += dw_;
    float4 o;
    o.x = rs * (dx_ * ww.x - a1 - hx * a2); o.y = rs * (dy_ * ww.y - a1 - hy * a2);
    o.z = rs * (dz_ * ww.z - a1 - hz * a2); o.w = rs * (dw_ * ww.w - a1 - hw * a2);
    st4(dh + g0, o);
    am = amax4(am, o);
  }
  amax_slot_commit_block(amax, am, ared, peek);
  float* o = part + (long long)blockIdx.y * 2 * p.per_frame;
  st4(o + e, aw);
  st4(o + p.per_frame + e, ab);
}

// Statistics AND parameter gradients of the frame LayerNorm's backward in one pass, WITHOUT the input gradient: for a consumer
// that evaluates dh = rstd (g - s1 - hhat s2) itself where it needs it (the fused MlpDWBN middle's backward does, for norm2: dh2
// is never written).  Same thread map as frameln_act_bwd_fused_kernel (thread = 4 elements of the frame, loop over the frames
// of a chunk: dw / db in registers); the block's share of the frame sums leaves as psum[f][blockIdx.x] = (sum g, sum g*hhat),
// summed in fixed order by the consumer.  per_frame must be a multiple of 1024 (every thread of every block is live).
__global__ __launch_bounds__(256) void frameln_act_bwd_pgrad_kernel(FlnParams p, const float* __restrict__ dout,
                                                                    float* __restrict__ psum, float* __restrict__ part, int frames,
                                                                    int frames_per_chunk) {
  __shared__ float red[4];
  const int e = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int nblk = gridDim.x;
  const unsigned long long seed = (p.seed && (p.drop_thresh || p.dp_thresh)) ? *p.seed : 0ull;
  const float4 ww = ld4(p.w + e), bb = ld4(p.b + e);
  float4 aw = make_float4(0.f, 0.f, 0.f, 0.f), ab = make_float4(0.f, 0.f, 0.f, 0.f);
  const int f0 = blockIdx.y * frames_per_chunk;
  const int f1 = min(frames, f0 + frames_per_chunk);
  for (long long f = f0; f < f1; ++f) {
    const long long g0 = f * p.per_frame + e;
    const float mu = p.mean[f], rs = p.rstd[f];
    const float fsc = fln_frame_scale(p, seed, f);
    const float4 v = ld4(p.h + g0), d = ld4(dout + g0);
    const float hx = (v.x - mu) * rs, hy = (v.y - mu) * rs, hz = (v.z - mu) * rs, hw = (v.w - mu) * rs;
    const float dx_ = d.x * fln_scale(p, seed, fsc, g0 + 0) * gelu_grad_f(hx * ww.x + bb.x);
    const float dy_ = d.y * fln_scale(p, seed, fsc, g0 + 1) * gelu_grad_f(hy * ww.y + bb.y);
    const float dz_ = d.z * fln_scale(p, seed, fsc, g0 + 2) * gelu_grad_f(hz * ww.z + bb.z);
    const float dw_ = d.w * fln_scale(p, seed, fsc, g0 + 3) * gelu_grad_f(hw * ww.w + bb.w);
    aw.x += dx_ * hx; aw.y += dy_ * hy; aw.z += dz_ * hz; aw.w += dw_ * hw;
    ab.x += dx_; ab.y += dy_; ab.z += dz_; ab.w += dw_;
    const float gx = dx_ * ww.x, gy = dy_ * ww.y, gz = dz_ * ww.z, gw = dw_ * ww.w;
    const float s1 = block_sum<4>((gx + gy) + (gz + gw), red);
    const float s2 = block_sum<4>((gx * hx + gy * hy) + (gz * hz + gw * hw), red);
    if (threadIdx.x == 0) { psum[(f * nblk + blockIdx.x) * 2] = s1; psum[(f * nblk + blockIdx.x) * 2 + 1] = s2; }
  }
  float* o = part + (long long)blockIdx.y * 2 * p.per_frame;
  st4(o + e, aw);
  st4(o + p.per_frame + e, ab);
}

int launch_sum_rows(const float* in, float* out, int nb, int stride, int ncols, hipStream_t stream, int accum, float* out_b,
                    int split) {
  // few partial rows (frame-LN params: 32 x 262144) -> 4 row lanes of 64 columns; many partial rows over few columns
  // (bias / LayerNorm / split-K column sums: 512 x 1024) -> 16-column blocks with 64 row lanes: 4x the blocks
  if (nb < 64 && ncols >= 4096 && ncols % 4 == 0 && stride % 4 == 0 && (!out_b || split % 4 == 0) && ((uintptr_t)in & 15) == 0)
    NPVP_LAUNCH(sum_rows_wide_kernel, dim3((ncols + sum_rows_wide_cols() - 1) / sum_rows_wide_cols()), dim3(1024), 0, stream, in, out,
                nb, stride, ncols, accum, out_b, out_b ? split : 0);
  else if (nb >= 64 && ncols <= 8192)
    NPVP_LAUNCH(sum_rows_kernel<16>, dim3((ncols + 15) / 16), dim3(1024), 0, stream, in, out, nb, stride, ncols, accum,
                       out_b, split);
  else
    NPVP_LAUNCH(sum_rows_kernel<64>, dim3((ncols + 63) / 64), dim3(nb >= 64 ? 1024 : 256), 0, stream, in, out, nb, stride,
                       ncols, accum, out_b, split);
  return hipGetLastError() == hipSuccess ? NPVP_OK : NPVP_ERR_LAUNCH;
}

// ---- many column reductions in ONE launch.  A backward pass leaves ~150 sets of partial rows behind (LayerNorm / frame-LN /
// depthwise parameter gradients: [nb][ncols] each, to be summed over nb into slices of the flat gradient buffer).  One launch per
// set was 150 launches of ~10 us on the gradient stream of an 8-clip step (and 150 graph nodes with their dispatch gaps when the step
// is replayed from a graph); the sets have no consumer before the optimiser, so they are queued on the host and summed by a
// handful of launches when the backward pass ends.  The jobs travel in the kernel's ARGUMENT block (no device table to keep alive,
// nothing for a graph replay to re-upload).  Per job the same scheme as sum_rows_kernel: a block owns `cw` columns (16 or 64) and
// splits the nb partial rows over 1024 / cw row lanes; fixed summation order.
struct SumRowsJob {
  const float* in; float* out; float* out_b;   // out_b (nullable): columns >= split go to out_b[c - split]
  int nb, stride, ncols, split;
  int accum;                                   // 1: += into out / out_b
  int mode;                                    // 0 plain; 1: depthwise-conv partials [10][Ch] (ncols = 10 Ch, split = Ch): tap t < 9 of channel c
                                               //    -> out[c*9 + t], the bias row -> out_b[c]   (mid_bwd_reduce_into_kernel's map)
};
constexpr int SRJ_MAX = 40;
struct SumRowsBatch { SumRowsJob j[SRJ_MAX]; int first[SRJ_MAX + 1]; int cw[SRJ_MAX]; int n; };

__global__ __launch_bounds__(1024) void sum_rows_multi_kernel(SumRowsBatch b) {
  __shared__ float red[1024];
  int lo = 0, hi = b.n - 1;                    // the job of this block: last j with first[j] <= blockIdx.x
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (b.first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
  const SumRowsJob& J = b.j[lo];
  const int CW = b.cw[lo], sh = CW == 16 ? 4 : 6;
  if (CW == sum_rows_wide_cols()) {              // (block-uniform; the wide path has no barrier)
    sum_rows_wide(J.in, J.out, J.out_b, J.nb, J.stride, J.ncols, J.out_b ? J.split : 0, J.accum,
                  (((int)blockIdx.x - b.first[lo]) * 1024 + (int)threadIdx.x) * 4);
    return;
  }
  const int cx = threadIdx.x & (CW - 1), rl = threadIdx.x >> sh, nrl = 1024 >> sh;
  const int c = ((int)blockIdx.x - b.first[lo]) * CW + cx;
  float s = 0.f;
  if (c < J.ncols)
    for (int r = rl; r < J.nb; r += nrl) s += J.in[(long long)r * J.stride + c];
  red[rl * CW + cx] = s;
  __syncthreads();
  if (rl == 0 && c < J.ncols) {
    float t = 0.f;
    for (int i = 0; i < nrl; ++i) t += red[i * CW + cx];
    float* o;
    if (J.mode == 1) { const int tap = c / J.split, ch = c - tap * J.split; o = tap < 9 ? J.out + ch * 9 + tap : J.out_b + ch; }
    else o = (J.out_b && c >= J.split) ? J.out_b + (c - J.split) : J.out + c;
    *o = J.accum ? *o + t : t;
  }
}

static inline int ew_blocks(long long total, int threads) {
  long long b = (total + threads - 1) / threads;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

static void fill_fln(FlnParams& p, const float* h, const float* mean, const float* rstd, const float* w, const float* b,
                     const float* res, int frames, int per_frame, float drop_p, unsigned int salt, float dp_p,
                     unsigned int dp_salt, int frames_per_sample, const unsigned long long* seed) {
  p.h = h; p.mean = mean; p.rstd = rstd; p.w = w; p.b = b; p.res = res; p.seed = seed;
  p.drop_thresh = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
  p.drop_inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  p.salt = salt;
  p.dp_thresh = dp_p > 0.f ? drop_threshold(dp_p) : 0u;
  p.dp_inv_keep = dp_p > 0.f ? 1.f / (1.f - dp_p) : 1.f;
  p.dp_salt = dp_salt; p.frames_per_sample = frames_per_sample > 0 ? frames_per_sample : 1;
  p.per_frame = per_frame; p.total4 = (long long)frames * per_frame / 4;
}

}  // namespace npvp

using namespace npvp;

extern "C" int npvp_layernorm_fwd(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd,
                                  long long rows, int C, float eps, int relu, float* amax, hipStream_t stream) {
  NPVP_CHECK_ARG(rows > 0, "layernorm: no rows");
  NPVP_CHECK_ARG(C % 256 == 0 && C >= 256 && C <= 1024, "layernorm: C must be 256, 512, 768 or 1024");
  // <= 4096 blocks (16 waves per CU x 4 rounds): short waves would pay the per-block amax commit and the parameter loads per row
  const long long nb = (rows + 3) / 4;
  dim3 grid((unsigned)(nb > 4096 ? 4096 : nb)), block(256);
  switch (C / 256) {
    case 1: NPVP_LAUNCH(ln_fwd_kernel<1>, grid, block, 0, stream, x, w, b, y, mean, rstd, rows, eps, relu, amax); break;
    case 2: NPVP_LAUNCH(ln_fwd_kernel<2>, grid, block, 0, stream, x, w, b, y, mean, rstd, rows, eps, relu, amax); break;
    case 3: NPVP_LAUNCH(ln_fwd_kernel<3>, grid, block, 0, stream, x, w, b, y, mean, rstd, rows, eps, relu, amax); break;
    default: NPVP_LAUNCH(ln_fwd_kernel<4>, grid, block, 0, stream, x, w, b, y, mean, rstd, rows, eps, relu, amax); break;
  }
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

static int ln_bwd_blocks(long long rows) {
  long long b = (rows + 3) / 4;
  return (int)(b > 512 ? 512 : b);      // 2048 / 4096 blocks: no change of the c2 step (331.0 / 331.7 / 331.3 ms)
}

extern "C" long long npvp_layernorm_bwd_workspace_bytes(long long rows, int C) {
  return (long long)ln_bwd_blocks(rows) * 2 * C * 4;
}

extern "C" int npvp_layernorm_bwd(const float* dy, const float* x, const float* w, const float* b, const float* mean,
                                  const float* rstd, float* dx, float* dw, float* db, long long rows, int C, int relu,
                                  const float* dres, int accumulate, float* amax, void* workspace, long long ws_bytes,
                                  hipStream_t stream) {
  NPVP_CHECK_ARG(rows > 0, "layernorm_bwd: no rows");
  NPVP_CHECK_ARG(C % 256 == 0 && C >= 256 && C <= 1024, "layernorm_bwd: C must be 256, 512, 768 or 1024");
  const int nb = ln_bwd_blocks(rows);
  NPVP_CHECK_ARG(workspace && ws_bytes >= (long long)nb * 2 * C * 4, "layernorm_bwd: workspace too small");
  float* part = (float*)workspace;
  dim3 grid(nb), block(256);
  switch (C / 256) {
    case 1: NPVP_LAUNCH(ln_bwd_kernel<1>, grid, block, 0, stream, dy, x, w, b, mean, rstd, dx, part, rows, relu, dres, amax); break;
    case 2: NPVP_LAUNCH(ln_bwd_kernel<2>, grid, block, 0, stream, dy, x, w, b, mean, rstd, dx, part, rows, relu, dres, amax); break;
    case 3: NPVP_LAUNCH(ln_bwd_kernel<3>, grid, block, 0, stream, dy, x, w, b, mean, rstd, dx, part, rows, relu, dres, amax); break;
    default: NPVP_LAUNCH(ln_bwd_kernel<4>, grid, block, 0, stream, dy, x, w, b, mean, rstd, dx, part, rows, relu, dres, amax); break;
  }
  NPVP_CHECK_LAUNCH();
  // partial rows are [dw(C) | db(C)]
  if (accumulate == 2) return NPVP_OK;      // the caller reduces the partials itself (npvp_layernorm_bwd_reduce)
  if (launch_sum_rows((const float*)part, dw, nb, 2 * C, 2 * C, stream, accumulate, db, C)) {
    npvp_set_error("layernorm_bwd: reduce launch failed");
    return NPVP_ERR_LAUNCH;
  }
  return NPVP_OK;
}

// second stage of npvp_layernorm_bwd(accumulate = 2) on a stream of the caller's choice: dw, db (+)= column sums of the
// partial rows left in `workspace`
extern "C" int npvp_layernorm_bwd_reduce(const void* workspace, float* dw, float* db, long long rows, int C, int accumulate,
                                         hipStream_t stream) {
  NPVP_CHECK_ARG(workspace && dw && db && rows > 0, "layernorm_bwd_reduce: bad arguments");
  if (launch_sum_rows((const float*)workspace, dw, ln_bwd_blocks(rows), 2 * C, 2 * C, stream, accumulate ? 1 : 0, db, C)) {
    npvp_set_error("layernorm_bwd_reduce: launch failed");
    return NPVP_ERR_LAUNCH;
  }
  return NPVP_OK;
}

// The same reduction as a JOB for npvp_sum_rows_multi (48 bytes at `job`, see include/npvp_hip.h): nothing is launched.
static void fill_job(void* job, const float* in, float* out, float* out_b, int nb, int stride, int ncols, int split, int accum, int mode) {
  SumRowsJob j;
  j.in = in; j.out = out; j.out_b = out_b; j.nb = nb; j.stride = stride; j.ncols = ncols; j.split = split; j.accum = accum; j.mode = mode;
  memcpy(job, &j, sizeof(j));
}
static_assert(sizeof(SumRowsJob) == 48, "a job is 48 bytes (npvp_amd/ops.py ReduceQueue packs them back to back)");

extern "C" int npvp_layernorm_bwd_reduce_job(const void* workspace, float* dw, float* db, long long rows, int C, int accumulate, void* job) {
  NPVP_CHECK_ARG(workspace && dw && db && rows > 0 && job, "layernorm_bwd_reduce_job: bad arguments");
  fill_job(job, (const float*)workspace, dw, db, ln_bwd_blocks(rows), 2 * C, 2 * C, C, accumulate ? 1 : 0, 0);
  return NPVP_OK;
}

// `jobs` = n SumRowsJob records in HOST memory (filled by the *_reduce_job entry points); ceil(n / 40) launches
extern "C" int npvp_sum_rows_multi(const void* jobs, int n, hipStream_t stream) {
  NPVP_CHECK_ARG(jobs && n > 0, "sum_rows_multi: no jobs");
  const SumRowsJob* J = (const SumRowsJob*)jobs;
  for (int at = 0; at < n; at += SRJ_MAX) {
    SumRowsBatch b;
    b.n = n - at < SRJ_MAX ? n - at : SRJ_MAX;
    int blocks = 0;
    for (int i = 0; i < b.n; ++i) {
      b.j[i] = J[at + i];
      NPVP_CHECK_ARG(b.j[i].in && b.j[i].out && b.j[i].nb > 0 && b.j[i].ncols > 0, "sum_rows_multi: bad job");
      const SumRowsJob& q = b.j[i];                                            // (launch_sum_rows' choice)
      const bool wide = q.mode == 0 && q.nb < 64 && q.ncols >= 4096 && q.ncols % 4 == 0 && q.stride % 4 == 0
                        && (!q.out_b || q.split % 4 == 0) && ((uintptr_t)q.in & 15) == 0;
      b.cw[i] = wide ? sum_rows_wide_cols() : (q.nb >= 64 && q.ncols <= 8192) ? 16 : 64;
      b.first[i] = blocks;
      blocks += (b.j[i].ncols + b.cw[i] - 1) / b.cw[i];
    }
    b.first[b.n] = blocks;
    NPVP_LAUNCH(sum_rows_multi_kernel, dim3(blocks), dim3(1024), 0, stream, b);
    NPVP_CHECK_LAUNCH();
  }
  return NPVP_OK;
}

// LayerNorm(C) (+ReLU) over the token rows of `frames` frames of 64 pixels, output in the reference's (frames, C, 8, 8) layout
// (K9).  C in {256, 512}.  mean / rstd are per token row, as npvp_layernorm_fwd writes them: the backward is npvp_transpose of dy
// + npvp_layernorm_bwd (a fused backward through the same LDS tile was 3x slower than those two kernels: 771 vs 239 us at c2).
extern "C" int npvp_layernorm_nchw_fwd(const float* x, const float* w, const float* b, float* out, float* mean, float* rstd,
                                       int frames, int P, int C, float eps, int relu, hipStream_t stream) {
  NPVP_CHECK_ARG(x && w && b && out && mean && rstd && frames > 0, "layernorm_nchw_fwd: bad arguments");
  NPVP_CHECK_ARG(P == 64 && (C == 256 || C == 512), "layernorm_nchw_fwd: 64-pixel frames, C = 256 or 512");
  if (C == 512) NPVP_LAUNCH(ln_nchw_fwd_kernel<512>, dim3(frames), dim3(256), 0, stream, x, w, b, out, mean, rstd, eps, relu);
  else NPVP_LAUNCH(ln_nchw_fwd_kernel<256>, dim3(frames), dim3(256), 0, stream, x, w, b, out, mean, rstd, eps, relu);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

extern "C" int npvp_frame_stats(const float* x, const float* add, float* mean, float* rstd, int frames, int T,
                                int per_frame, float eps, hipStream_t stream) {
  NPVP_CHECK_ARG(frames > 0 && T > 0 && frames % T == 0, "frame_stats: frames must be a multiple of T");
  NPVP_CHECK_ARG(per_frame % 4 == 0, "frame_stats: per_frame must be a multiple of 4");
  NPVP_LAUNCH(frame_stats_kernel, dim3(frames), dim3(512), 0, stream, x, add, mean, rstd, T, per_frame, eps);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

extern "C" int npvp_posfuse_fwd(const float* x, const float* add, const float* beta, const float* gamma, float* y,
                                float* mean, float* rstd, int N, int T, int per_frame, float eps, float* amax, hipStream_t stream) {
  NPVP_CHECK_ARG(N > 0 && T > 0 && per_frame % 4 == 0, "posfuse: bad shape");
  const int frames = N * T;
  if (per_frame == 32768) {          // 8 x 8 x 512: the frame lives in the block's registers
    NPVP_LAUNCH((posfuse_fwd_frame_kernel<8>), dim3(frames), dim3(1024), 0, stream, x, add, beta, gamma, y, mean, rstd, T,
                       eps, amax);
    NPVP_CHECK_LAUNCH();
    return NPVP_OK;
  }
  NPVP_LAUNCH(frame_stats_kernel, dim3(frames), dim3(512), 0, stream, x, add, mean, rstd, T, per_frame, eps);
  NPVP_CHECK_LAUNCH();
  const long long total4 = (long long)frames * per_frame / 4;
  NPVP_LAUNCH(posfuse_apply_kernel, dim3(ew_blocks(total4, 256)), dim3(256), 0, stream, x, add, beta, gamma,
                     (const float*)mean, (const float*)rstd, y, T, per_frame, total4, amax);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

// LayerNorm(C = 512) + positional fuse of frames of P = 64 token rows in one kernel (see ln_posfuse_fwd_frame_kernel): x [N*T*P][C];
// y1 = LN(x), ln_mean / ln_rstd [N*T*P]; fused [N*T][P*C], pf_mean / pf_rstd [N*T]; add [N][P*C] or NULL, beta / gamma [T][P*C]
extern "C" int npvp_ln_posfuse_fwd(const float* x, const float* lw, const float* lb, float ln_eps, float* y1, float* ln_mean,
                                   float* ln_rstd, const float* add, const float* beta, const float* gamma, float* fused,
                                   float* pf_mean, float* pf_rstd, int N, int T, int P, int C, float pf_eps, float* y1_amax,
                                   float* fused_amax, hipStream_t stream) {
  NPVP_CHECK_ARG(N > 0 && T > 0 && P == 64 && C == 512, "ln_posfuse_fwd: frames of 64 token rows x 512 channels only");
  NPVP_CHECK_ARG(x && lw && lb && y1 && ln_mean && ln_rstd && beta && fused && pf_mean && pf_rstd, "ln_posfuse_fwd: null argument");
  NPVP_LAUNCH(ln_posfuse_fwd_frame_kernel, dim3(N * T), dim3(1024), 0, stream, x, lw, lb, ln_eps, y1, ln_mean, ln_rstd, add,
                     beta, gamma, fused, pf_mean, pf_rstd, T, pf_eps, y1_amax, fused_amax);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

// 1 when npvp_posfuse_bwd computes d beta / d gamma inside its apply pass (the batch loop in the thread): enough (t, e) work to
// fill the device, or so few samples that the launches saved matter more
extern "C" int npvp_posfuse_bwd_fused(int N, int T, int per_frame) {
  if (per_frame % 1024 != 0) return 0;
  const long long blocks = (long long)T * (per_frame / 1024);
  return (blocks >= 128 || N <= 16) ? 1 : 0;
}

int npvp_reduce_mid_launch(const float* in, float* out, int A, int B, long long Cc, float scale, hipStream_t stream, int accumulate = 0);   // elementwise.hip

// du [N*T, per_frame]; dbeta / dgamma [T, per_frame] (nullable) = sum over the batch of dy / dy*uhat; dyxh (nullable) = scratch
// for dy*uhat, needed only when npvp_posfuse_bwd_fused(N, T, per_frame) == 0 and dgamma is wanted; ws = 2*N*T floats
extern "C" int npvp_posfuse_bwd(const float* dy, const float* x, const float* add, const float* gamma, const float* mean,
                                const float* rstd, float* du, float* dyxh, float* dbeta, float* dgamma, int N, int T,
                                int per_frame, int accumulate, void* workspace, long long ws_bytes, hipStream_t stream) {
  const int frames = N * T;
  NPVP_CHECK_ARG(N > 0 && T > 0 && per_frame % 4 == 0, "posfuse_bwd: bad shape");
  NPVP_CHECK_ARG(workspace && ws_bytes >= (long long)frames * 2 * 4, "posfuse_bwd: workspace too small");
  NPVP_CHECK_ARG(!dgamma || gamma, "posfuse_bwd: dgamma without gamma");
  float* s1 = (float*)workspace; float* s2 = s1 + frames;
  NPVP_LAUNCH(posfuse_bwd_stats_kernel, dim3(frames), dim3(512), 0, stream, dy, x, add, gamma, mean, rstd, s1, s2, T,
                     per_frame);
  NPVP_CHECK_LAUNCH();
  if (npvp_posfuse_bwd_fused(N, T, per_frame)) {
    NPVP_LAUNCH(posfuse_bwd_apply_nsum_kernel, dim3(T * (per_frame / 1024)), dim3(256), 0, stream, dy, x, add, gamma, mean,
                       rstd, (const float*)s1, (const float*)s2, du, dbeta, dgamma, N, T, per_frame, accumulate ? 1 : 0);
    NPVP_CHECK_LAUNCH();
    return NPVP_OK;
  }
  NPVP_CHECK_ARG(!dgamma || dyxh, "posfuse_bwd: this shape needs the dy*uhat scratch (dyxh) for dgamma");
  const long long total4 = (long long)frames * per_frame / 4;
  NPVP_LAUNCH(posfuse_bwd_apply_kernel, dim3(ew_blocks(total4, 256)), dim3(256), 0, stream, dy, x, add, gamma, mean,
                     rstd, (const float*)s1, (const float*)s2, du, dgamma ? dyxh : nullptr, T, per_frame, total4);
  NPVP_CHECK_LAUNCH();
  if (dbeta) { const int rc = npvp_reduce_mid_launch(dy, dbeta, 1, N, (long long)T * per_frame, 1.f, stream, accumulate); if (rc) return rc; }
  if (dgamma) { const int rc = npvp_reduce_mid_launch(dyxh, dgamma, 1, N, (long long)T * per_frame, 1.f, stream, accumulate); if (rc) return rc; }
  return NPVP_OK;
}

// 'instance' form of the positional fuse: x [N*T][P][C], add [N][P][C] or NULL, beta / gamma [T][P][C]; mean / rstd [N*T][C]
extern "C" int npvp_posfuse_instance_fwd(const float* x, const float* add, const float* beta, const float* gamma, float* y,
                                         float* mean, float* rstd, int N, int T, int P, int C, float eps, float* y_amax,
                                         hipStream_t stream) {
  NPVP_CHECK_ARG(N > 0 && T > 0 && P > 0 && P <= PFI_MAXP && C > 0, "posfuse_instance: bad shape (P <= 64)");
  NPVP_LAUNCH(posfuse_inst_fwd_kernel, dim3((C + 255) / 256, N * T), dim3(256), 0, stream, x, add, beta, gamma, y, mean,
                     rstd, T, P, C, eps, y_amax);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

extern "C" int npvp_posfuse_instance_bwd(const float* dy, const float* x, const float* add, const float* gamma, const float* mean,
                                         const float* rstd, float* du, float* dyxh, int N, int T, int P, int C,
                                         hipStream_t stream) {
  NPVP_CHECK_ARG(N > 0 && T > 0 && P > 0 && P <= PFI_MAXP && C > 0, "posfuse_instance_bwd: bad shape (P <= 64)");
  NPVP_LAUNCH(posfuse_inst_bwd_kernel, dim3((C + 255) / 256, N * T), dim3(256), 0, stream, dy, x, add, gamma, mean, rstd,
                     du, dyxh, T, P, C);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

extern "C" int npvp_frameln_act_fwd(const float* h, const float* mean, const float* rstd, const float* w, const float* b,
                                    const float* res, float* out, int frames, int per_frame, float drop_p,
                                    unsigned int salt, float dp_p, unsigned int dp_salt, int frames_per_sample,
                                    const unsigned long long* seed, float* amax, hipStream_t stream) {
  NPVP_CHECK_ARG(frames > 0 && per_frame % 4 == 0, "frameln_act: bad shape");
  NPVP_CHECK_ARG((drop_p == 0.f && dp_p == 0.f) || seed, "frameln_act: dropout needs a device seed");
  FlnParams p;
  fill_fln(p, h, mean, rstd, w, b, res, frames, per_frame, drop_p, salt, dp_p, dp_salt, frames_per_sample, seed);
  NPVP_LAUNCH(frameln_act_fwd_kernel, dim3(ew_blocks(p.total4, 256)), dim3(256), 0, stream, p, out, amax);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

// frame statistics from the producer's partials part [frames][J][2] (mean_j, M2_j over nb values each: the rowstats of
// npvp_gemm_f32 or the part2 of npvp_mlpdw_mid_fwd_parts); mean / rstd [frames] are OUTPUTS (for backward).  per_frame % 4096 == 0.
extern "C" int npvp_frameln_act_fwd_parts(const float* h, const float* part, int J, float nb, float eps, float* mean, float* rstd,
                                          const float* w, const float* b, const float* res, float* out, int frames,
                                          int per_frame, float drop_p, unsigned int salt, float dp_p, unsigned int dp_salt,
                                          int frames_per_sample, const unsigned long long* seed, float* amax, hipStream_t stream) {
  NPVP_CHECK_ARG(frames > 0 && per_frame % 4096 == 0 && part && J > 0 && nb > 0.f && mean && rstd, "frameln_act_fwd_parts: bad arguments");
  NPVP_CHECK_ARG((drop_p == 0.f && dp_p == 0.f) || seed, "frameln_act: dropout needs a device seed");
  NPVP_CHECK_ARG((long long)frames * (per_frame / 4096) < (1ll << 31), "frameln_act_fwd_parts: too many blocks");
  FlnParams p;
  fill_fln(p, h, nullptr, nullptr, w, b, res, frames, per_frame, drop_p, salt, dp_p, dp_salt, frames_per_sample, seed);
  NPVP_LAUNCH(frameln_act_fwd_parts_kernel, dim3((unsigned)((long long)frames * (per_frame / 4096))), dim3(256), 0, stream, p,
                     part, J, nb, eps, mean, rstd, out, amax);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

// frame chunks (grid.y) of the one-pass backward: 32 x 128 = 4096 workgroups.  With 8 (1024 workgroups, each walking 224 frames
// at c2) the kernel was as fast stand-alone (921 vs 923 us for statistics + apply) but lost CU slots to the co-resident
// weight-gradient GEMM: c2 step 342.7 -> 338.3 ms (three A/B pairs on one box; 64 chunks: 338.1).
// frame chunks of the backward kernels' parameter-gradient partials ([chunks][2 * per_frame] floats, summed later).  32 for the
// large workloads (with 8 the kernels lost their CU share to a co-resident weight-gradient GEMM); for fewer
// than 512 frames at most 16 chunks of at least two frames: an 8-clip shard's 160 frames in 32 chunks left partials 40 % the size
// of the tensor itself behind every frame LayerNorm (33 MB per site, written and read again: 1.2 ms of a 33 ms step).
static int fln_chunks(int frames) {
  if (frames >= 512) return 32;
  const int c = frames / 2 < 16 ? frames / 2 : 16;
  return c < 1 ? 1 : c;
}

extern "C" long long npvp_frameln_act_bwd_workspace_bytes(int frames, int per_frame) {
  return ((long long)frames * 2 * FLN_PARTS + (long long)fln_chunks(frames) * 2 * per_frame) * 4;
}

// dh [frames, per_frame]; dw, db [per_frame]
extern "C" int npvp_frameln_act_bwd(const float* dout, const float* h, const float* mean, const float* rstd, const float* w,
                                    const float* b, float* dh, float* dw, float* db, int frames, int per_frame,
                                    float drop_p, unsigned int salt, float dp_p, unsigned int dp_salt,
                                    int frames_per_sample, const unsigned long long* seed, int accumulate, float* amax,
                                    void* workspace, long long ws_bytes, hipStream_t stream) {
  NPVP_CHECK_ARG(frames > 0 && per_frame % 4 == 0, "frameln_act_bwd: bad shape");
  NPVP_CHECK_ARG(workspace && ws_bytes >= npvp_frameln_act_bwd_workspace_bytes(frames, per_frame),
                 "frameln_act_bwd: workspace too small");
  FlnParams p;
  fill_fln(p, h, mean, rstd, w, b, nullptr, frames, per_frame, drop_p, salt, dp_p, dp_salt, frames_per_sample, seed);
  NPVP_CHECK_ARG(per_frame % (4 * FLN_PARTS) == 0, "frameln_act_bwd: per_frame must be a multiple of 16");
  float* psum = (float*)workspace; float* part = psum + (long long)frames * 2 * FLN_PARTS;
  NPVP_LAUNCH(frameln_act_bwd_stats_kernel, dim3(frames, FLN_PARTS), dim3(512), 0, stream, p, dout, psum);
  NPVP_CHECK_LAUNCH();
  const int chunks = fln_chunks(frames), fpc = (frames + chunks - 1) / chunks;
  const int nchunks = (frames + fpc - 1) / fpc;
  NPVP_LAUNCH(frameln_act_bwd_fused_kernel, dim3((per_frame / 4 + 255) / 256, nchunks), dim3(256), 0, stream, p,
                     dout, (const float*)psum, dh, part, frames, fpc, FLN_PARTS, amax);
  NPVP_CHECK_LAUNCH();
  if (accumulate == 2) return NPVP_OK;      // the caller reduces the partials itself (npvp_frameln_act_bwd_reduce)
  if (launch_sum_rows((const float*)part, dw, nchunks, 2 * per_frame, 2 * per_frame, stream, accumulate, db, per_frame)) {
    npvp_set_error("frameln_act_bwd: reduce launch failed");
    return NPVP_ERR_LAUNCH;
  }
  return NPVP_OK;
}

// The same with the statistics supplied by the producer of dout: psum [frames][nparts][2] = partial (sum g, sum g*hhat)
// (npvp_mlpdw_mid_bwd emits them for norm1) - ONE pass instead of two.  No dropout / drop-path here (norm1 has none).
// workspace: same layout and size as npvp_frameln_act_bwd (the first frames*2*FLN_PARTS floats stay unused).
extern "C" int npvp_frameln_act_bwd_apply(const float* dout, const float* h, const float* mean, const float* rstd, const float* w,
                                          const float* b, const float* psum, int nparts, float* dh, float* dw, float* db,
                                          int frames, int per_frame, int accumulate, float* amax, void* workspace,
                                          long long ws_bytes, hipStream_t stream) {
  NPVP_CHECK_ARG(frames > 0 && per_frame % 4 == 0 && psum && nparts > 0, "frameln_act_bwd_apply: bad arguments");
  NPVP_CHECK_ARG(workspace && ws_bytes >= npvp_frameln_act_bwd_workspace_bytes(frames, per_frame),
                 "frameln_act_bwd_apply: workspace too small");
  FlnParams p;
  fill_fln(p, h, mean, rstd, w, b, nullptr, frames, per_frame, 0.f, 0u, 0.f, 0u, 1, nullptr);
  float* part = (float*)workspace + (long long)frames * 2 * FLN_PARTS;
  const int chunks = fln_chunks(frames), fpc = (frames + chunks - 1) / chunks;
  const int nchunks = (frames + fpc - 1) / fpc;
  NPVP_LAUNCH(frameln_act_bwd_fused_kernel, dim3((per_frame / 4 + 255) / 256, nchunks), dim3(256), 0, stream, p,
                     dout, psum, dh, part, frames, fpc, nparts, amax);
  NPVP_CHECK_LAUNCH();
  if (accumulate == 2) return NPVP_OK;
  if (launch_sum_rows((const float*)part, dw, nchunks, 2 * per_frame, 2 * per_frame, stream, accumulate, db, per_frame)) {
    npvp_set_error("frameln_act_bwd_apply: reduce launch failed");
    return NPVP_ERR_LAUNCH;
  }
  return NPVP_OK;
}

// Statistics and parameter gradients only (see frameln_act_bwd_pgrad_kernel): psum [frames][per_frame / 1024][2] receives the
// partial (sum g, sum g*hhat); dw / db as npvp_frameln_act_bwd (same workspace layout: npvp_frameln_act_bwd_reduce applies).
extern "C" int npvp_frameln_act_bwd_pgrad(const float* dout, const float* h, const float* mean, const float* rstd, const float* w,
                                          const float* b, float* psum, float* dw, float* db, int frames, int per_frame,
                                          float drop_p, unsigned int salt, float dp_p, unsigned int dp_salt, int frames_per_sample,
                                          const unsigned long long* seed, int accumulate, void* workspace, long long ws_bytes,
                                          hipStream_t stream) {
  NPVP_CHECK_ARG(frames > 0 && per_frame % 1024 == 0 && psum, "frameln_act_bwd_pgrad: per_frame must be a multiple of 1024");
  NPVP_CHECK_ARG(workspace && ws_bytes >= npvp_frameln_act_bwd_workspace_bytes(frames, per_frame),
                 "frameln_act_bwd_pgrad: workspace too small");
  FlnParams p;
  fill_fln(p, h, mean, rstd, w, b, nullptr, frames, per_frame, drop_p, salt, dp_p, dp_salt, frames_per_sample, seed);
  float* part = (float*)workspace + (long long)frames * 2 * FLN_PARTS;
  const int chunks = fln_chunks(frames), fpc = (frames + chunks - 1) / chunks;
  const int nchunks = (frames + fpc - 1) / fpc;
  NPVP_LAUNCH(frameln_act_bwd_pgrad_kernel, dim3(per_frame / 1024, nchunks), dim3(256), 0, stream, p, dout, psum, part,
                     frames, fpc);
  NPVP_CHECK_LAUNCH();
  if (accumulate == 2) return NPVP_OK;
  if (launch_sum_rows((const float*)part, dw, nchunks, 2 * per_frame, 2 * per_frame, stream, accumulate, db, per_frame)) {
    npvp_set_error("frameln_act_bwd_pgrad: reduce launch failed");
    return NPVP_ERR_LAUNCH;
  }
  return NPVP_OK;
}

extern "C" int npvp_frameln_act_bwd_reduce_job(const void* workspace, float* dw, float* db, int frames, int per_frame, int accumulate,
                                               void* job) {
  NPVP_CHECK_ARG(workspace && dw && db && frames > 0 && job, "frameln_act_bwd_reduce_job: bad arguments");
  const int chunks = fln_chunks(frames), fpc = (frames + chunks - 1) / chunks, nchunks = (frames + fpc - 1) / fpc;
  const float* part = (const float*)workspace + 2 * FLN_PARTS * (long long)frames;
  fill_job(job, part, dw, db, nchunks, 2 * per_frame, 2 * per_frame, per_frame, accumulate ? 1 : 0, 0);
  return NPVP_OK;
}

extern "C" int npvp_frameln_act_bwd_reduce(const void* workspace, float* dw, float* db, int frames, int per_frame,
                                           int accumulate, hipStream_t stream) {
  NPVP_CHECK_ARG(workspace && dw && db && frames > 0, "frameln_act_bwd_reduce: bad arguments");
  const int chunks = fln_chunks(frames), fpc = (frames + chunks - 1) / chunks, nchunks = (frames + fpc - 1) / fpc;
  const float* part = (const float*)workspace + 2 * FLN_PARTS * (long long)frames;
  if (launch_sum_rows(part, dw, nchunks, 2 * per_frame, 2 * per_frame, stream, accumulate ? 1 : 0, db, per_frame)) {
    npvp_set_error("frameln_act_bwd_reduce: launch failed");
    return NPVP_ERR_LAUNCH;
  }
  return NPVP_OK;
}
