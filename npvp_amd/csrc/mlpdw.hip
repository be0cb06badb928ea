// Depthwise 3x3 convolution of MlpDWBN (ref/models/VidHRFormer.py:351-358, nn.Conv2d(groups=Ch,
// padding=1), cross-correlation) over the channels-last hidden tensor [F, H*W, Ch], fwd, input
// gradient (same kernel, taps flipped) and weight/bias gradient.  HBM bound: one read + one write of
// the hidden tensor (the 9 neighbour taps hit L1/L2).  Weights are tap-major [9][Ch] so that a lane's
// 4 channels are one float4.
#include "common.h"
#include <cstring>
#include <cstdlib>

namespace npvp {

__global__ void dwconv3x3_kernel(const float* __restrict__ a, const float* __restrict__ wt, const float* __restrict__ bias,
                                 float* __restrict__ out, int H, int W, int Ch, long long total4, int flip) {
  const int c4n = Ch / 4, P = H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const long long fp = i / c4n;
    const int p = (int)(fp % P);
    const long long f = fp / P;
    const int h = p / W, w = p - h * W;
    float4 acc = bias ? ld4(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float* af = a + f * P * Ch + c;
#pragma unroll
    for (int ky = -1; ky <= 1; ++ky) {
      const int hh = h + ky;
      if (hh < 0 || hh >= H) continue;
#pragma unroll
      for (int kx = -1; kx <= 1; ++kx) {
        const int ww = w + kx;
        if (ww < 0 || ww >= W) continue;
        const int tap = (ky + 1) * 3 + (kx + 1);
        const float4 wv = ld4(wt + (flip ? 8 - tap : tap) * Ch + c);
        const float4 v = ld4(af + (long long)(hh * W + ww) * Ch);
        acc.x += wv.x * v.x; acc.y += wv.y * v.y; acc.z += wv.z * v.z; acc.w += wv.w * v.w;
      }
    }
    st4(out + fp * Ch + c, acc);
  }
}


// ---- 8x8 feature grid (every shipped config): register-window forms.  The generic kernel above issues 9 activation +
// 9 weight loads per output float4 - it is bound by the L1/TA request rate (2.3 TB/s effective), not by HBM.  Here a
// thread owns one (frame, 4 channels) column and slides a 3-row window through registers: each input element is
// loaded ONCE, the 9 taps stay in registers, lanes = consecutive channel quads (1 KB contiguous per pixel and wave).
// STATS: the kernel also emits, per block, the (mean, M2) of its 256 threads x 256 outputs (Chan combination of per-thread
// sums taken about the thread's first output) - with Ch % 1024 == 0 a block lies inside one frame, so the frame
// LayerNorm that follows the convolution in MlpDWBN gets its statistics without a pass over the output
// (frame_stats_finalize_kernel merges the Ch/1024 partials of a frame).
template <int HH, int WW, bool STATS>
__global__ __launch_bounds__(256) void dwconv3x3_win_kernel(const float* __restrict__ a, const float* __restrict__ wt,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int Ch, long long nthreads, int flip, float* __restrict__ part) {
  __shared__ float red[4];
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (!STATS && i >= nthreads) return;          // (with STATS the grid is exact: frames * Ch/4 is a multiple of 256)
  float shift = 0.f, s1 = 0.f, s2 = 0.f;
  const int c4n = Ch / 4;
  const int c = (int)(i % c4n) * 4;
  const long long f = i / c4n;
  float4 wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wv[t] = ld4(wt + (flip ? 8 - t : t) * Ch + c);
  const float4 bv = bias ? ld4(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float* af = a + f * (HH * WW) * Ch + c;
  float* of = out + f * (HH * WW) * Ch + c;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 r0[WW], r1[WW], r2[WW];
#pragma unroll
  for (int w = 0; w < WW; ++w) { r0[w] = z; r1[w] = ld4(af + (long long)w * Ch); }
#pragma unroll
  for (int h = 0; h < HH; ++h) {
#pragma unroll
    for (int w = 0; w < WW; ++w) r2[w] = (h + 1 < HH) ? ld4(af + (long long)((h + 1) * WW + w) * Ch) : z;
#pragma unroll
    for (int w = 0; w < WW; ++w) {
      float4 acc = bv;
#pragma unroll
      for (int kx = -1; kx <= 1; ++kx) {
        const int ww = w + kx;
        if (ww < 0 || ww >= WW) continue;
        const float4 w0 = wv[kx + 1], w1 = wv[3 + kx + 1], w2 = wv[6 + kx + 1];
        const float4 x0 = r0[ww], x1 = r1[ww], x2 = r2[ww];
        acc.x += w0.x * x0.x + w1.x * x1.x + w2.x * x2.x; acc.y += w0.y * x0.y + w1.y * x1.y + w2.y * x2.y;
        acc.z += w0.z * x0.z + w1.z * x1.z + w2.z * x2.z; acc.w += w0.w * x0.w + w1.w * x1.w + w2.w * x2.w;
      }
      st4(of + (long long)(h * WW + w) * Ch, acc);
      if constexpr (STATS) {
        if (h == 0 && w == 0) shift = acc.x;
        const float a0 = acc.x - shift, a1 = acc.y - shift, a2 = acc.z - shift, a3 = acc.w - shift;
        s1 += (a0 + a1) + (a2 + a3);
        s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
      }
    }
#pragma unroll
    for (int w = 0; w < WW; ++w) { r0[w] = r1[w]; r1[w] = r2[w]; }
  }
  if constexpr (STATS) {
    const float n = (float)(HH * WW * 4), m1 = s1 / n, mean_t = shift + m1, m2_t = s2 - s1 * m1;
    const float mean_b = block_sum<4>(mean_t, red) / 256.f;
    const float d = mean_t - mean_b;
    const float m2_b = block_sum<4>(m2_t + n * d * d, red);
    if (threadIdx.x == 0) { part[blockIdx.x * 2] = mean_b; part[blockIdx.x * 2 + 1] = m2_b; }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused middle of MlpDWBN (ref/models/VidHRFormer.py:381-385): norm1 + GELU + depthwise 3x3 (+ the statistics norm2 needs)
// in ONE pass over the hidden tensor.  a1 = GELU(LayerNorm((Ch,H,W))(h1)) is never written to memory: forward reads h1 and
// writes h2 (2 passes over [R, Ch] instead of 4), backward recomputes a1 from h1 where the weight gradient needs it.
// Same register window as dwconv3x3_win_kernel, but VEC = 2 channels per thread instead of 4: the LayerNorm / GELU
// arithmetic and the two affine loads per element need registers and latency hiding that a 4-channel window (1 wave per
// SIMD) cannot give (that first attempt lost 176 us per block in backward); at 2 channels the kernels run 3-5 waves per SIMD.
template <int VEC> struct Vec { float v[VEC]; };
template <int VEC> __device__ __forceinline__ Vec<VEC> ldv(const float* p) {
  Vec<VEC> r;
  if constexpr (VEC == 4) { const float4 t = ld4(p); r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; }
  else if constexpr (VEC == 2) { const float2 t = *reinterpret_cast<const float2*>(p); r.v[0] = t.x; r.v[1] = t.y; }
  else r.v[0] = *p;
  return r;
}
template <int VEC> __device__ __forceinline__ void stv(float* p, const Vec<VEC>& r) {
  if constexpr (VEC == 4) st4(p, make_float4(r.v[0], r.v[1], r.v[2], r.v[3]));
  else if constexpr (VEC == 2) *reinterpret_cast<float2*>(p) = make_float2(r.v[0], r.v[1]);
  else *p = r.v[0];
}

// h2[f,p,c] = bias[c] + sum_tap wt[tap][c] * a1[f, p + off(tap), c],  a1 = gelu((h1 - mean1[f]) rstd1[f] w1n[p,c] + b1n[p,c]);
// part[block] = (mean, M2) of the block's 256 x 64 x VEC outputs (a block lies inside one frame: (Ch / VEC) % 256 == 0).
template <int VEC>
__global__ __launch_bounds__(256, 4) void mlpdw_mid_fwd_kernel(const float* __restrict__ h1, const float* __restrict__ mean1,
                                                            const float* __restrict__ rstd1, const float* __restrict__ w1n,
                                                            const float* __restrict__ b1n, const float* __restrict__ wt,
                                                            const float* __restrict__ bias, float* __restrict__ h2,
                                                            float* __restrict__ part, int Ch, const float* __restrict__ part1,
                                                            int J1, float nb1, float eps1, float* __restrict__ mean1_out,
                                                            float* __restrict__ rstd1_out) {
  constexpr int HH = 8, WW = 8;
  __shared__ float red[4];
  // a block lies inside one frame ((Ch / VEC) % 256 == 0): the frame base is wave-uniform (SGPR pair) and every access is
  // base + a 32-bit offset - with per-thread 64-bit pointers the 8 x 4 addresses of a row alone took 64 VGPRs
  const int bpf = Ch / VEC / 256;                                   // blocks per frame
  const long long f = blockIdx.x / bpf;
  const int c = ((blockIdx.x - (int)f * bpf) * 256 + threadIdx.x) * VEC;
  float mu, rs;
  if (part1) {          // norm1's statistics from the partials the fc1 GEMM's epilogue left (no finalize launch); saved for backward
    frame_stats_merge(part1, f, J1, nb1, eps1, mu, rs);
    if (blockIdx.x == f * bpf && threadIdx.x == 0) { mean1_out[f] = mu; rstd1_out[f] = rs; }
  } else { mu = mean1[f]; rs = rstd1[f]; }
  Vec<VEC> wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wv[t] = ldv<VEC>(wt + t * Ch + c);
  const Vec<VEC> bv = ldv<VEC>(bias + c);
  const float* hf = h1 + f * (HH * WW) * Ch;
  float* of = h2 + f * (HH * WW) * Ch;
  Vec<VEC> r0[WW], r1[WW], r2[WW];
  auto load_row = [&](Vec<VEC>* row, int hh) {
#pragma unroll
    for (int w = 0; w < WW; ++w) {
      const int pix = hh * WW + w, off = pix * Ch + c;
      const Vec<VEC> x = ldv<VEC>(hf + off), ww = ldv<VEC>(w1n + off), bb = ldv<VEC>(b1n + off);
#pragma unroll
      for (int e = 0; e < VEC; ++e) row[w].v[e] = gelu_f((x.v[e] - mu) * rs * ww.v[e] + bb.v[e]);
    }
  };
#pragma unroll
  for (int w = 0; w < WW; ++w)
#pragma unroll
    for (int e = 0; e < VEC; ++e) r0[w].v[e] = 0.f;
  load_row(r1, 0);
  float shift = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll 1
  for (int h = 0; h < HH; ++h) {          // NOT unrolled: unrolled, the compiler hoists a frame's 64 x 3 loads (256 VGPRs)
    if (h + 1 < HH) load_row(r2, h + 1);
    else {
#pragma unroll
      for (int w = 0; w < WW; ++w)
#pragma unroll
        for (int e = 0; e < VEC; ++e) r2[w].v[e] = 0.f;
    }
#pragma unroll
    for (int w = 0; w < WW; ++w) {
      Vec<VEC> acc = bv;
#pragma unroll
      for (int kx = -1; kx <= 1; ++kx) {
        const int ww = w + kx;
        if (ww < 0 || ww >= WW) continue;
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          acc.v[e] += wv[kx + 1].v[e] * r0[ww].v[e] + wv[3 + kx + 1].v[e] * r1[ww].v[e] + wv[6 + kx + 1].v[e] * r2[ww].v[e];
      }
      stv<VEC>(of + (h * WW + w) * Ch + c, acc);
      if (h == 0 && w == 0) shift = acc.v[0];
#pragma unroll
      for (int e = 0; e < VEC; ++e) { const float a = acc.v[e] - shift; s1 += a; s2 += a * a; }
    }
#pragma unroll
    for (int w = 0; w < WW; ++w) { r0[w] = r1[w]; r1[w] = r2[w]; }
  }
  const float n = (float)(HH * WW * VEC), m1 = s1 / n, mean_t = shift + m1, m2_t = s2 - s1 * m1;
  const float mean_b = block_sum<4>(mean_t, red) / 256.f;
  const float d = mean_t - mean_b;
  const float m2_b = block_sum<4>(m2_t + n * d * d, red);
  if (threadIdx.x == 0) { part[blockIdx.x * 2] = mean_b; part[blockIdx.x * 2 + 1] = m2_b; }
}

// Backward of the fused middle for the frames of one chunk (grid.y) and VEC channels per thread:
//   da1[f,p,c]  = sum_tap wt[tap][c] * dh2[f, p - off(tap), c]                       (input gradient of the convolution)
//   dwt[tap][c] += sum_{f,p} dh2[f,p,c] * a1[f, p + off(tap), c],  db[c] += sum dh2   (a1 recomputed from h1)
//   psum[f][block] = (sum g, sum g*hhat) over the block's channels, g = da1 * gelu'(y1) * w1n   (norm1's backward statistics:
//                    frameln_act_bwd then needs no statistics pass of its own)
// part[chunk][tap 0..8 | bias][Ch] receives the weight-gradient partial sums (summed by sum_rows_kernel).
#ifndef NPVP_MID_BWD_WAVES
#define NPVP_MID_BWD_WAVES 2
#endif
// FUSE2: `dh2` is da2, the gradient w.r.t. a2 = drop(gelu(norm2(h2))), and the kernel evaluates norm2's input gradient
//   dh2 = rstd2 (g - s1 - hhat2 s2),  g = da2 * mask * gelu'(y2) * w2n,  (s1, s2) = frame means of (g, g*hhat2)
// element by element as it loads the window (the frame sums arrive as n2.psum2 partials from frameln_act_bwd_pgrad_kernel): dh2
// is never written or read - two passes over the [R, hidden] tensor less per MlpDWBN backward.
struct MidN2 {
  const float* h2; const float* mean2; const float* rstd2; const float* w2n; const float* b2n;
  const float* psum2; int nparts2;
  const unsigned long long* seed; unsigned int drop_thresh; float drop_inv_keep; unsigned int salt;
};

// Round 5 measured a different thread map for the FUSE2 form - the 8 pixel columns on 8 lanes (neighbours by DPP row shifts), four
// channels per thread, float4 loads (7 per row instead of 56 dword loads), no block-wide barrier at all - in the belief that this
// kernel is bound by its 448 load instructions per frame and its four barriers per frame.  It is not: bit-for-bit the same results,
// 203 VGPRs, and the layer's backward went 5.04 -> 5.43 ms at the c2 size (profiles/r05_mlpdw_mid_bwd_cols.txt).  The kernel
// executes ~110 vector instructions per element (two GELU' evaluations and one GELU with their exponentials and reciprocals, the
// mask hash, 18 tap multiply-adds): 0.92 M elements per CU x 110 / 64 lanes per cycle = 1.6 M cycles = 0.9 ms at 1.7 GHz - what
// it takes.  It is VALU bound; only fewer instructions per element would move it.  The column-lane kernel is in the git history.
template <int VEC, bool FUSE2>
__global__ __launch_bounds__(256, NPVP_MID_BWD_WAVES) void mlpdw_mid_bwd_kernel(const float* __restrict__ dh2, const float* __restrict__ h1,
                                                            const float* __restrict__ mean1, const float* __restrict__ rstd1,
                                                            const float* __restrict__ w1n, const float* __restrict__ b1n,
                                                            const float* __restrict__ wt, float* __restrict__ da1,
                                                            float* __restrict__ part, float* __restrict__ psum, int Ch,
                                                            int frames, int frames_per_chunk, MidN2 n2) {
  constexpr int HH = 8, WW = 8;
  __shared__ float red[4];
  const unsigned long long seed2 = (FUSE2 && n2.seed && n2.drop_thresh) ? *n2.seed : 0ull;
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  const int nblk = gridDim.x;
  Vec<VEC> wv[9], aw[9], ab;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    wv[t] = ldv<VEC>(wt + t * Ch + c);
#pragma unroll
    for (int e = 0; e < VEC; ++e) aw[t].v[e] = 0.f;
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) ab.v[e] = 0.f;
  const int f0 = blockIdx.y * frames_per_chunk, f1 = min(frames, f0 + frames_per_chunk);
  for (long long f = f0; f < f1; ++f) {
    const float mu = mean1[f], rs = rstd1[f];
    const float* hf = h1 + f * (HH * WW) * Ch;           // wave-uniform bases + 32-bit offsets (see the forward kernel)
    const float* gf = dh2 + f * (HH * WW) * Ch;
    float* of = da1 + f * (HH * WW) * Ch;
    float mu2 = 0.f, rs2 = 0.f, c1 = 0.f, c2 = 0.f;
    const float* h2f = nullptr;
    if constexpr (FUSE2) {
      mu2 = n2.mean2[f]; rs2 = n2.rstd2[f];
      h2f = n2.h2 + f * (HH * WW) * Ch;
      const int j = threadIdx.x;                          // frame sums of norm2's backward: nparts2 <= 256 partials, fixed order
      const float p1 = j < n2.nparts2 ? n2.psum2[(f * n2.nparts2 + j) * 2] : 0.f;
      const float p2 = j < n2.nparts2 ? n2.psum2[(f * n2.nparts2 + j) * 2 + 1] : 0.f;
      const float inv = 1.f / (float)(HH * WW * Ch);
      c1 = block_sum<4>(p1, red) * inv;
      c2 = block_sum<4>(p2, red) * inv;
    }
    // windows: a = a1 rows (h-1, h, h+1); g = dh2 rows; for the centre row also gelu'(y1) * w1n and hhat
    Vec<VEC> a0[WW], a1r[WW], a2[WW], g0[WW], g1[WW], g2[WW], gp1[WW], xh1[WW], gp2[WW], xh2[WW];
    auto load_row = [&](Vec<VEC>* arow, Vec<VEC>* grow, Vec<VEC>* gprow, Vec<VEC>* xhrow, int hh) {
      if constexpr (FUSE2) {
        // norm2's input gradient for the row, four pixels at a time (the scheduling fences keep the loads of a group from
        // being hoisted over the whole row: the window already holds 10 rows of 8 values)
#pragma unroll
        for (int w0 = 0; w0 < WW; w0 += 4) {
#pragma unroll
          for (int w = w0; w < w0 + 4; ++w) {
            const int off = (hh * WW + w) * Ch + c;
            const Vec<VEC> d2 = ldv<VEC>(gf + off), x2 = ldv<VEC>(h2f + off), w2 = ldv<VEC>(n2.w2n + off), b2 = ldv<VEC>(n2.b2n + off);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              const float xh2 = (x2.v[e] - mu2) * rs2;
              float m = 1.f;
              if (n2.drop_thresh)
                m = drop_scale(seed2, n2.salt, (unsigned long long)(f * (HH * WW) * Ch + off + e), n2.drop_thresh, n2.drop_inv_keep);
              const float dyln = d2.v[e] * m * gelu_grad_f(xh2 * w2.v[e] + b2.v[e]);
              grow[w].v[e] = rs2 * (dyln * w2.v[e] - c1 - xh2 * c2);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int w = 0; w < WW; ++w) {
        const int pix = hh * WW + w, off = pix * Ch + c;
        const Vec<VEC> x = ldv<VEC>(hf + off), ww = ldv<VEC>(w1n + off), bb = ldv<VEC>(b1n + off);
        if constexpr (!FUSE2) grow[w] = ldv<VEC>(gf + off);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float xh = (x.v[e] - mu) * rs, y = xh * ww.v[e] + bb.v[e];
          float E;
          const float phi = gelu_phi(y, E);
          arow[w].v[e] = y * phi;
          gprow[w].v[e] = fmaf(y * 0.39894228040143267794f, E, phi) * ww.v[e];
          xhrow[w].v[e] = xh;
        }
      }
    };
#pragma unroll
    for (int w = 0; w < WW; ++w)
#pragma unroll
      for (int e = 0; e < VEC; ++e) { a0[w].v[e] = 0.f; g0[w].v[e] = 0.f; }
    load_row(a1r, g1, gp1, xh1, 0);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 1
    for (int h = 0; h < HH; ++h) {
      if (h + 1 < HH) load_row(a2, g2, gp2, xh2, h + 1);
      else {
#pragma unroll
        for (int w = 0; w < WW; ++w)
#pragma unroll
          for (int e = 0; e < VEC; ++e) { a2[w].v[e] = 0.f; g2[w].v[e] = 0.f; }
      }
#pragma unroll
      for (int w = 0; w < WW; ++w) {
        Vec<VEC> acc;
#pragma unroll
        for (int e = 0; e < VEC; ++e) { acc.v[e] = 0.f; ab.v[e] += g1[w].v[e]; }
#pragma unroll
        for (int kx = -1; kx <= 1; ++kx) {
          const int ww = w + kx;
          if (ww < 0 || ww >= WW) continue;
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            // input gradient: flipped taps on the dh2 window
            acc.v[e] += wv[8 - (kx + 1)].v[e] * g0[ww].v[e] + wv[8 - (3 + kx + 1)].v[e] * g1[ww].v[e] + wv[8 - (6 + kx + 1)].v[e] * g2[ww].v[e];
            // weight gradient: dh2 at the output pixel times a1 at its 3x3 neighbourhood
            const float d = g1[w].v[e];
            // one v_fmac_f32 each, kept out of the compiler's v_pk_fma_f32 pairing: the paired form needs 242 VGPRs
            // (this one 188), is 5 % slower, and its results were not reproducible run to run while a weight-gradient
            // GEMM shared the CUs
            fmac_scalar(aw[kx + 1].v[e], d, a0[ww].v[e]);
            fmac_scalar(aw[3 + kx + 1].v[e], d, a1r[ww].v[e]);
            fmac_scalar(aw[6 + kx + 1].v[e], d, a2[ww].v[e]);
          }
        }
        stv<VEC>(of + (h * WW + w) * Ch + c, acc);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { const float g = acc.v[e] * gp1[w].v[e]; s1 += g; s2 += g * xh1[w].v[e]; }
      }
#pragma unroll
      for (int w = 0; w < WW; ++w) { a0[w] = a1r[w]; a1r[w] = a2[w]; g0[w] = g1[w]; g1[w] = g2[w]; gp1[w] = gp2[w]; xh1[w] = xh2[w]; }
    }
    s1 = block_sum<4>(s1, red);
    s2 = block_sum<4>(s2, red);
    if (threadIdx.x == 0) { psum[(f * nblk + blockIdx.x) * 2] = s1; psum[(f * nblk + blockIdx.x) * 2 + 1] = s2; }
  }
  float* o = part + (long long)blockIdx.y * 10 * Ch;
#pragma unroll
  for (int t = 0; t < 9; ++t) stv<VEC>(o + t * Ch + c, aw[t]);
  stv<VEC>(o + 9 * Ch + c, ab);
}

// per frame: J partials (mean_j, M2_j) over nb values each -> mean, rstd
__global__ void frame_stats_finalize_kernel(const float* __restrict__ part, int J, float nb, float* __restrict__ mean,
                                            float* __restrict__ rstd, int frames, float eps) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= frames) return;
  float m = 0.f;
  for (int j = 0; j < J; ++j) m += part[(f * J + j) * 2];
  m /= J;
  float m2 = 0.f;
  for (int j = 0; j < J; ++j) { const float d = part[(f * J + j) * 2] - m; m2 += part[(f * J + j) * 2 + 1] + nb * d * d; }
  mean[f] = m;
  rstd[f] = rsqrtf(m2 / (nb * J) + eps);
}

// weight / bias gradient, same window: thread = (4 channels, frame chunk); per frame one load of a and of dout.
template <int HH, int WW>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_win_kernel(const float* __restrict__ a, const float* __restrict__ dout,
                                                                  float* __restrict__ part, int Ch, int frames,
                                                                  int frames_per_chunk) {
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (c >= Ch) return;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 aw[9], ab = z;
#pragma unroll
  for (int t = 0; t < 9; ++t) aw[t] = z;
  const int f0 = blockIdx.y * frames_per_chunk, f1 = min(frames, f0 + frames_per_chunk);
  for (long long f = f0; f < f1; ++f) {
    const float* af = a + f * (HH * WW) * Ch + c;
    const float* df = dout + f * (HH * WW) * Ch + c;
    float4 r0[WW], r1[WW], r2[WW];
#pragma unroll
    for (int w = 0; w < WW; ++w) { r0[w] = z; r1[w] = ld4(af + (long long)w * Ch); }
#pragma unroll
    for (int h = 0; h < HH; ++h) {
#pragma unroll
      for (int w = 0; w < WW; ++w) r2[w] = (h + 1 < HH) ? ld4(af + (long long)((h + 1) * WW + w) * Ch) : z;
#pragma unroll
      for (int w = 0; w < WW; ++w) {
        const float4 d = ld4(df + (long long)(h * WW + w) * Ch);
        ab.x += d.x; ab.y += d.y; ab.z += d.z; ab.w += d.w;
#pragma unroll
        for (int kx = -1; kx <= 1; ++kx) {
          const int ww = w + kx;
          if (ww < 0 || ww >= WW) continue;
          const float4 x0 = r0[ww], x1 = r1[ww], x2 = r2[ww];
          float4& t0 = aw[kx + 1]; float4& t1 = aw[3 + kx + 1]; float4& t2 = aw[6 + kx + 1];
          t0.x += d.x * x0.x; t0.y += d.y * x0.y; t0.z += d.z * x0.z; t0.w += d.w * x0.w;
          t1.x += d.x * x1.x; t1.y += d.y * x1.y; t1.z += d.z * x1.z; t1.w += d.w * x1.w;
          t2.x += d.x * x2.x; t2.y += d.y * x2.y; t2.z += d.z * x2.z; t2.w += d.w * x2.w;
        }
      }
#pragma unroll
      for (int w = 0; w < WW; ++w) { r0[w] = r1[w]; r1[w] = r2[w]; }
    }
  }
  float* o = part + (long long)blockIdx.y * 10 * Ch;
#pragma unroll
  for (int t = 0; t < 9; ++t) st4(o + t * Ch + c, aw[t]);
  st4(o + 9 * Ch + c, ab);
}

// part[chunk][tap 0..8 | bias][Ch]:  dW[tap][c] = sum_{f,p} dout[f,p,c] * a[f, p + tap, c];  db[c] = sum dout.
// Block = 64 channel-quads x 4 pixel groups (pixel p handled by group p & 3): 4x the loads in flight of the first
// version (one thread per channel-quad walking all 64 pixels), fixed-order LDS reduction over the 4 groups.
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const float* __restrict__ a, const float* __restrict__ dout,
                                                              float* __restrict__ part, int H, int W, int Ch, int frames,
                                                              int frames_per_chunk) {
  __shared__ float4 red[3][10][64];
  const int cx = threadIdx.x & 63, pg = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cx) * 4;
  const bool live = c < Ch;
  const int P = H * W;
  float4 aw[9], ab = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int t = 0; t < 9; ++t) aw[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int f0 = blockIdx.y * frames_per_chunk, f1 = min(frames, f0 + frames_per_chunk);
  if (live) {
    for (long long f = f0; f < f1; ++f) {
      const float* af = a + f * P * Ch + c;
      const float* df = dout + f * P * Ch + c;
      for (int p = pg; p < P; p += 4) {
        const int h = p / W, w = p - h * W;
        const float4 d = ld4(df + (long long)p * Ch);
        ab.x += d.x; ab.y += d.y; ab.z += d.z; ab.w += d.w;
#pragma unroll
        for (int ky = -1; ky <= 1; ++ky) {
          const int hh = h + ky;
          if (hh < 0 || hh >= H) continue;
#pragma unroll
          for (int kx = -1; kx <= 1; ++kx) {
            const int ww = w + kx;
            if (ww < 0 || ww >= W) continue;
            const float4 v = ld4(af + (long long)(hh * W + ww) * Ch);
            float4& t = aw[(ky + 1) * 3 + (kx + 1)];
            t.x += d.x * v.x; t.y += d.y * v.y; t.z += d.z * v.z; t.w += d.w * v.w;
          }
        }
      }
    }
  }
  if (pg > 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t) red[pg - 1][t][cx] = aw[t];
    red[pg - 1][9][cx] = ab;
  }
  __syncthreads();
  if (pg == 0 && live) {
    float* o = part + (long long)blockIdx.y * 10 * Ch;
#pragma unroll
    for (int t = 0; t < 10; ++t) {
      float4 s = t < 9 ? aw[t] : ab;
#pragma unroll
      for (int g = 0; g < 3; ++g) { const float4 q = red[g][t][cx]; s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w; }
      st4(o + t * Ch + c, s);
    }
  }
}

// im2col for a 3x3 / pad 1 convolution over channels-last frames (EventEncoder conv2, ref/models/submodules.py:376):
//   fwd (col2im = 0): out[(f,p)][tap*C + c] = in[f][p + off(tap)][c] (0 outside the grid)     in [F][P][C], out [F*P][9C]
//   bwd (col2im = 1): out[f][q][c] = sum_tap in[(f, q - off(tap))][tap*C + c]                  in [F*P][9C], out [F][P][C]
__global__ void im2col3x3_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int C, long long total4,
                                 int col2im) {
  const int c4n = C / 4, P = H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    if (!col2im) {
      const int c = (int)(i % c4n) * 4;
      long long r = i / c4n;
      const int tap = (int)(r % 9); r /= 9;                 // r = f*P + p
      const int p = (int)(r % P);
      const long long f = r / P;
      const int h = p / W + tap / 3 - 1, w = p % W + tap % 3 - 1;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (h >= 0 && h < H && w >= 0 && w < W) v = ld4(in + (f * P + h * W + w) * C + c);
      st4(out + r * 9 * C + (long long)tap * C + c, v);
    } else {
      const int c = (int)(i % c4n) * 4;
      const long long r = i / c4n;                           // r = f*P + q
      const int q = (int)(r % P);
      const long long f = r / P;
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int h = q / W - (tap / 3 - 1), w = q % W - (tap % 3 - 1);
        if (h < 0 || h >= H || w < 0 || w >= W) continue;
        const float4 v = ld4(in + (f * P + h * W + w) * 9 * C + (long long)tap * C + c);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      st4(out + r * C + c, s);
    }
  }
}

static int dw_chunks(int frames) { return frames < 256 ? frames : 256; }

}  // namespace npvp

using namespace npvp;

extern "C" int npvp_dwconv3x3(const float* a, const float* wt, const float* bias, float* out, int frames, int H, int W,
                              int Ch, int flip, hipStream_t stream) {
  NPVP_CHECK_ARG(frames > 0 && H > 0 && W > 0 && Ch % 4 == 0, "dwconv: bad shape");
  if (H == 8 && W == 8) {
    const long long nthreads = (long long)frames * (Ch / 4);
    NPVP_LAUNCH((dwconv3x3_win_kernel<8, 8, false>), dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, stream, a,
                       wt, bias, out, Ch, nthreads, flip, (float*)nullptr);
    NPVP_CHECK_LAUNCH();
    return NPVP_OK;
  }
  const long long total4 = (long long)frames * H * W * Ch / 4;
  long long blocks = (total4 + 255) / 256; if (blocks > 8192) blocks = 8192;
  NPVP_LAUNCH(dwconv3x3_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, wt, bias, out, H, W, Ch, total4, flip);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

extern "C" int npvp_frame_stats_finalize(const float* part, int parts_per_frame, float values_per_part, float* mean,
                                         float* rstd, int frames, float eps, hipStream_t stream) {
  NPVP_CHECK_ARG(part && mean && rstd && frames > 0 && parts_per_frame > 0, "frame_stats_finalize: bad arguments");
  NPVP_LAUNCH(frame_stats_finalize_kernel, dim3((frames + 255) / 256), dim3(256), 0, stream, part, parts_per_frame,
                     values_per_part, mean, rstd, frames, eps);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

// forward convolution that also returns the per-frame LayerNorm statistics (mean, rstd over the frame's H*W*Ch outputs)
// of its result.  8x8 grid, Ch % 1024 == 0; workspace >= frames * (Ch/1024) * 8 bytes.
extern "C" int npvp_dwconv3x3_stats(const float* a, const float* wt, const float* bias, float* out, float* mean, float* rstd,
                                    int frames, int H, int W, int Ch, float eps, void* workspace, long long ws_bytes,
                                    hipStream_t stream) {
  NPVP_CHECK_ARG(frames > 0 && H == 8 && W == 8 && Ch > 0 && Ch % 1024 == 0, "dwconv_stats: needs an 8x8 grid and Ch % 1024 == 0");
  const int J = Ch / 1024;
  NPVP_CHECK_ARG(workspace && ws_bytes >= (long long)frames * J * 8, "dwconv_stats: workspace too small");
  const long long nthreads = (long long)frames * (Ch / 4);
  NPVP_LAUNCH((dwconv3x3_win_kernel<8, 8, true>), dim3((unsigned)(nthreads / 256)), dim3(256), 0, stream, a, wt, bias, out,
                     Ch, nthreads, 0, (float*)workspace);
  NPVP_CHECK_LAUNCH();
  NPVP_LAUNCH(frame_stats_finalize_kernel, dim3((frames + 255) / 256), dim3(256), 0, stream, (const float*)workspace, J,
                     65536.f, mean, rstd, frames, eps);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

// Fused MlpDWBN middle, forward (8x8 grid, Ch % 512 == 0): h2 = dwconv3x3(gelu(frame_ln(h1))) + bias, and the frame
// statistics (mean2, rstd2) of h2.  wt [9][Ch] tap-major, w1n / b1n [64][Ch] channels-last.  workspace >= frames*(Ch/512)*8 B.
extern "C" int npvp_mlpdw_mid_fwd(const float* h1, const float* mean1, const float* rstd1, const float* w1n, const float* b1n,
                                  const float* wt, const float* bias, float* h2, float* mean2, float* rstd2, int frames,
                                  int H, int W, int Ch, float eps, void* workspace, long long ws_bytes, hipStream_t stream) {
  NPVP_CHECK_ARG(frames > 0 && H == 8 && W == 8 && Ch > 0 && Ch % 512 == 0, "mlpdw_mid_fwd: needs an 8x8 grid and Ch % 512 == 0");
  const int J = Ch / 512;
  NPVP_CHECK_ARG(workspace && ws_bytes >= (long long)frames * J * 8, "mlpdw_mid_fwd: workspace too small");
  const long long nthreads = (long long)frames * (Ch / 2);
  NPVP_LAUNCH((mlpdw_mid_fwd_kernel<2>), dim3((unsigned)(nthreads / 256)), dim3(256), 0, stream, h1, mean1, rstd1, w1n, b1n,
                     wt, bias, h2, (float*)workspace, Ch, (const float*)nullptr, 0, 0.f, 0.f, (float*)nullptr, (float*)nullptr);
  NPVP_CHECK_LAUNCH();
  NPVP_LAUNCH(frame_stats_finalize_kernel, dim3((frames + 255) / 256), dim3(256), 0, stream, (const float*)workspace, J,
                     32768.f, mean2, rstd2, frames, eps);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

// The same without either statistics launch: norm1's statistics come as the partials part1 [frames][J1][2] (over nb1 values each)
// that the fc1 GEMM's epilogue left (rowstats of npvp_gemm_f32: J1 = Ch / 64, nb1 = 4096) and are merged by every block for its
// frame (mean1 / rstd1 are OUTPUTS here, for backward); h2's partials part2 [frames][Ch / 512][2] (32 768 values each) are left
// for npvp_frameln_act_fwd_parts.
extern "C" int npvp_mlpdw_mid_fwd_parts(const float* h1, const float* part1, int J1, float nb1, float* mean1, float* rstd1,
                                        const float* w1n, const float* b1n, const float* wt, const float* bias, float* h2,
                                        float* part2, int frames, int H, int W, int Ch, float eps, hipStream_t stream) {
  NPVP_CHECK_ARG(frames > 0 && H == 8 && W == 8 && Ch > 0 && Ch % 512 == 0, "mlpdw_mid_fwd_parts: needs an 8x8 grid and Ch % 512 == 0");
  NPVP_CHECK_ARG(part1 && J1 > 0 && nb1 > 0.f && mean1 && rstd1 && part2, "mlpdw_mid_fwd_parts: bad arguments");
  const long long nthreads = (long long)frames * (Ch / 2);
  NPVP_LAUNCH((mlpdw_mid_fwd_kernel<2>), dim3((unsigned)(nthreads / 256)), dim3(256), 0, stream, h1, (const float*)nullptr,
                     (const float*)nullptr, w1n, b1n, wt, bias, h2, part2, Ch, part1, J1, nb1, eps, mean1, rstd1);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

static int mid_chunks(int frames) { return frames < 128 ? frames : 128; }      // 256 / 512 chunks: no change of the c2 step

// workspace of npvp_mlpdw_mid_bwd: weight-gradient partials [chunks][10][Ch]
extern "C" long long npvp_mlpdw_mid_bwd_workspace_bytes(int frames, int Ch) {
  return (long long)mid_chunks(frames) * 10 * Ch * 4;
}

// Fused MlpDWBN middle, backward: da1 (gradient w.r.t. a1 = gelu(norm1(h1)), [frames, 64, Ch]), dwt_db [10][Ch] (depthwise
// weight taps + bias gradient, written or accumulated), and psum [frames][Ch/256][2] = the statistics norm1's backward needs
// (pass to npvp_frameln_act_bwd_apply with nparts = Ch / 256).
static int mid_bwd_launch(const float* dh2, const float* h1, const float* mean1, const float* rstd1, const float* w1n,
                          const float* b1n, const float* wt, float* da1, float* dwt_db, float* psum, int frames, int H, int W,
                          int Ch, int accumulate, void* workspace, long long ws_bytes, hipStream_t stream, const MidN2* n2) {
  NPVP_CHECK_ARG(frames > 0 && H == 8 && W == 8 && Ch > 0 && Ch % 512 == 0, "mlpdw_mid_bwd: needs an 8x8 grid and Ch % 512 == 0");
  NPVP_CHECK_ARG(workspace && ws_bytes >= npvp_mlpdw_mid_bwd_workspace_bytes(frames, Ch), "mlpdw_mid_bwd: workspace too small");
  const int chunks = mid_chunks(frames), fpc = (frames + chunks - 1) / chunks, nchunks = (frames + fpc - 1) / fpc;
  // one channel per thread: the windows of a1, dh2, gelu'(y1) w1n and hhat (10 rows of 8) fit 242 VGPRs without spilling
  if (n2)
    NPVP_LAUNCH((mlpdw_mid_bwd_kernel<1, true>), dim3(Ch / 256, nchunks), dim3(256), 0, stream, dh2, h1, mean1, rstd1, w1n,
                       b1n, wt, da1, (float*)workspace, psum, Ch, frames, fpc, *n2);
  else
    NPVP_LAUNCH((mlpdw_mid_bwd_kernel<1, false>), dim3(Ch / 256, nchunks), dim3(256), 0, stream, dh2, h1, mean1, rstd1, w1n,
                       b1n, wt, da1, (float*)workspace, psum, Ch, frames, fpc, MidN2{});
  NPVP_CHECK_LAUNCH();
  if (accumulate == 2) return NPVP_OK;        // the caller reduces the partials (npvp_mlpdw_mid_bwd_reduce)
  const int rc = launch_sum_rows((const float*)workspace, dwt_db, nchunks, 10 * Ch, 10 * Ch, stream, accumulate);
  if (rc) { npvp_set_error("mlpdw_mid_bwd: reduce launch failed"); return rc; }
  return NPVP_OK;
}

extern "C" int npvp_mlpdw_mid_bwd(const float* dh2, const float* h1, const float* mean1, const float* rstd1, const float* w1n,
                                  const float* b1n, const float* wt, float* da1, float* dwt_db, float* psum, int frames, int H,
                                  int W, int Ch, int accumulate, void* workspace, long long ws_bytes, hipStream_t stream) {
  return mid_bwd_launch(dh2, h1, mean1, rstd1, w1n, b1n, wt, da1, dwt_db, psum, frames, H, W, Ch, accumulate, workspace, ws_bytes,
                        stream, nullptr);
}

// The same with norm2's input gradient evaluated inside (see MidN2): da2 = gradient w.r.t. a2 = drop(gelu(norm2(h2))) with
// elementwise dropout (drop_p, salt; 0 = none), psum2 [frames][nparts2 <= 256][2] = the partial (sum g, sum g*hhat2) that
// npvp_frameln_act_bwd_pgrad produced for norm2.
extern "C" int npvp_mlpdw_mid_bwd_n2(const float* da2, const float* h2, const float* mean2, const float* rstd2, const float* w2n,
                                     const float* b2n, const float* psum2, int nparts2, float drop_p, unsigned int salt,
                                     const unsigned long long* seed, const float* h1, const float* mean1, const float* rstd1,
                                     const float* w1n, const float* b1n, const float* wt, float* da1, float* dwt_db, float* psum,
                                     int frames, int H, int W, int Ch, int accumulate, void* workspace, long long ws_bytes,
                                     hipStream_t stream) {
  NPVP_CHECK_ARG(h2 && mean2 && rstd2 && w2n && b2n && psum2 && nparts2 > 0 && nparts2 <= 256, "mlpdw_mid_bwd_n2: bad norm2 arguments");
  NPVP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed), "mlpdw_mid_bwd_n2: bad dropout arguments");
  MidN2 n2;
  n2.h2 = h2; n2.mean2 = mean2; n2.rstd2 = rstd2; n2.w2n = w2n; n2.b2n = b2n; n2.psum2 = psum2; n2.nparts2 = nparts2;
  n2.seed = seed; n2.drop_thresh = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
  n2.drop_inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f; n2.salt = salt;
  return mid_bwd_launch(da2, h1, mean1, rstd1, w1n, b1n, wt, da1, dwt_db, psum, frames, H, W, Ch, accumulate, workspace, ws_bytes,
                        stream, &n2);
}

// partial sums [chunks][10][Ch] (npvp_mlpdw_mid_bwd* with accumulate = 2) -> the Conv2d parameter gradients, in place:
// gw[c][tap] += sum_chunks part[.][tap][c], gb[c] += sum_chunks part[.][9][c]  (reduction, transposition and accumulation in ONE
// launch: it replaces a sum_rows launch on the compute stream + npvp_dwtb_accumulate)
__global__ void mid_bwd_reduce_into_kernel(const float* __restrict__ part, float* __restrict__ gw, float* __restrict__ gb, int Ch,
                                           int nchunks) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;            // index into [10][Ch]
  if (i >= 10 * Ch) return;
  float s = 0.f;
  for (int k = 0; k < nchunks; ++k) s += part[(long long)k * 10 * Ch + i];
  const int tap = i / Ch, c = i - tap * Ch;
  if (tap < 9) gw[c * 9 + tap] += s; else gb[c] += s;
}

extern "C" int npvp_mlpdw_mid_bwd_reduce_into(const void* workspace, float* gw, float* gb, int frames, int Ch,
                                              hipStream_t stream) {
  NPVP_CHECK_ARG(workspace && gw && gb && frames > 0 && Ch > 0, "mlpdw_mid_bwd_reduce_into: bad arguments");
  const int chunks = mid_chunks(frames), fpc = (frames + chunks - 1) / chunks, nchunks = (frames + fpc - 1) / fpc;
  NPVP_LAUNCH(mid_bwd_reduce_into_kernel, dim3((10 * Ch + 255) / 256), dim3(256), 0, stream, (const float*)workspace, gw, gb,
                     Ch, nchunks);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

// the same as a 48-byte job record for npvp_sum_rows_multi (norm.hip: struct SumRowsJob, mode 1); nothing is launched
extern "C" int npvp_mlpdw_mid_bwd_reduce_job(const void* workspace, float* gw, float* gb, int frames, int Ch, void* job) {
  NPVP_CHECK_ARG(workspace && gw && gb && frames > 0 && Ch > 0 && job, "mlpdw_mid_bwd_reduce_job: bad arguments");
  const int chunks = mid_chunks(frames), fpc = (frames + chunks - 1) / chunks, nchunks = (frames + fpc - 1) / fpc;
  struct { const float* in; float* out; float* out_b; int nb, stride, ncols, split, accum, mode; } j =
      {(const float*)workspace, gw, gb, nchunks, 10 * Ch, 10 * Ch, Ch, 1, 1};
  static_assert(sizeof(j) == 48, "SumRowsJob");
  memcpy(job, &j, sizeof(j));
  return NPVP_OK;
}

extern "C" int npvp_mlpdw_mid_bwd_reduce(const void* workspace, float* dwt_db, int frames, int Ch, int accumulate,
                                         hipStream_t stream) {
  const int chunks = mid_chunks(frames), fpc = (frames + chunks - 1) / chunks, nchunks = (frames + fpc - 1) / fpc;
  const int rc = launch_sum_rows((const float*)workspace, dwt_db, nchunks, 10 * Ch, 10 * Ch, stream, accumulate ? 1 : 0);
  if (rc) { npvp_set_error("mlpdw_mid_bwd_reduce: launch failed"); return rc; }
  return NPVP_OK;
}

extern "C" int npvp_im2col3x3(const float* in, float* out, int frames, int H, int W, int C, int col2im, hipStream_t stream) {
  NPVP_CHECK_ARG(frames > 0 && H > 0 && W > 0 && C % 4 == 0, "im2col: bad shape");
  const long long total4 = (long long)frames * H * W * (col2im ? 1 : 9) * C / 4;
  long long blocks = (total4 + 255) / 256; if (blocks > 8192) blocks = 8192;
  NPVP_LAUNCH(im2col3x3_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in, out, H, W, C, total4, col2im);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

extern "C" long long npvp_dwconv3x3_wgrad_workspace_bytes(int frames, int Ch) {
  return (long long)dw_chunks(frames) * 10 * Ch * 4;
}

// dwt [9][Ch], db [Ch] must be CONTIGUOUS as one [10][Ch] buffer: db = dwt + 9*Ch
extern "C" int npvp_dwconv3x3_wgrad(const float* a, const float* dout, float* dwt_db, int frames, int H, int W, int Ch,
                                    void* workspace, long long ws_bytes, hipStream_t stream) {
  NPVP_CHECK_ARG(frames > 0 && H > 0 && W > 0 && Ch % 4 == 0, "dwconv_wgrad: bad shape");
  NPVP_CHECK_ARG(workspace && ws_bytes >= npvp_dwconv3x3_wgrad_workspace_bytes(frames, Ch), "dwconv_wgrad: workspace too small");
  const int chunks = dw_chunks(frames), fpc = (frames + chunks - 1) / chunks, nchunks = (frames + fpc - 1) / fpc;
  if (H == 8 && W == 8)
    NPVP_LAUNCH((dwconv3x3_wgrad_win_kernel<8, 8>), dim3((Ch / 4 + 255) / 256, nchunks), dim3(256), 0, stream, a, dout,
                       (float*)workspace, Ch, frames, fpc);
  else
    NPVP_LAUNCH(dwconv3x3_wgrad_kernel, dim3((Ch / 4 + 63) / 64, nchunks), dim3(256), 0, stream, a, dout,
                       (float*)workspace, H, W, Ch, frames, fpc);
  NPVP_CHECK_LAUNCH();
  const int rc = launch_sum_rows((const float*)workspace, dwt_db, nchunks, 10 * Ch, 10 * Ch, stream);
  if (rc) { npvp_set_error("dwconv_wgrad: reduce launch failed"); return rc; }
  return NPVP_OK;
}
