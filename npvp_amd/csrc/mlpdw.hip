// Depthwise 3x3 convolution of MlpDWBN (ref/models/VidHRFormer.py:351-358, nn.Conv2d(groups=Ch,
// padding=1), cross-correlation) over the channels-last hidden tensor [F, H*W, Ch], fwd, input
// gradient (same kernel, taps flipped) and weight/bias gradient.  HBM bound: one read + one write of
// the hidden tensor (the 9 neighbour taps hit L1/L2).  Weights are tap-major [9][Ch] so that a lane's
// 4 channels are one float4.
#include "common.h"

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
  const long long total4 = (long long)frames * H * W * Ch / 4;
  long long blocks = (total4 + 255) / 256; if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dwconv3x3_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, wt, bias, out, H, W, Ch, total4, flip);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

extern "C" int npvp_im2col3x3(const float* in, float* out, int frames, int H, int W, int C, int col2im, hipStream_t stream) {
  NPVP_CHECK_ARG(frames > 0 && H > 0 && W > 0 && C % 4 == 0, "im2col: bad shape");
  const long long total4 = (long long)frames * H * W * (col2im ? 1 : 9) * C / 4;
  long long blocks = (total4 + 255) / 256; if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(im2col3x3_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in, out, H, W, C, total4, col2im);
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
  hipLaunchKernelGGL(dwconv3x3_wgrad_kernel, dim3((Ch / 4 + 63) / 64, nchunks), dim3(256), 0, stream, a, dout,
                     (float*)workspace, H, W, Ch, frames, fpc);
  NPVP_CHECK_LAUNCH();
  const int rc = launch_sum_rows((const float*)workspace, dwt_db, nchunks, 10 * Ch, 10 * Ch, stream);
  if (rc) { npvp_set_error("dwconv_wgrad: reduce launch failed"); return rc; }
  return NPVP_OK;
}
