// Epilogues of the frozen Stage-1 autoencoder around the predictor (SURVEY 8f #1 stage 2; ref/models/ResNetAutoEncoder.py:
// 51-261, ref/models/submodules.py:9-95).  The autoencoder is frozen and in eval mode in Stage 2, so every
// conv -> BatchNorm(running statistics) -> ReLU triple is ONE convolution with folded weights (host side, once) followed by
// ONE pass  out = act(conv + bias[c]) (+ skip)  instead of the reference's bias-add, BatchNorm and ReLU (and skip-add) passes
// over the activation.  The convolutions themselves stay MIOpen's.  HBM bound: 8 B per element (+ 4 with a skip).
//   layout 0: x [outer][C], channel = column (torch.channels_last memory of an (N,C,H,W) tensor: outer = N*H*W)
//   layout 1: x [outer][inner], channel = outer % C (contiguous NCHW: outer = N*C planes of inner = H*W)
//   act: 0 none, 1 ReLU, 2 tanh, 3 sigmoid
#include "common.h"

namespace npvp {

__device__ __forceinline__ float ae_act(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return tanhf(v);
  if (act == 3) return 1.f / (1.f + __expf(-v));
  return v;
}

template <bool VEC4>
__global__ void bias_act_kernel(const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ res,
                                float* __restrict__ out, long long outer, long long inner, int C, int layout, int act) {
  const long long n = outer * inner;
  if (VEC4) {
    const long long n4 = n / 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
      const long long e = i * 4;
      float4 v = ld4(x + e);
      float b0, b1, b2, b3;
      if (layout == 0) { const int c = (int)(e % inner); b0 = bias[c]; b1 = bias[c + 1]; b2 = bias[c + 2]; b3 = bias[c + 3]; }
      else { b0 = b1 = b2 = b3 = bias[(int)((e / inner) % C)]; }
      v.x = ae_act(v.x + b0, act); v.y = ae_act(v.y + b1, act); v.z = ae_act(v.z + b2, act); v.w = ae_act(v.w + b3, act);
      if (res) { const float4 r = ld4(res + e); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
      st4(out + e, v);
    }
  } else {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
      const int c = layout == 0 ? (int)(e % inner) : (int)((e / inner) % C);
      float v = ae_act(x[e] + bias[c], act);
      if (res) v += res[e];
      out[e] = v;
    }
  }
}

// dx = g * act'(.) expressed through the forward OUTPUT y (no skip): ReLU y > 0, tanh 1 - y^2, sigmoid y (1 - y)
__global__ void act_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ dx, long long n,
                               int act) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    const float yy = y[e];
    const float d = act == 1 ? (yy > 0.f ? 1.f : 0.f) : act == 2 ? 1.f - yy * yy : act == 3 ? yy * (1.f - yy) : 1.f;
    dx[e] = g[e] * d;
  }
}

static inline int ae_blocks(long long work) {
  long long b = (work + 255) / 256;
  return (int)(b < 1 ? 1 : b > 16384 ? 16384 : b);
}

}  // namespace npvp

using namespace npvp;

extern "C" int npvp_bias_act(const float* x, const float* bias, const float* residual, float* out, long long outer,
                             long long inner, int C, int layout, int act, hipStream_t stream) {
  NPVP_CHECK_ARG(x && bias && out && outer > 0 && inner > 0 && C > 0, "bias_act: empty problem");
  NPVP_CHECK_ARG(layout == 0 || layout == 1, "bias_act: layout 0 (channel = column) or 1 (channel = plane)");
  NPVP_CHECK_ARG(layout == 1 || inner == C, "bias_act: layout 0 needs inner == C");
  NPVP_CHECK_ARG(act >= 0 && act <= 3, "bias_act: act in 0..3");
  const bool vec = inner % 4 == 0 && (((uintptr_t)x | (uintptr_t)out | (uintptr_t)residual) & 15) == 0;
  if (vec) NPVP_LAUNCH(bias_act_kernel<true>, dim3(ae_blocks(outer * inner / 4)), dim3(256), 0, stream, x, bias, residual, out, outer, inner, C, layout, act);
  else NPVP_LAUNCH(bias_act_kernel<false>, dim3(ae_blocks(outer * inner)), dim3(256), 0, stream, x, bias, residual, out, outer, inner, C, layout, act);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

extern "C" int npvp_act_bwd(const float* g, const float* y, float* dx, long long n, int act, hipStream_t stream) {
  NPVP_CHECK_ARG(g && y && dx && n > 0, "act_bwd: empty problem");
  NPVP_CHECK_ARG(act >= 0 && act <= 3, "act_bwd: act in 0..3");
  NPVP_LAUNCH(act_bwd_kernel, dim3(ae_blocks(n)), dim3(256), 0, stream, g, y, dx, n, act);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}
