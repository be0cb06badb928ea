// Input side of the step (ref/utils/dataset.py:835-858 VidToTensor + VidNormalize): decoded frames travel to the
// device as the uint8 HWC images PIL produced (a quarter of the PCIe bytes of the reference's fp32 CHW tensors) and become the
// normalised fp32 (F, C, H, W) frames the encoder reads in ONE pass:  dst[f][c][y][x] = src[f][y][x][c] * scale[c] + shift[c]
// with scale = 1 / (255 std), shift = -mean / std.  HBM bound: 1 B read + 4 B written per sample.
#include "common.h"
#include <stdint.h>

namespace npvp {

struct ChanAffine { float scale[4], shift[4]; };

// block = 256 consecutive pixels of one frame; C <= 4 channels interleaved in the source
template <int C>
__global__ __launch_bounds__(256) void u8hwc_to_f32chw_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                                              int HW, ChanAffine a) {
  const long long f = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const uint8_t* s = src + (f * HW + p) * C;
  float* d = dst + f * C * (long long)HW + p;
#pragma unroll
  for (int c = 0; c < C; ++c) d[(long long)c * HW] = (float)s[c] * a.scale[c] + a.shift[c];
}

}  // namespace npvp

using namespace npvp;

// mean / std: HOST arrays of C floats (torchvision Normalize semantics on ToTensor()'s [0,1] values)
extern "C" int npvp_u8hwc_to_f32chw(const void* src, float* dst, long long frames, int H, int W, int C, const float* mean,
                                    const float* std, hipStream_t stream) {
  NPVP_CHECK_ARG(src && dst && frames > 0 && H > 0 && W > 0, "u8hwc_to_f32chw: empty problem");
  NPVP_CHECK_ARG(C == 1 || C == 3 || C == 4, "u8hwc_to_f32chw: 1, 3 or 4 channels");
  NPVP_CHECK_ARG(mean && std, "u8hwc_to_f32chw: mean / std needed");
  NPVP_CHECK_ARG(frames < 65536, "u8hwc_to_f32chw: at most 65535 frames per call");
  ChanAffine a = {};
  for (int c = 0; c < C; ++c) {
    NPVP_CHECK_ARG(std[c] != 0.f, "u8hwc_to_f32chw: zero std");
    a.scale[c] = 1.f / (255.f * std[c]); a.shift[c] = -mean[c] / std[c];
  }
  const int HW = H * W;
  const dim3 grid((HW + 255) / 256, (unsigned)frames), block(256);
  if (C == 1) NPVP_LAUNCH(u8hwc_to_f32chw_kernel<1>, grid, block, 0, stream, (const uint8_t*)src, dst, HW, a);
  else if (C == 3) NPVP_LAUNCH(u8hwc_to_f32chw_kernel<3>, grid, block, 0, stream, (const uint8_t*)src, dst, HW, a);
  else NPVP_LAUNCH(u8hwc_to_f32chw_kernel<4>, grid, block, 0, stream, (const uint8_t*)src, dst, HW, a);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}
