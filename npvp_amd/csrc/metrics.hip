// Evaluation metrics on device: what ref/utils/metrics.py computes per predicted frame with torch ops on
// whatever device the frames live on - PSNR / MSEScore (:12-43: a squared-difference reduction per image) and SSIM
// (:46-108: five 11x11 Gaussian-filtered maps by grouped conv2d, then a pointwise formula and a mean).
// Both are HBM bound: each image pair is read ONCE (the reference's SSIM reads img1/img2 five times and writes and
// re-reads nine intermediate maps), per-image results leave as one float each.  Algorithmic bytes: 8 B per pixel.
// Reductions are fixed-order (per-tile partials, then one thread per image sums them): results are run-to-run identical.
#include "common.h"

namespace npvp {

// partial[n][chunk] = sum over the chunk's elements of (x - y)^2, elements of image n = [n*per, (n+1)*per)
__global__ __launch_bounds__(256) void sqdiff_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             long long per, float inv_range, float* __restrict__ partial) {
  __shared__ float red[4];
  const long long n = blockIdx.y;
  const int chunks = gridDim.x;
  const long long chunk = (per + chunks - 1) / chunks, lo = blockIdx.x * chunk, hi = min(per, lo + chunk);
  const float* xp = x + n * per;
  const float* yp = y + n * per;
  float s = 0.f;
  if (((per | lo) & 3) == 0 && (((uintptr_t)xp | (uintptr_t)yp) & 15) == 0) {
    const long long hi4 = lo + ((hi - lo) & ~3ll);
    for (long long i = lo + threadIdx.x * 4ll; i < hi4; i += 1024) {
      const float4 a = ld4(xp + i), b = ld4(yp + i);
      const float d0 = (a.x - b.x) * inv_range, d1 = (a.y - b.y) * inv_range, d2 = (a.z - b.z) * inv_range, d3 = (a.w - b.w) * inv_range;
      s += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    for (long long i = hi4 + threadIdx.x; i < hi; i += 256) { const float d = (xp[i] - yp[i]) * inv_range; s += d * d; }
  } else {
    for (long long i = lo + threadIdx.x; i < hi; i += 256) { const float d = (xp[i] - yp[i]) * inv_range; s += d * d; }
  }
  s = block_sum<4>(s, red);
  if (threadIdx.x == 0) partial[n * chunks + blockIdx.x] = s;
}

// out[n] = scale * sum_j partial[n][j]  (fixed order)
__global__ void partial_sum_kernel(const float* __restrict__ partial, int J, float scale, float* __restrict__ out, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int j = 0; j < J; ++j) s += partial[(long long)n * J + j];
  out[n] = s * scale;
}

// ---- SSIM.  Block = one 32 x 32 output tile of one (image, channel) plane; WIN <= 11 taps (odd), zero padding.
struct SsimTaps { float w[11]; };
constexpr int ST = 32, SH = ST + 10, SLD = SH + 1;      // tile, tile + halo, LDS row stride of the halo tiles

template <int WIN>
__global__ __launch_bounds__(256) void ssim_tile_kernel(const float* __restrict__ a, const float* __restrict__ b, int H, int W,
                                                        int tiles_w, int tiles_per_plane, SsimTaps taps,
                                                        float* __restrict__ partial) {
  constexpr int R = WIN / 2, HH = ST + 2 * R;
  __shared__ float ta[SH * SLD], tb[SH * SLD];
  __shared__ float hz[5][SH][ST + 1];
  __shared__ float red[4];
  const long long plane = blockIdx.y;
  const int tile = blockIdx.x, ty = tile / tiles_w, tx = tile - ty * tiles_w;
  const int y0 = ty * ST - R, x0 = tx * ST - R;
  const float* ap = a + plane * H * W;
  const float* bp = b + plane * H * W;
  for (int i = threadIdx.x; i < HH * HH; i += 256) {
    const int r = i / HH, c = i - r * HH, y = y0 + r, x = x0 + c;
    const bool in = y >= 0 && y < H && x >= 0 && x < W;
    ta[r * SLD + c] = in ? ap[(long long)y * W + x] : 0.f;
    tb[r * SLD + c] = in ? bp[(long long)y * W + x] : 0.f;
  }
  __syncthreads();
  // horizontal pass: rows of the halo tile x ST columns
  for (int i = threadIdx.x; i < HH * ST; i += 256) {
    const int r = i / ST, c = i - r * ST;
    float s1 = 0.f, s2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
#pragma unroll
    for (int k = 0; k < WIN; ++k) {
      const float w = taps.w[k], u = ta[r * SLD + c + k], v = tb[r * SLD + c + k];
      s1 += w * u; s2 += w * v; s11 += w * (u * u); s22 += w * (v * v); s12 += w * (u * v);
    }
    hz[0][r][c] = s1; hz[1][r][c] = s2; hz[2][r][c] = s11; hz[3][r][c] = s22; hz[4][r][c] = s12;
  }
  __syncthreads();
  // vertical pass + the SSIM formula (ref/utils/metrics.py:89-104), 4 pixels per thread
  float acc = 0.f;
  for (int i = threadIdx.x; i < ST * ST; i += 256) {
    const int r = i / ST, c = i - r * ST;
    if (ty * ST + r >= H || tx * ST + c >= W) continue;
    float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k < WIN; ++k) {
      const float w = taps.w[k];
      m1 += w * hz[0][r + k][c]; m2 += w * hz[1][r + k][c]; e11 += w * hz[2][r + k][c]; e22 += w * hz[3][r + k][c];
      e12 += w * hz[4][r + k][c];
    }
    const float m1s = m1 * m1, m2s = m2 * m2, m12 = m1 * m2;
    const float v1 = e11 - m1s, v2 = e22 - m2s, cv = e12 - m12;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    acc += ((2.f * m12 + C1) * (2.f * cv + C2)) / ((m1s + m2s + C1) * (v1 + v2 + C2));
  }
  acc = block_sum<4>(acc, red);
  if (threadIdx.x == 0) partial[plane * tiles_per_plane + tile] = acc;
}

}  // namespace npvp

using namespace npvp;

static int sq_chunks(long long per) {
  long long c = (per + 16383) / 16384;          // >= 16 K elements per block
  return (int)(c < 1 ? 1 : c > 256 ? 256 : c);
}

extern "C" long long npvp_sqdiff_workspace_bytes(int N, long long per_image) { return (long long)N * sq_chunks(per_image) * 4; }

// out[n] = scale * sum_i ((x[n][i] - y[n][i]) / data_range)^2: PSNR takes scale = 1/per_image (the mean, then -10 log10 on the
// N results), MSEScore scale = 1 and data_range = 1.
extern "C" int npvp_sqdiff_per_image(const float* x, const float* y, int N, long long per_image, float data_range, float scale,
                                     float* out, void* workspace, long long ws_bytes, hipStream_t stream) {
  NPVP_CHECK_ARG(x && y && out && N > 0 && per_image > 0, "sqdiff_per_image: empty problem");
  NPVP_CHECK_ARG(data_range > 0.f, "sqdiff_per_image: data_range must be positive");
  NPVP_CHECK_ARG(workspace && ws_bytes >= npvp_sqdiff_workspace_bytes(N, per_image), "sqdiff_per_image: workspace too small");
  const int chunks = sq_chunks(per_image);
  NPVP_LAUNCH(sqdiff_partial_kernel, dim3(chunks, N), dim3(256), 0, stream, x, y, per_image, 1.f / data_range, (float*)workspace);
  NPVP_CHECK_LAUNCH();
  NPVP_LAUNCH(partial_sum_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, (const float*)workspace, chunks, scale, out, N);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

extern "C" long long npvp_ssim_workspace_bytes(int N, int C, int H, int W) {
  return (long long)N * C * ((H + ST - 1) / ST) * ((W + ST - 1) / ST) * 4;
}

// out[n] = mean over (C, H, W) of the SSIM map of image n; taps = the normalised 1-D Gaussian (window_size floats, odd,
// <= 11): the reference's 2-D window is its outer product (ref/utils/metrics.py:78-83)
extern "C" int npvp_ssim_per_image(const float* img1, const float* img2, int N, int C, int H, int W, const float* taps_host,
                                   int window_size, float* out, void* workspace, long long ws_bytes, hipStream_t stream) {
  NPVP_CHECK_ARG(img1 && img2 && out && N > 0 && C > 0 && H > 0 && W > 0, "ssim_per_image: empty problem");
  NPVP_CHECK_ARG(taps_host && (window_size == 11 || window_size == 7 || window_size == 5 || window_size == 3),
                 "ssim_per_image: window_size must be 3, 5, 7 or 11");
  NPVP_CHECK_ARG(workspace && ws_bytes >= npvp_ssim_workspace_bytes(N, C, H, W), "ssim_per_image: workspace too small");
  NPVP_CHECK_ARG((long long)N * C < 65536, "ssim_per_image: more than 65535 planes in one call");
  SsimTaps t = {};
  for (int i = 0; i < window_size; ++i) t.w[i] = taps_host[i];
  const int tw = (W + ST - 1) / ST, th = (H + ST - 1) / ST, tpp = tw * th;
  const dim3 grid(tpp, N * C), block(256);
  float* part = (float*)workspace;
  switch (window_size) {
    case 11: NPVP_LAUNCH(ssim_tile_kernel<11>, grid, block, 0, stream, img1, img2, H, W, tw, tpp, t, part); break;
    case 7: NPVP_LAUNCH(ssim_tile_kernel<7>, grid, block, 0, stream, img1, img2, H, W, tw, tpp, t, part); break;
    case 5: NPVP_LAUNCH(ssim_tile_kernel<5>, grid, block, 0, stream, img1, img2, H, W, tw, tpp, t, part); break;
    default: NPVP_LAUNCH(ssim_tile_kernel<3>, grid, block, 0, stream, img1, img2, H, W, tw, tpp, t, part); break;
  }
  NPVP_CHECK_LAUNCH();
  NPVP_LAUNCH(partial_sum_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, (const float*)part, C * tpp,
                     1.f / ((float)C * H * W), out, N);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}
