// Small HBM-bound helpers of the predictor step: dropout / drop-path scaling of a gradient
// (backward of the fused GEMM epilogues), batched [R x C] <-> [C x R] transposes between the
// reference's (N,T,C,H,W) tensors and the canonical (N*T, H*W, C) layout
// (ref/models/VidHRFormer.py:34,50,137-138,159), mid-axis reductions (event-coding mean over T,
// ref/models/Predictor.py:346; sums over the batch for the positional-table gradients), the
// decoder-only gradient-norm clip and AdamW update of the training step
// (ref/models/Predictor.py:135-136,197).
#include "common.h"

namespace npvp {

__global__ void drop_apply_kernel(const float* __restrict__ x, float* __restrict__ out, long long rows, int ncols,
                                  DropSpec d, const unsigned long long* __restrict__ seedp, float* __restrict__ amax) {
  const unsigned long long seed = *seedp;
  const int c4n = ncols / 4;
  __shared__ float ared[16];
  const unsigned int peek = amax_peek_block(amax);
  float am = 0.f;
  const long long total4 = rows * c4n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / c4n;
    const int col = (int)(i - row * c4n) * 4;
    float4 v = ld4(x + row * ncols + col);
    if (d.mode == 0) {
      v.x *= drop_spec_scale(d, seed, row, col + 0, ncols);
      v.y *= drop_spec_scale(d, seed, row, col + 1, ncols);
      v.z *= drop_spec_scale(d, seed, row, col + 2, ncols);
      v.w *= drop_spec_scale(d, seed, row, col + 3, ncols);
    } else {
      const float s = drop_spec_scale(d, seed, row, col, ncols);
      v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    }
    st4(out + row * ncols + col, v);
    am = amax4(am, v);
  }
  amax_slot_commit_block(amax, am, ared, peek);
}

// in [B][R][C] -> out [B][C][R], 32x32 LDS tiles, 256 threads (32 x 8)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
  __shared__ float tile[32][33];
  const long long b = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* ib = in + b * R * C;
  float* ob = out + b * R * C;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    if (r < R && c < C) tile[ty + 8 * k][tx] = ib[(long long)r * C + c];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (r < R && c < C) ob[(long long)c * R + r] = tile[tx][ty + 8 * k];
  }
}

// out[a][c] = scale * sum_b in[a][b][c]
__global__ void reduce_mid_kernel(const float* __restrict__ in, float* __restrict__ out, int A, int B, long long Cc,
                                  float scale, int accum) {
  const long long c4n = Cc / 4, total4 = (long long)A * c4n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const long long a = i / c4n, c = (i - a * c4n) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* p = in + a * B * Cc + c;
    for (int b = 0; b < B; ++b) {
      const float4 v = ld4(p + (long long)b * Cc);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    s.x *= scale; s.y *= scale; s.z *= scale; s.w *= scale;
    if (accum) { const float4 o = ld4(out + a * Cc + c); s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    st4(out + a * Cc + c, s);
  }
}

// out[a][b][c] = scale * in[a][c]   (backward of the mean over T)
__global__ void broadcast_mid_kernel(const float* __restrict__ in, float* __restrict__ out, int A, int B, long long Cc,
                                     float scale) {
  const long long c4n = Cc / 4, total4 = (long long)A * B * c4n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const long long ab = i / c4n, c = (i - ab * c4n) * 4, a = ab / B;
    float4 v = ld4(in + a * Cc + c);
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    st4(out + ab * Cc + c, v);
  }
}

// part[chunk][n] = sum over the chunk's rows of x[row][n]   (bias gradients: ref nn.Linear / Conv2d bias)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                             long long rows, int N, long long ld, int rows_per_chunk) {
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (c >= N) return;
  const long long r0 = (long long)blockIdx.y * rows_per_chunk;
  const long long r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long long r = r0; r < r1; ++r) {
    const float4 v = ld4(x + r * ld + c);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  st4(part + (long long)blockIdx.y * N + c, s);
}

// ---- gradient-norm clip + AdamW on flat fp32 buffers
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long long n, float* __restrict__ part) {
  __shared__ float red[4];
  float s = 0.f;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i + 3 < n; i += (long long)gridDim.x * blockDim.x * 4) {
    const float4 v = ld4(g + i);
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) for (long long i = n & ~3ll; i < n; ++i) s += g[i] * g[i];
  s = block_sum<4>(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// out[0] = total L2 norm, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))  (torch clip_grad_norm_)
__global__ void clip_coef_kernel(const float* __restrict__ part, int nb, float max_norm, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
  s = block_sum<4>(s, red);
  if (threadIdx.x == 0) {
    const float nrm = sqrtf(s);
    out[0] = nrm;
    out[1] = fminf(1.f, max_norm / (nrm + 1e-6f));
  }
}

// torch.optim.AdamW (decoupled weight decay, bias-corrected), hyper = {lr, step} in device memory so a
// captured graph sees the scheduler's value; elements in [clip_begin, clip_end) are scaled by clip[1].
__device__ __forceinline__ void adamw_one(float& pi, float& gi, float& mi, float& vi, const bool clipped, const float coef,
                                          const float decay, const float beta1, const float beta2, const float step_size,
                                          const float inv_sqrt_bc2, const float eps) {
  if (clipped) gi *= coef;
  pi *= decay;
  mi = beta1 * mi + (1.f - beta1) * gi;
  vi = beta2 * vi + (1.f - beta2) * gi * gi;
  pi -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
}
// Four elements per thread and trip (float4 loads / stores of p, g, m, v: 28 - 32 bytes of traffic per element is all this kernel
// is; the scalar form ran at 2.7 TB/s, 617 us of an 8-clip step); the arithmetic per element is the scalar form's, operation for
// operation.  n4 = n / 4 vector elements (the buffers are allocations: 256-byte aligned), the last n % 4 by one thread.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long long n, const float* __restrict__ hyper, float beta1, float beta2, float eps, float wd,
                             const float* __restrict__ clip, long long clip_begin, long long clip_end, int write_back_grad) {
  const float lr = hyper[0], step = hyper[1];
  const float bc1 = 1.f - powf(beta1, step), bc2 = 1.f - powf(beta2, step);
  const float coef = clip ? clip[1] : 1.f;
  const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2), decay = 1.f - lr * wd;
  const long long n4 = n >> 2;
  for (long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i4 < n4; i4 += (long long)gridDim.x * blockDim.x) {
    const long long i = i4 << 2;
    float4 pp = ld4(p + i), gg = ld4(g + i), mm = ld4(m + i), vv = ld4(v + i);
    const bool c0 = i >= clip_begin && i < clip_end, c1 = i + 1 >= clip_begin && i + 1 < clip_end;
    const bool c2 = i + 2 >= clip_begin && i + 2 < clip_end, c3 = i + 3 >= clip_begin && i + 3 < clip_end;
    adamw_one(pp.x, gg.x, mm.x, vv.x, c0, coef, decay, beta1, beta2, step_size, inv_sqrt_bc2, eps);
    adamw_one(pp.y, gg.y, mm.y, vv.y, c1, coef, decay, beta1, beta2, step_size, inv_sqrt_bc2, eps);
    adamw_one(pp.z, gg.z, mm.z, vv.z, c2, coef, decay, beta1, beta2, step_size, inv_sqrt_bc2, eps);
    adamw_one(pp.w, gg.w, mm.w, vv.w, c3, coef, decay, beta1, beta2, step_size, inv_sqrt_bc2, eps);
    if (write_back_grad && (c0 || c1 || c2 || c3)) st4(g + i, gg);
    st4(m + i, mm); st4(v + i, vv); st4(p + i, pp);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long long i = n4 << 2; i < n; ++i) {
      float pi = p[i], gi = g[i], mi = m[i], vi = v[i];
      const bool c = i >= clip_begin && i < clip_end;
      adamw_one(pi, gi, mi, vi, c, coef, decay, beta1, beta2, step_size, inv_sqrt_bc2, eps);
      if (write_back_grad && c) g[i] = gi;
      m[i] = mi; v[i] = vi; p[i] = pi;
    }
}

static inline int ew_blocks(long long total, int threads) {
  long long b = (total + threads - 1) / threads;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace npvp

using namespace npvp;

extern "C" int npvp_drop_apply(const float* x, float* out, long long rows, int ncols, float p, int mode, int g1, int g2,
                               const unsigned long long* seed, unsigned int salt, float* amax, hipStream_t stream) {
  NPVP_CHECK_ARG(rows > 0 && ncols % 4 == 0, "drop_apply: bad shape");
  NPVP_CHECK_ARG(p > 0.f && p < 1.f && seed, "drop_apply: needs 0 < p < 1 and a device seed");
  const DropSpec d = make_drop_spec(p, salt, mode, g1, g2);
  NPVP_LAUNCH(drop_apply_kernel, dim3(ew_blocks(rows * (ncols / 4), 256)), dim3(256), 0, stream, x, out, rows, ncols,
                     d, seed, amax);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

extern "C" int npvp_transpose(const float* in, float* out, int batch, int R, int C, hipStream_t stream) {
  NPVP_CHECK_ARG(batch > 0 && R > 0 && C > 0 && batch <= 65535, "transpose: bad shape");
  NPVP_LAUNCH(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32, batch), dim3(256), 0, stream, in, out, R, C);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

// gradient of the depthwise 3x3 parameters out of the tap-major table the fused MlpDWBN backward produces: gw [C][9] += dwtb
// [9][C] transposed, gb [C] += dwtb[9][:]  (one launch instead of a transpose and autograd's two accumulate adds)
__global__ void dwtb_accumulate_kernel(const float* __restrict__ dwtb, float* __restrict__ gw, float* __restrict__ gb, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 9 * C) {
    const int c = i / 9, tap = i - 9 * c;
    gw[i] += dwtb[tap * C + c];
  } else if (i < 10 * C) {
    gb[i - 9 * C] += dwtb[i];
  }
}

// the inverse direction, forward: nn.Conv2d(C, C, 3, groups=C) parameters -> the tap-major table [10][C] the depthwise kernels read
// (one launch for a transpose and a row copy)
__global__ void dwtb_build_kernel(const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ wtb, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 9 * C) {
    const int tap = i / C, c = i - tap * C;
    wtb[i] = w[c * 9 + tap];
  } else if (i < 10 * C) {
    wtb[i] = b ? b[i - 9 * C] : 0.f;
  }
}

extern "C" int npvp_dwtb_build(const float* w, const float* b, float* wtb, int C, hipStream_t stream) {
  NPVP_CHECK_ARG(C > 0 && w && wtb, "dwtb_build: bad arguments");
  NPVP_LAUNCH(dwtb_build_kernel, dim3((10 * C + 255) / 256), dim3(256), 0, stream, w, b, wtb, C);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

extern "C" int npvp_dwtb_accumulate(const float* dwtb, float* gw, float* gb, int C, hipStream_t stream) {
  NPVP_CHECK_ARG(C > 0 && dwtb && gw && gb, "dwtb_accumulate: bad arguments");
  NPVP_LAUNCH(dwtb_accumulate_kernel, dim3((10 * C + 255) / 256), dim3(256), 0, stream, dwtb, gw, gb, C);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

int npvp_reduce_mid_launch(const float* in, float* out, int A, int B, long long Cc, float scale, hipStream_t stream, int accumulate) {
  NPVP_CHECK_ARG(A > 0 && B > 0 && Cc > 0 && Cc % 4 == 0, "reduce_mid: bad shape");
  NPVP_LAUNCH(reduce_mid_kernel, dim3(ew_blocks((long long)A * Cc / 4, 256)), dim3(256), 0, stream, in, out, A, B, Cc,
                     scale, accumulate ? 1 : 0);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

extern "C" int npvp_reduce_mid(const float* in, float* out, int A, int B, long long Cc, float scale, int accumulate,
                               hipStream_t stream) {
  return npvp_reduce_mid_launch(in, out, A, B, Cc, scale, stream, accumulate);
}

extern "C" int npvp_broadcast_mid(const float* in, float* out, int A, int B, long long Cc, float scale, hipStream_t stream) {
  NPVP_CHECK_ARG(A > 0 && B > 0 && Cc > 0 && Cc % 4 == 0, "broadcast_mid: bad shape");
  NPVP_LAUNCH(broadcast_mid_kernel, dim3(ew_blocks((long long)A * B * Cc / 4, 256)), dim3(256), 0, stream, in, out, A,
                     B, Cc, scale);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

static int colsum_chunks(long long rows) { return (int)(rows < 128 ? rows : 128); }

extern "C" long long npvp_colsum_workspace_bytes(long long rows, int N) { return (long long)colsum_chunks(rows) * N * 4; }

// out[n] = sum_r x[r][n]
extern "C" int npvp_colsum(const float* x, long long rows, int N, long long ld, float* out, int accumulate, void* workspace,
                           long long ws_bytes, hipStream_t stream) {
  NPVP_CHECK_ARG(rows > 0 && N > 0 && N % 4 == 0 && ld % 4 == 0, "colsum: bad shape");
  NPVP_CHECK_ARG(workspace && ws_bytes >= npvp_colsum_workspace_bytes(rows, N), "colsum: workspace too small");
  const int chunks = colsum_chunks(rows);
  const int rpc = (int)((rows + chunks - 1) / chunks), nchunks = (int)((rows + rpc - 1) / rpc);
  NPVP_LAUNCH(colsum_partial_kernel, dim3((N / 4 + 255) / 256, nchunks), dim3(256), 0, stream, x, (float*)workspace,
                     rows, N, ld, rpc);
  NPVP_CHECK_LAUNCH();
  const int rc = launch_sum_rows((const float*)workspace, out, nchunks, N, N, stream, accumulate);
  if (rc) { npvp_set_error("colsum: reduce launch failed"); return rc; }
  return NPVP_OK;
}

namespace npvp {
// ---- scalar losses of the step (ref/models/criterion.py:99-121 L1Loss, :341-354 Div_KL): deterministic two-stage sums, fixed
// order, no atomics and no semaphore.  torch's multi-block reductions zero a 4-byte semaphore with a memset NODE when they are
// captured; memset nodes are what the ROCm 7.2 prepared-packet replay does not execute reliably (profiles/r06_graph_alloc_hazard.txt), so
// the captured step contains none.
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                                         float* __restrict__ part) {
  __shared__ float red[4];
  float s = 0.f;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i + 3 < n; i += (long long)gridDim.x * blockDim.x * 4) {
    const float4 u = ld4(a + i), v = ld4(b + i);
    s += (fabsf(u.x - v.x) + fabsf(u.y - v.y)) + (fabsf(u.z - v.z) + fabsf(u.w - v.w));
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) for (long long i = n & ~3ll; i < n; ++i) s += fabsf(a[i] - b[i]);
  s = block_sum<4>(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void sum_partial_kernel(const float* __restrict__ x, long long n, float* __restrict__ part) {
  __shared__ float red[4];
  float s = 0.f;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i + 3 < n; i += (long long)gridDim.x * blockDim.x * 4) {
    const float4 v = ld4(x + i);
    s += (v.x + v.y) + (v.z + v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) for (long long i = n & ~3ll; i < n; ++i) s += x[i];
  s = block_sum<4>(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// out[0] = (sum of the partials / div) * mul
__global__ void sum_finish_kernel(const float* __restrict__ part, int nb, float div, float mul, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
  s = block_sum<4>(s, red);
  if (threadIdx.x == 0) out[0] = (s / div) * mul;
}

// da = sgn(a - b) * ((gout * lam) / n): the gradient torch's abs -> mean -> mul chain hands to `a`, in its order of operations
__global__ __launch_bounds__(256) void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                                     const float* __restrict__ gout, float lam, float* __restrict__ da) {
  const float c = (gout[0] * lam) / (float)n;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i + 3 < n; i += (long long)gridDim.x * blockDim.x * 4) {
    const float4 u = ld4(a + i), v = ld4(b + i);
    float4 r;
    r.x = c * (float)((u.x > v.x) - (u.x < v.x)); r.y = c * (float)((u.y > v.y) - (u.y < v.y));
    r.z = c * (float)((u.z > v.z) - (u.z < v.z)); r.w = c * (float)((u.w > v.w) - (u.w < v.w));
    st4(da + i, r);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) for (long long i = n & ~3ll; i < n; ++i) da[i] = c * (float)((a[i] > b[i]) - (a[i] < b[i]));
}

}  // namespace npvp

static long long loss_blocks(long long n) { long long nb = (n / 4 + 255) / 256; if (nb > 1024) nb = 1024; if (nb < 1) nb = 1; return nb; }

// out[0] = lam * mean |a - b|; workspace >= 1024 floats
extern "C" int npvp_l1_mean(const float* a, const float* b, long long n, float lam, float* out, void* workspace, long long ws_bytes,
                            hipStream_t stream) {
  NPVP_CHECK_ARG(a && b && out && n > 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0, "l1_mean: bad buffers");
  NPVP_CHECK_ARG(workspace && ws_bytes >= 1024 * 4, "l1_mean: workspace too small");
  const long long nb = loss_blocks(n);
  NPVP_LAUNCH(l1_partial_kernel, dim3((unsigned)nb), dim3(256), 0, stream, a, b, n, (float*)workspace);
  NPVP_CHECK_LAUNCH();
  NPVP_LAUNCH(sum_finish_kernel, dim3(1), dim3(256), 0, stream, (const float*)workspace, (int)nb, (float)n, lam, out);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

extern "C" int npvp_l1_mean_bwd(const float* a, const float* b, long long n, const float* gout, float lam, float* da, hipStream_t stream) {
  NPVP_CHECK_ARG(a && b && gout && da && n > 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0 && ((uintptr_t)da % 16) == 0,
                 "l1_mean_bwd: bad buffers");
  long long nb = (n / 4 + 255) / 256; if (nb > 4096) nb = 4096; if (nb < 1) nb = 1;
  NPVP_LAUNCH(l1_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, stream, a, b, n, gout, lam, da);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

// out[0] = sum x; workspace >= 1024 floats
extern "C" int npvp_sum_all(const float* x, long long n, float* out, void* workspace, long long ws_bytes, hipStream_t stream) {
  NPVP_CHECK_ARG(x && out && n > 0 && ((uintptr_t)x % 16) == 0, "sum_all: bad buffer");
  NPVP_CHECK_ARG(workspace && ws_bytes >= 1024 * 4, "sum_all: workspace too small");
  const long long nb = loss_blocks(n);
  NPVP_LAUNCH(sum_partial_kernel, dim3((unsigned)nb), dim3(256), 0, stream, x, n, (float*)workspace);
  NPVP_CHECK_LAUNCH();
  NPVP_LAUNCH(sum_finish_kernel, dim3(1), dim3(256), 0, stream, (const float*)workspace, (int)nb, 1.f, 1.f, out);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

// out2 = {norm, clip coefficient}; workspace >= 1024 floats
extern "C" int npvp_grad_norm_clip(const float* g, long long n, float max_norm, float* out2, void* workspace,
                                   long long ws_bytes, hipStream_t stream) {
  NPVP_CHECK_ARG(n > 0 && ((uintptr_t)g % 16) == 0, "grad_norm_clip: bad buffer");
  NPVP_CHECK_ARG(workspace && ws_bytes >= 1024 * 4, "grad_norm_clip: workspace too small");
  long long nb = (n / 4 + 255) / 256; if (nb > 1024) nb = 1024; if (nb < 1) nb = 1;
  NPVP_LAUNCH(sumsq_partial_kernel, dim3((unsigned)nb), dim3(256), 0, stream, g, n, (float*)workspace);
  NPVP_CHECK_LAUNCH();
  NPVP_LAUNCH(clip_coef_kernel, dim3(1), dim3(256), 0, stream, (const float*)workspace, (int)nb, max_norm, out2);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

extern "C" int npvp_adamw_step(float* p, float* g, float* m, float* v, long long n, const float* hyper, float beta1,
                               float beta2, float eps, float weight_decay, const float* clip, long long clip_begin,
                               long long clip_end, int write_back_grad, hipStream_t stream) {
  NPVP_CHECK_ARG(n > 0 && hyper, "adamw: bad arguments");
  NPVP_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adamw: the flat buffers must be 16-byte aligned");
  NPVP_LAUNCH(adamw_kernel, dim3(ew_blocks((n + 3) / 4, 256)), dim3(256), 0, stream, p, g, m, v, n, hyper, beta1, beta2, eps,
                     weight_decay, clip, clip_begin, clip_end, write_back_grad);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}
