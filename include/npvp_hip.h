/* libnpvp_hip.so - C ABI of the MI355X (gfx950) kernels behind NPVP's Stage-2 predictor
 * hot path.  The reference has NO native boundary for this path: it is pure Python that
 * reaches ATen through torch.nn modules ("zero CUDA kernels, zero C++").  The
 * entry points below are what a reference-side binding replaces those torch calls with;
 * each one names the reference call sites whose arithmetic it takes over.  INTEGRATION.md
 * shows the ctypes stub a maintainer adds to ref/models/VidHRFormer.py.
 *
 * Conventions (all entry points)
 *   - plain pointers + sizes, fp32, row-major, no torch types; every pointer is DEVICE memory
 *     owned and allocated by the caller (including workspaces and saved-for-backward tensors);
 *   - the library never allocates, frees, synchronises or keeps global mutable state (but a launch counter and the
 *     thread-local error string); all work
 *     is enqueued on `stream` (a hipStream_t), so calls are graph-capturable and re-entrant;
 *   - return 0 on success, <0 on error (-1 bad argument, -2 launch failure, -3 workspace);
 *     npvp_last_error() gives the thread-local message; nothing throws;
 *   - canonical activation layout: x[F][P][C], F = N*T frames (f = n*T + t), P = H*W, C contiguous;
 *   - dropout / drop-path masks are a counter hash of (*seed, salt, element key): `seed` is a
 *     device uint64 (so a captured graph sees a new value per replay), `salt` identifies the call
 *     site; backward entry points replay the forward mask from the same (seed, salt).
 *
 * Where this header differs from SURVEY 8(b)'s sketch it is authoritative:
 *   - the data-parallel exchange (`npvp_dp_*`, at the end of this header) is the one part of the library WITH process state (one
 *     RCCL communicator per process); RCCL is looked up at the first npvp_dp_* call, not linked.  npvp_amd/dp.py drives either
 *     these entry points (NPVP_DP_COMM=c) or torch.distributed's ProcessGroupNCCL (default - the same RCCL; a PyTorch host
 *     already owns a communicator set, SyncBatchNorm's statistics travel on it either way).
 *   - `npvp_<op>_workspace_bytes` exists for the entry points that NEED a workspace (GEMM split-K, LayerNorm / frame-LN /
 *     fused-middle parameter-gradient partials, colsum, metrics); the others take none.
 */
#ifndef NPVP_HIP_H
#define NPVP_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* npvp_stream_t; /* hipStream_t */

int npvp_version(void);
const char* npvp_last_error(void);
/* Diagnostics: kernels this library has launched in this process so far (one relaxed atomic add per launch).  bench.py reads it
 * around a step so that its record says how many launches a step is. */
long long npvp_launch_count(void);
/* A stream of the lowest priority the device offers, for work nothing on the critical path waits for (the in-place
 * weight-gradient writes); least / greatest receive the device's priority range (nullable). */
void* npvp_stream_create_low_priority(int* least, int* greatest);
int npvp_stream_destroy(void* stream);
/* Timing events that survive a HIP-graph capture (the benchmark's live per-kernel probe; the reference has no counterpart - its
 * timings are time.time() around a forward, ref/Inference.ipynb:463-467).  npvp_event_record on a CAPTURING stream becomes an
 * external event-record node (hipEventRecordWithFlags / hipEventRecordExternal): every replay stamps the event, and after a
 * synchronisation npvp_event_elapsed_ms(e0, e1) is the time between the two stamps of the LAST replay; on an ordinary stream it is a
 * plain hipEventRecord.  elapsed < 0: an event was never recorded or has not completed. */
void* npvp_event_create(void);
int npvp_event_record(void* event, void* stream);
float npvp_event_elapsed_ms(void* event0, void* event1);
int npvp_event_destroy(void* event);

/* Node census of a captured step (a hipGraph_t, e.g. torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()): counts[16] indexed by
 * hipGraphNodeType (0 kernel, 1 memcpy, 2 memset, ...), memset_bytes[0 .. max_memsets) the sizes of the first memset nodes; returns
 * the number of nodes or a negative error.  The training step of this library contains no memset node, and the host layer checks
 * it: on ROCm 7.2 a graph replayed from prepared packets (the runtime's default) does not execute its memset nodes reliably - stale fill
 * patterns once the process has issued other memsets (tools/graph_memset_node_repro.py, profiles/r06_graph_alloc_hazard.txt). */
long long npvp_graph_node_counts(void* graph, long long* counts, long long* memset_bytes, int max_memsets);

/* ---- GEMM (every nn.Linear / 1x1 Conv2d / MHA in- and out-projection and their backward:
 * ref/models/VidHRFormer.py:71-72,111,184-185,225 (token FFN), :345,364,380,387 (MlpDWBN fc1/fc2),
 * torch.nn.MultiheadAttention in_proj/out_proj at :70,180,192,270; NRMLP linears
 * ref/models/submodules.py:275,288,295).
 *   C[M,N] = epilogue(alpha * op(A)[M,K] op(B)[K,N])
 *   a_kc=1: A is [M][K]; a_kc=0: A is [K][M].   b_kc=1: B is [N][K]; b_kc=0: B is [K][N].
 *   epilogue order: +bias[N] -> aux_out (pre-activation copy) -> act -> dropout -> +residual
 *   act: 0 none, 1 GELU(erf), 2 ReLU, 3 multiply by GELU'(aux_in), 4 multiply by [aux_in > 0]
 *   drop_mode 0: per element; 1: per row group key=(row/drop_g1)%drop_g2 (DropPath)
 * K % 32 == 0, M % 4 == 0, N % 4 == 0, lda/ldb % 4 == 0, A/B 16-byte aligned.
 * precision 0: exact fp32-input MFMA (v_mfma_f32_32x32x2_f32; parity triage).  precision 4: every fp32 operand is
 * split into three bf16 terms and the six leading cross products are accumulated in fp32 on v_mfma_f32_32x32x16_bf16
 * (relative product error ~2^-23: fp32-grade).  precision 5: two terms, three products (~2^-16; opt-in for weight gradients).
 * precision 6 (the path's default): TWO fp16 terms per operand, three products on v_mfma_f32_32x32x16_f16 (~2^-22 per
 * product: fp32-grade at half the matrix instructions of precision 4).  fp16 has no exponent range to spare, so each operand
 * comes with an AMAX SLOT (2 KB of device memory holding 32 words whose maximum bounds |operand|; see npvp_amax) and is scaled by the power
 * of two that puts the bound into [2^14, 2^15) - exact, undone in the epilogue.  It is taken by (a) a_kc = 1 launches with
 * fp16 planes in `b_pre` (npvp_split_weight_f16) and both a_amax / b_amax, (b) weight gradients (a_kc = b_kc = 0, plain
 * epilogue, K >= 1024) with both slots; every other launch with precision 6 runs as precision 4 without planes.
 * c_amax (nullable, any precision, unsplit launches): the kernel adds the bound of the values it stores to C to that slot -
 * the next GEMM's a_amax, at no extra pass.
 * Kernels: gemm_wide_kernel (128 x 256 tiles, 4 waves, A split on the fly, B = pre-split planes `b_pre` copied by LDS-DMA:
 * the large forward / dgrad shapes), gemm_wgrad_wide_kernel (weight gradients over >= 32 K token rows: row-major staging,
 * transposing LDS reads, split-K), gemm_split_db_kernel (128 x 128 tiles: everything else).
 * colsum_a (a_kc = 0 only, nullable): receives colsum_a[m] = sum_k A[k][m] - the bias gradient falls out of the
 * weight-gradient GEMM's own operand staging (dW = dy^T x, db = column sums of dy), no extra pass over dy.
 * accumulate = 1: C += result and colsum_a += sums - a weight / bias gradient is accumulated straight into the live
 * .grad slice of the flat gradient buffer (no temporary, no separate add kernel); the same flag exists on the other
 * entry points that produce parameter gradients (npvp_layernorm_bwd, npvp_frameln_act_bwd, npvp_colsum).
 * When the tile count is small and K large (weight gradients) the reduction is split over
 * workgroups through `workspace` (npvp_gemm_workspace_bytes; 0 = never split). */
long long npvp_gemm_workspace_bytes(int M, int N, int K);
/* which kernel npvp_gemm_f32 picks for a shape: 0 gemm_f32_kernel, 1 gemm_split_db_kernel (128 x 128 tiles), 2
 * gemm_wide_kernel<2,4,2,2> (128 x 256 tiles, weight planes by LDS-DMA), 3 gemm_wgrad_wide_kernel, 4 gemm_wide_kernel<2,2,2,2>
 * (the same kernel on 128 x 128 tiles: outputs too small to fill the chip with wide tiles).  Pure function (measurement aid). */
int npvp_gemm_kernel_id(int a_kc, int b_kc, int M, int N, int K, int precision, int has_planes);
int npvp_gemm_f32(int a_kc, int b_kc, int M, int N, int K, const float* A, long long lda, const float* B, long long ldb,
                  float* C, long long ldc, const float* bias, int act, const float* aux_in, float* aux_out,
                  const float* residual, long long ldr, float drop_p, int drop_mode, int drop_g1, int drop_g2,
                  const unsigned long long* seed, unsigned int salt, float alpha, int precision, float* colsum_a,
                  const void* b_pre, int accumulate, float* rowstats, const float* a_amax, const float* b_amax,
                  float* c_amax, unsigned int* range_flag, float adrop_p, int adrop_g1, int adrop_g2, unsigned int adrop_salt,
                  void* workspace, long long ws_bytes, npvp_stream_t stream);
/* Range of the precision-6 (two-term fp16) arithmetic, per ROW.  Operands are scaled by ONE power of two per tensor (their amax
 * slot), so a row that lies 2^18 or more below its tensor's bound would keep only subnormal low terms.  Forward / dgrad launches
 * (kernel ids 5 / 7) repair this themselves: every thread keeps the running |max| of the rows of A it stages, and a tile in which
 * some row lies that far below the bound is computed a second time with every row scaled by the power of two of its OWN maximum
 * (exact; results bit-identical to the one-pass form when no row qualifies; cost: that tile twice).  A weight-gradient launch
 * (kernel id 6; the rows of dW are the COLUMNS of its operand A = dy) only detects the condition per workgroup K-chunk and, if
 * `range_flag` (nullable, a device counter the caller zeroed) is given, adds to it; the caller then re-runs that weight gradient
 * with precision 4, whose bf16 terms have fp32's exponent range (npvp_amd.ops.linear_wgrad: at once in strict mode, from the next
 * step on otherwise).  tests/test_hip_ops.py::test_heavy_tailed_rows_keep_their_precision. */
/* adrop_p > 0 (precision 6, launches that npvp_gemm_kernel_id maps to the fp16 kernels 5 / 6 / 7 only; anything else is an
 * argument error): a ROW-GROUP mask on the rows of operand A - row r is multiplied by the DropPath decision of group
 * (r / adrop_g1) % adrop_g2 (0 or 1 / (1 - p); same (seed, salt) stream as the forward site).  This is the backward of a DropPath
 * site (ref/models/VidHRFormer.py:513-525) without a masked copy of dy: for a dgrad (A = dy [M][K]) the mask is folded into
 * the scale of the rows a thread stages, for a weight gradient (A = dy [K][M], adrop_g1 % 16 == 0) into the scale of the K-step,
 * and the bias gradient colsum_a sums the masked rows. */
/* rowstats (nullable; default precision, a_kc = b_kc = 1, bias-only epilogue, M % 64 == 0, N % 128 == 0): receives
 * [M/64][N/64][2] partial (mean, M2) statistics of the output per frame of 64 rows and block of 64 columns;
 * npvp_frame_stats_finalize turns them into the frame LayerNorm's (mean, rstd): no statistics pass over C. */
int npvp_frame_stats_finalize(const float* part, int parts_per_frame, float values_per_part, float* mean, float* rstd,
                              int frames, float eps, npvp_stream_t stream);
/* Weights change once per optimiser step but are staged by every tile of three GEMMs: split them ONCE into the bf16
 * term planes the split-precision kernel consumes (3 terms x N*K bf16 each, blocked like the LDS image).
 * F feeds y = x w^T (pass as b_pre with b_kc = 1), D feeds dx = dy w (b_pre with b_kc = 0).  b_pre is optional
 * (NULL = split B on the fly, 128 x 128 kernel) and only honoured by precision 4 with a_kc = 1. */
int npvp_split_weight(const float* w, long long ld, int N, int K, void* F, void* D, npvp_stream_t stream);
/* The same for MANY weight views in one launch (after the optimiser step, ref/models/Predictor.py:136 opt.step()): desc is a
 * DEVICE array of `count` records of six 64-bit words {w, ld, N, K, F, D} (pointers as integers). */
int npvp_split_weights_batched(const void* desc, int count, npvp_stream_t stream);
/* ---- amax slots (precision 6).  A slot is 2 KB of device memory, zero before its tensor is produced: 32 words, 64 bytes
 * apart (word i at float offset 16 i; the rest is padding that keeps every word in its own memory sector).  Producers raise
 * single words with integer atomic max (the bit pattern of a non-negative float orders like the float: order independent, so
 * deterministic) and the tensor's bound is the maximum of the 32 words.  npvp_amax is the stand-alone producer for a
 * [rows][cols] matrix (row stride ld) that no kernel of this library wrote: slot = max(slot, |x|). */
int npvp_amax(const float* x, long long rows, long long cols, long long ld, float* slot, npvp_stream_t stream);

/* ---- weight gradients whose split-K reduction rides in the NEXT weight-gradient launch.  dW = dy^T x over 10^3..10^5 token rows
 * has only a handful of output tiles, so the precision-6 kernel (id 6) splits the reduction over ~512 workgroups and leaves
 * `splits` partial slabs to sum.  npvp_gemm_f32 sums them with a launch of its own (170 launches per training step of an 8-clip
 * shard, each an idle tail behind an MFMA-bound kernel).  The chained form defers that sum: it fills `my_job` and returns; the job
 * is handed as `prev_job` to the next chained launch ON THE SAME STREAM, where extra workgroups of that launch do it (HBM-bound work
 * beside MFMA-bound work, no launch, same summation order as the stand-alone reduction: bit-identical), or to
 * npvp_splitk_reduce_job when no launch follows (the end of a backward pass).  The caller keeps `workspace` alive until the job
 * has been handed on.  dw / db: accumulated into when `accumulate` (GradSink), else overwritten - by the job, not by this launch.
 * range_flag, adrop_*: as in npvp_gemm_f32. */
typedef struct npvp_reduce_job {
  const float* ws; float* out; long long ldc; int M, N, splits, accum; float alpha; int blocks; const float* cs_part; float* cs_out;
} npvp_reduce_job_t;                                                                 /* 64 bytes, plain data */
int npvp_wgrad_f16_chainable(int M, int N, int K);                                    /* dW [M][N] over K rows: 1 if the kernel takes it with splits > 1 */
long long npvp_wgrad_f16_chain_workspace_bytes(int M, int N, int K);
int npvp_wgrad_f16_chained(int M, int N, int K, const float* dy, long long lda, const float* x, long long ldb, float* dw,
                           long long ldc, float* db, int accumulate, const float* a_amax, const float* b_amax,
                           unsigned int* range_flag, float adrop_p, int adrop_g1, int adrop_g2, unsigned int adrop_salt,
                           const unsigned long long* seed, const void* prev_job, void* my_job, void* workspace,
                           long long ws_bytes, npvp_stream_t stream);
int npvp_splitk_reduce_job(const void* job, npvp_stream_t stream);
/* n job records (64 bytes each, back to back in HOST memory at `jobs`) in ceil(n / 32) launches; two records of one launch must not
 * name the same `out`. */
int npvp_splitk_reduce_multi(const void* jobs, int n, npvp_stream_t stream);
/* The backward of ONE linear layer y = x w^T (+ b) as ONE launch (precision 6): dx[R,K] = epilogue((adrop mask) dy[R,N] w[N,K]) and
 * dw[N,K] += dy^T x, db[N] += column sums of dy.  Both GEMMs read dy and nothing of each other; on the 8-clip shards of the
 * data-parallel configurations each of them alone leaves half the chip idle (256 workgroups), together they fill it - and a step
 * captured single-stream into a HIP graph (no gradient stream to overlap them) keeps the overlap inside the launch.  Arguments:
 * the dgrad half as npvp_gemm_f32(a_kc = 1, b_kc = 0, M = R, N = K, K = N) takes them - w_planes_d / w_amax = the weight's D planes
 * and slot (npvp_split_weight_f16), act 0 / 3 / 4 with aux_in [R][ldx], the dropout REPLAY of a forward site, residual, dx_amax
 * (nullable) - and the weight-gradient half as npvp_wgrad_f16_chained takes them (always accumulating; its split-K reduction is
 * handed on in my_job, the previous launch's comes in prev_job; npvp_splitk_reduce_job runs the last one).
 * npvp_linear_bwd_f16_takes(R, N, K): 1 when the pair is taken - a small-tile dgrad (fewer than 512 tiles of 128 x 256) and a
 * chainable weight gradient (R >= 1024); everything else stays two launches. */
int npvp_linear_bwd_f16_takes(int R, int N, int K);
int npvp_linear_bwd_f16(int R, int N, int K, const float* dy, long long ldy, const float* dy_amax, const void* w_planes_d,
                        const float* w_amax, float* dx, long long ldx, int act, const float* aux_in, const float* residual,
                        long long ldr, float drop_p, int drop_mode, int drop_g1, int drop_g2, unsigned int drop_salt, float* dx_amax,
                        const float* x, long long ldxx, const float* x_amax, float* dw, long long ldw, float* db,
                        unsigned int* range_flag, float adrop_p, int adrop_g1, int adrop_g2, unsigned int adrop_salt,
                        const unsigned long long* seed, const void* prev_job, void* my_job, void* workspace, long long ws_bytes,
                        npvp_stream_t stream);
/* w [N][K] -> the fp16 planes of precision 6: F[2 terms][K/8][N][8 over k], D[2 terms][N/8][K][8 over n] (2*N*K fp16 each,
 * either may be null), scaled by the power of two of w's amax, which is (re)computed into amax_slot (zeroed here first).
 * npvp_split_weights_f16: the same for `count` views in three stream operations; desc is a DEVICE array of records of eight
 * 64-bit words {w, ld, N, K, F, D, amax_slot, 0}; amax_table / amax_bytes (nullable) name the contiguous memory that holds
 * all the records' slots and is zeroed first (otherwise the caller zeroes the slots). */
int npvp_split_weight_f16(const float* w, long long ld, int N, int K, void* F, void* D, float* amax_slot, npvp_stream_t stream);
int npvp_split_weights_f16(const void* desc, int count, void* amax_table, long long amax_bytes, npvp_stream_t stream);
/* Producer-side slots: every entry point below whose output can be the A operand of a GEMM takes a nullable `amax` slot
 * (after its last data argument) and adds the bound of the values it stores to it - no pass over the tensor is spent on it. */

/* ---- token LayerNorm(C) (ref/models/VidHRFormer.py:65-66,69,77,175-176,179,189,194-195; shared final
 * norm :47-48,150-151; relu=1 fuses the decoder's F.relu_ :159).  C in {256,512,768,1024}.
 * mean/rstd [rows] are saved for backward (nullable in fwd). */
int npvp_layernorm_fwd(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd, long long rows,
                       int C, float eps, int relu, float* y_amax, npvp_stream_t stream);
long long npvp_layernorm_bwd_workspace_bytes(long long rows, int C);
int npvp_layernorm_bwd(const float* dy, const float* x, const float* w, const float* b, const float* mean,
                       const float* rstd, float* dx, float* dw, float* db, long long rows, int C, int relu,
                       const float* dres /* nullable: dx += dres, the residual branch's gradient */,
                       int accumulate /* 1: dw, db +=; 2: leave the partial sums in workspace */, float* dx_amax,
                       void* workspace, long long ws_bytes, npvp_stream_t stream);
/* second stage of npvp_layernorm_bwd(accumulate = 2), on a stream of the caller's choice (the gradient stream) */
int npvp_layernorm_bwd_reduce(const void* workspace, float* dw, float* db, long long rows, int C, int accumulate,
                              npvp_stream_t stream);

/* Deferred parameter-gradient reductions.  The second stages above (npvp_layernorm_bwd_reduce, npvp_frameln_act_bwd_reduce,
 * npvp_mlpdw_mid_bwd_reduce_into) each sum a set of partial rows into a gradient slice nobody reads before the optimiser: ~150
 * small launches per backward pass.  The *_reduce_job forms take the same arguments and, instead of launching, write a 48-byte job
 * record to HOST memory at `job`; npvp_sum_rows_multi runs n such records (back to back at `jobs`) with ceil(n / 40) launches, the
 * records travelling in the kernels' argument blocks (nothing to upload or keep alive; graph-capturable).  Same per-column
 * summation scheme as the single launches, fixed order.  The workspaces must stay alive until npvp_sum_rows_multi has been
 * enqueued behind their producers (same stream, or a stream ordered after it). */
int npvp_layernorm_bwd_reduce_job(const void* workspace, float* dw, float* db, long long rows, int C, int accumulate, void* job);
int npvp_frameln_act_bwd_reduce_job(const void* workspace, float* dw, float* db, int frames, int per_frame, int accumulate, void* job);
int npvp_mlpdw_mid_bwd_reduce_job(const void* workspace, float* gw, float* gb, int frames, int Ch, void* job);
int npvp_sum_rows_multi(const void* jobs, int n, npvp_stream_t stream);

/* The decoder's final LayerNorm + ReLU writing the reference's (N,T,C,H,W) tensor directly (ref/models/VidHRFormer.py:150-159:
 * norm, relu_, permute(0,1,4,2,3)).  P must be 64, C 256 or 512; mean / rstd per token row as npvp_layernorm_fwd writes them, so
 * the backward is npvp_transpose of dy followed by npvp_layernorm_bwd. */
int npvp_layernorm_nchw_fwd(const float* x, const float* w, const float* b, float* out_nchw, float* mean, float* rstd, int frames,
                            int P, int C, float eps, int relu, npvp_stream_t stream);

/* ---- PosFeatFuser 'layer' (ref/models/submodules.py:432-454: GroupNorm(1,C,affine=False) over one
 * frame's C*H*W elements, then xhat*(1+gamma)+beta).  x [N*T][per_frame], add [N][per_frame] or NULL
 * (the `+ query_evt` of ref/models/VidHRFormer.py:211,236), beta/gamma [T][per_frame] (gamma NULL for
 * fuse_method 'Add').  mean/rstd [N*T] are outputs. */
int npvp_frame_stats(const float* x, const float* add, float* mean, float* rstd, int frames, int T, int per_frame,
                     float eps, npvp_stream_t stream);
int npvp_posfuse_fwd(const float* x, const float* add, const float* beta, const float* gamma, float* y, float* mean,
                     float* rstd, int N, int T, int per_frame, float eps, float* y_amax, npvp_stream_t stream);
/* bwd: du = d(x+add) [N*T][per_frame]; dbeta / dgamma [T][per_frame] (nullable) = the sums over the batch of dy and dy*uhat (the
 * reference's autograd sums them where beta / gamma broadcast over N).  npvp_posfuse_bwd_fused(N, T, per_frame) == 1: they come out
 * of the apply pass itself (the batch loop runs inside the thread; dyxh is not touched and may be NULL); == 0 (few (t, e) columns,
 * many samples): dyxh [N*T][per_frame] is scratch for dy*uhat and two reductions follow.  accumulate = 1: dbeta / dgamma += (a
 * positional table feeds ~30 sub-layers per step: its gradient is summed in place instead of by autograd's add kernels). */
/* The pre-norm LayerNorm of an attention sub-layer and the positional fuse of its output in ONE kernel (frames of P = 64 token rows,
 * C = 512): y1 = LayerNorm(x) [N*T*P][C] with its row statistics ln_mean / ln_rstd [N*T*P], fused = posfuse(y1 (+ add)) with
 * its frame statistics pf_mean / pf_rstd [N*T]; the same results as npvp_layernorm_fwd followed by npvp_posfuse_fwd. */
int npvp_ln_posfuse_fwd(const float* x, const float* lw, const float* lb, float ln_eps, float* y1, float* ln_mean, float* ln_rstd,
                        const float* add, const float* beta, const float* gamma, float* fused, float* pf_mean, float* pf_rstd,
                        int N, int T, int P, int C, float pf_eps, float* y1_amax, float* fused_amax, npvp_stream_t stream);
int npvp_posfuse_bwd_fused(int N, int T, int per_frame);
int npvp_posfuse_bwd(const float* dy, const float* x, const float* add, const float* gamma, const float* mean,
                     const float* rstd, float* du, float* dyxh, float* dbeta, float* dgamma, int N, int T, int per_frame,
                     int accumulate, void* workspace, long long ws_bytes /* >= 8*N*T */, npvp_stream_t stream);

/* param_free_norm_type = 'instance' of the same module (ref/models/submodules.py:427-431: InstanceNorm2d(affine=False), statistics
 * per (frame, channel) over the P = H*W <= 64 pixels): x [N*T][P][C] channels-last, add [N][P][C] or NULL, beta / gamma [T][P][C],
 * mean / rstd [N*T][C].  No shipped configuration uses it; it is here so that the module's whole constructor surface runs. */
int npvp_posfuse_instance_fwd(const float* x, const float* add, const float* beta, const float* gamma, float* y, float* mean,
                              float* rstd, int N, int T, int P, int C, float eps, float* y_amax, npvp_stream_t stream);
int npvp_posfuse_instance_bwd(const float* dy, const float* x, const float* add, const float* gamma, const float* mean,
                              const float* rstd, float* du, float* dyxh, int N, int T, int P, int C, npvp_stream_t stream);

/* ---- MlpDWBN inner stages (ref/models/VidHRFormer.py:374-392) on the channels-last hidden tensor
 * h [frames][per_frame = H*W*Ch]:  out = res + droppath_n( drop( GELU( LayerNorm((Ch,H,W))(h) ) ) )
 * with the per-element affine w,b given channels-last [H*W][Ch]; mean/rstd from npvp_frame_stats. */
int npvp_frameln_act_fwd(const float* h, const float* mean, const float* rstd, const float* w, const float* b,
                         const float* res, float* out, int frames, int per_frame, float drop_p, unsigned int salt,
                         float dp_p, unsigned int dp_salt, int frames_per_sample, const unsigned long long* seed,
                         float* out_amax, npvp_stream_t stream);
long long npvp_frameln_act_bwd_workspace_bytes(int frames, int per_frame);
int npvp_frameln_act_bwd(const float* dout, const float* h, const float* mean, const float* rstd, const float* w,
                         const float* b, float* dh, float* dw, float* db, int frames, int per_frame, float drop_p,
                         unsigned int salt, float dp_p, unsigned int dp_salt, int frames_per_sample,
                         const unsigned long long* seed, int accumulate /* 1: dw, db +=; 2: leave partials */,
                         float* dh_amax, void* workspace, long long ws_bytes, npvp_stream_t stream);
int npvp_frameln_act_bwd_reduce(const void* workspace, float* dw, float* db, int frames, int per_frame, int accumulate,
                                npvp_stream_t stream);
/* depthwise 3x3, zero pad 1 (ref/models/VidHRFormer.py:351-358); wt is tap-major [9][Ch]; flip=1 gives the
 * input gradient.  wgrad writes one contiguous [10][Ch] buffer: 9 taps then the bias gradient. */
int npvp_dwconv3x3(const float* a, const float* wt, const float* bias, float* out, int frames, int H, int W, int Ch,
                   int flip, npvp_stream_t stream);
/* the same forward convolution, also returning the frame-LayerNorm statistics (mean, rstd over H*W*Ch per frame) of its
 * output, so that MlpDWBN's norm2 needs no statistics pass (8x8 grid, Ch % 1024 == 0; workspace >= frames*Ch/1024*8 B) */
int npvp_dwconv3x3_stats(const float* a, const float* wt, const float* bias, float* out, float* mean, float* rstd, int frames,
                         int H, int W, int Ch, float eps, void* workspace, long long ws_bytes, npvp_stream_t stream);
/* ---- fused middle of MlpDWBN (ref/models/VidHRFormer.py:381-385: norm1 -> GELU -> depthwise 3x3 [-> norm2's statistics]),
 * 8x8 grid, Ch % 512 == 0.  Forward: h2 = dwconv3x3(gelu(LayerNorm((Ch,H,W))(h1))) + bias in ONE pass over the hidden tensor
 * (a1 is never materialised) plus mean2 / rstd2 of h2 per frame.  wt [9][Ch] tap-major, bias [Ch], w1n / b1n [H*W][Ch].
 * workspace >= frames * (Ch/512) * 8 bytes. */
int npvp_mlpdw_mid_fwd(const float* h1, const float* mean1, const float* rstd1, const float* w1n, const float* b1n,
                       const float* wt, const float* bias, float* h2, float* mean2, float* rstd2, int frames, int H, int W,
                       int Ch, float eps, void* workspace, long long ws_bytes, npvp_stream_t stream);
/* The forward of MlpDWBN's middle and of its frame LayerNorms WITHOUT statistics launches: the consumer merges the partial
 * (mean_j, M2_j) pairs its producer left (part [frames][J][2], nb values per partial) and writes mean / rstd (outputs, for backward).
 * npvp_mlpdw_mid_fwd_parts: part1 = the rowstats of the fc1 GEMM (J1 = Ch / 64, nb1 = 4096); part2 [frames][Ch / 512][2] (32 768
 * values each) receives h2's partials.  npvp_frameln_act_fwd_parts: npvp_frameln_act_fwd with (part, J, nb, eps) in, mean / rstd
 * out; per_frame % 4096 == 0. */
int npvp_mlpdw_mid_fwd_parts(const float* h1, const float* part1, int J1, float nb1, float* mean1, float* rstd1, const float* w1n,
                             const float* b1n, const float* wt, const float* bias, float* h2, float* part2, int frames, int H,
                             int W, int Ch, float eps, npvp_stream_t stream);
int npvp_frameln_act_fwd_parts(const float* h, const float* part, int J, float nb, float eps, float* mean, float* rstd,
                               const float* w, const float* b, const float* res, float* out, int frames, int per_frame,
                               float drop_p, unsigned int salt, float dp_p, unsigned int dp_salt, int frames_per_sample,
                               const unsigned long long* seed, float* out_amax, npvp_stream_t stream);
/* Backward: da1 = conv^T(dh2); dwt_db [10][Ch] (9 tap rows + bias row; accumulate 1: +=, 2: leave the partials in workspace
 * for npvp_mlpdw_mid_bwd_reduce) with a1 recomputed from h1; psum [frames][Ch/256][2] = partial (sum g, sum g*hhat),
 * g = da1 * gelu'(y1) * w1n: the statistics of norm1's backward (npvp_frameln_act_bwd_apply, nparts = Ch/256). */
long long npvp_mlpdw_mid_bwd_workspace_bytes(int frames, int Ch);
int npvp_mlpdw_mid_bwd(const float* dh2, const float* h1, const float* mean1, const float* rstd1, const float* w1n,
                       const float* b1n, const float* wt, float* da1, float* dwt_db, float* psum, int frames, int H, int W,
                       int Ch, int accumulate, void* workspace, long long ws_bytes, npvp_stream_t stream);
int npvp_mlpdw_mid_bwd_reduce(const void* workspace, float* dwt_db, int frames, int Ch, int accumulate, npvp_stream_t stream);
/* the partials of an accumulate = 2 call straight into the Conv2d parameter gradients (weight [Ch][1][3][3], bias [Ch]), in place:
 * reduction over the chunks, transposition and accumulation in one launch */
int npvp_mlpdw_mid_bwd_reduce_into(const void* workspace, float* gw, float* gb, int frames, int Ch, npvp_stream_t stream);
/* The same with norm2's backward inside (ref VidHRFormer.py:385-387: act2(norm2(.)) + Dropout): da2 = gradient w.r.t.
 * a2 = drop(gelu(norm2(h2))); the kernel evaluates dh2 = rstd2 (g - s1 - hhat2 s2), g = da2 * mask * gelu'(y2) * w2n, element by
 * element while it fills its window, so dh2 is never written or read (two passes over the [R, Ch] tensor less).  psum2
 * [frames][nparts2 <= 256][2] = partial (sum g, sum g*hhat2) from npvp_frameln_act_bwd_pgrad, which also produces norm2's
 * parameter gradients.  drop_p / salt / seed: the forward's elementwise dropout (0 = none). */
int npvp_mlpdw_mid_bwd_n2(const float* da2, const float* h2, const float* mean2, const float* rstd2, const float* w2n,
                          const float* b2n, const float* psum2, int nparts2, float drop_p, unsigned int salt,
                          const unsigned long long* seed, const float* h1, const float* mean1, const float* rstd1,
                          const float* w1n, const float* b1n, const float* wt, float* da1, float* dwt_db, float* psum, int frames,
                          int H, int W, int Ch, int accumulate, void* workspace, long long ws_bytes, npvp_stream_t stream);
/* frame-LN backward WITHOUT the input gradient: psum [frames][per_frame / 1024][2] = partial (sum g, sum g*hhat) for a consumer
 * that evaluates dh itself (npvp_mlpdw_mid_bwd_n2); dw / db, accumulate and workspace exactly as npvp_frameln_act_bwd
 * (npvp_frameln_act_bwd_reduce applies).  per_frame % 1024 == 0. */
int npvp_frameln_act_bwd_pgrad(const float* dout, const float* h, const float* mean, const float* rstd, const float* w,
                               const float* b, float* psum, float* dw, float* db, int frames, int per_frame, float drop_p,
                               unsigned int salt, float dp_p, unsigned int dp_salt, int frames_per_sample,
                               const unsigned long long* seed, int accumulate, void* workspace, long long ws_bytes,
                               npvp_stream_t stream);
/* frame-LN backward with the statistics supplied by the producer of dout (one pass instead of two; no dropout):
 * psum [frames][nparts][2]; workspace as npvp_frameln_act_bwd. */
int npvp_frameln_act_bwd_apply(const float* dout, const float* h, const float* mean, const float* rstd, const float* w,
                               const float* b, const float* psum, int nparts, float* dh, float* dw, float* db, int frames,
                               int per_frame, int accumulate, float* dh_amax, void* workspace, long long ws_bytes,
                               npvp_stream_t stream);

/* im2col / col2im of the EventEncoder's dense 3x3 conv (ref/models/submodules.py:376), channels-last:
 * col2im=0: in [F][H*W][C] -> out [F*H*W][9*C] (tap-major columns); col2im=1: the adjoint. */
int npvp_im2col3x3(const float* in, float* out, int frames, int H, int W, int C, int col2im, npvp_stream_t stream);
long long npvp_dwconv3x3_wgrad_workspace_bytes(int frames, int Ch);
int npvp_dwconv3x3_wgrad(const float* a, const float* dout, float* dwt_db, int frames, int H, int W, int Ch,
                         void* workspace, long long ws_bytes, npvp_stream_t stream);

/* ---- attention cores (the bmm/softmax/dropout/bmm inside torch.nn.MultiheadAttention as called at
 * ref/models/VidHRFormer.py:104-107 (encoder temporal, masked), :221 (decoder temporal), :239 (enc-dec),
 * :298-300 (spatial window; window gather = ref :447-475 as index math)).
 *   mode 0 spatial: dim0 = N*T frames, groups = frames x windows, L = S = ws*ws
 *   mode 1 temporal: dim0 = N, groups = N x P pixels, L = Tq, S = Tk (rows (n*T + t)*P + p)
 *   mask_mode 1 = encoder quirk ref :100-102.  q NOT pre-scaled; head_dim must be 64; L, S <= 128 (up to 32: the MFMA kernels every shipped
 *   configuration runs on; 33 .. 128: generic kernels - one workgroup per (group, head), operands in LDS, scalar fp32 arithmetic:
 *   the reference's nn.MultiheadAttention has no length limit, ref VidHRFormer.py:94-107). */
int npvp_attn_fwd(const float* q, long long ld_q, const float* k, long long ld_k, const float* v, long long ld_v, float* o,
                  long long ld_o, int mode, int dim0, int P, int W, int ws, int Tq, int Tk, int heads, int head_dim,
                  int mask_mode, float drop_p, const unsigned long long* seed, unsigned int salt, float* o_amax,
                  npvp_stream_t stream);
int npvp_attn_bwd(const float* q, long long ld_q, const float* k, long long ld_k, const float* v, long long ld_v,
                  const float* go, long long ld_o, float* dq, long long ld_dq, float* dk, long long ld_dk, float* dv,
                  long long ld_dv, int mode, int dim0, int P, int W, int ws, int Tq, int Tk, int heads, int head_dim,
                  int mask_mode, float drop_p, const unsigned long long* seed, unsigned int salt, float* dq_amax,
                  float* dk_amax /* may equal dq_amax: one slot for a packed q|k gradient */, float* dv_amax,
                  npvp_stream_t stream);

/* ---- layout / reductions / masks */
int npvp_drop_apply(const float* x, float* out, long long rows, int ncols, float p, int mode, int g1, int g2,
                    const unsigned long long* seed, unsigned int salt, float* out_amax, npvp_stream_t stream);
int npvp_transpose(const float* in, float* out, int batch, int R, int C, npvp_stream_t stream); /* [B][R][C]->[B][C][R] */
/* Gradient slots of nn.Conv2d(C, C, 3, groups=C) (ref VidHRFormer.py:351-358: weight [C][1][3][3], bias [C]) from the tap-major
 * table [10][C] (9 weight rows + the bias row) that npvp_mlpdw_mid_bwd / npvp_dwconv3x3_wgrad produce: gw[c][tap] += dwtb[tap][c],
 * gb[c] += dwtb[9][c].  Accumulates in place (the flat gradient buffer of the optimiser). */
int npvp_dwtb_accumulate(const float* dwtb, float* gw, float* gb, int C, npvp_stream_t stream);
/* forward direction: weight [C][1][3][3] + bias [C] (NULL = zeros) -> the tap-major table wtb [10][C] (9 tap rows + the bias row) */
int npvp_dwtb_build(const float* w, const float* b, float* wtb, int C, npvp_stream_t stream);
int npvp_reduce_mid(const float* in, float* out, int A, int B, long long Cc, float scale, int accumulate, npvp_stream_t stream);    /* accumulate = 1: out += */
int npvp_broadcast_mid(const float* in, float* out, int A, int B, long long Cc, float scale, npvp_stream_t stream);

/* out[n] = sum_r x[r][n]: bias gradients of every Linear / Conv2d on the path */
long long npvp_colsum_workspace_bytes(long long rows, int N);
int npvp_colsum(const float* x, long long rows, int N, long long ld, float* out, int accumulate /* out += */,
                void* workspace, long long ws_bytes, npvp_stream_t stream);

/* ---- scalar losses of shared_step (ref/models/Predictor.py:172-194): L1Loss = lam * mean|a - b| (ref/models/criterion.py:99-121)
 * and the sum under Div_KL (ref/models/criterion.py:341-354).  Two-stage sums in a fixed order: deterministic, and - unlike a
 * multi-block torch reduction - without a semaphore that a memset node of a captured graph has to clear.  out / gout are device
 * scalars; l1_mean_bwd writes da = sgn(a - b) * ((gout * lam) / n).  workspace >= 1024 floats. */
int npvp_l1_mean(const float* a, const float* b, long long n, float lam, float* out, void* workspace, long long ws_bytes,
                 npvp_stream_t stream);
int npvp_l1_mean_bwd(const float* a, const float* b, long long n, const float* gout, float lam, float* da, npvp_stream_t stream);
int npvp_sum_all(const float* x, long long n, float* out, void* workspace, long long ws_bytes, npvp_stream_t stream);

/* ---- optimiser step of training_step_no_gan (ref/models/Predictor.py:135-136,197): clip_grad_norm_
 * over a flat gradient range, then torch.optim.AdamW semantics on flat buffers.  hyper = {lr, step}
 * and clip = {norm, coef} live in device memory. */
int npvp_grad_norm_clip(const float* g, long long n, float max_norm, float* out2, void* workspace, long long ws_bytes,
                        npvp_stream_t stream);
int npvp_adamw_step(float* p, float* g, float* m, float* v, long long n, const float* hyper, float beta1, float beta2,
                    float eps, float weight_decay, const float* clip, long long clip_begin, long long clip_end,
                    int write_back_grad, npvp_stream_t stream);

/* ---- evaluation metrics on device (ref/utils/metrics.py:12-43 PSNR / MSEScore, :46-108 SSIM, called per
 * predicted time-step by pred_ave_metrics :110-140).  Images are [N][C][H][W] fp32 (per_image = C*H*W); out is N floats.
 *   sqdiff: out[n] = scale * sum_i ((x[n][i] - y[n][i]) / data_range)^2   (PSNR: scale = 1/per_image, then -10 log10(. + 1e-8)
 *           on the N results; MSEScore: scale = 1, data_range = 1)
 *   ssim:   out[n] = mean over (C,H,W) of the SSIM map, zero-padded Gaussian window; `taps` is a HOST array of window_size
 *           floats (the normalised 1-D Gaussian; the reference's 2-D window is its outer product, :78-83); window_size in
 *           {3,5,7,11}; N*C < 65536. */
long long npvp_sqdiff_workspace_bytes(int N, long long per_image);
int npvp_sqdiff_per_image(const float* x, const float* y, int N, long long per_image, float data_range, float scale, float* out,
                          void* workspace, long long ws_bytes, npvp_stream_t stream);
long long npvp_ssim_workspace_bytes(int N, int C, int H, int W);
int npvp_ssim_per_image(const float* img1, const float* img2, int N, int C, int H, int W, const float* taps /* host */,
                        int window_size, float* out, void* workspace, long long ws_bytes, npvp_stream_t stream);

/* ---- input frames (ref/utils/dataset.py:835-858 VidToTensor + VidNormalize): uint8 HWC frames as decoded ->
 * normalised fp32 (frames, C, H, W): dst = (src / 255 - mean[c]) / std[c].  mean / std are HOST arrays of C floats; C in {1,3,4};
 * frames < 65536. */
int npvp_u8hwc_to_f32chw(const void* src_u8, float* dst, long long frames, int H, int W, int C, const float* mean,
                         const float* std, npvp_stream_t stream);

/* ---- frozen Stage-1 autoencoder epilogues (SURVEY 8f #1 stage 2; ref/models/ResNetAutoEncoder.py:51-261, submodules.py:9-95):
 * with BatchNorm folded into the frozen convolution weights, conv -> BN -> ReLU (-> + skip) is MIOpen's convolution plus ONE pass
 *   out = act(x + bias[c]) + residual      act: 0 none, 1 ReLU, 2 tanh, 3 sigmoid; residual nullable
 *   layout 0: x [outer][inner = C], channel = column (channels_last memory); layout 1: x [outer = N*C][inner = H*W] (NCHW).
 * act_bwd: dx = g * act'(.) from the forward output y (no residual), the decoder's input-gradient path. */
int npvp_bias_act(const float* x, const float* bias, const float* residual, float* out, long long outer, long long inner, int C,
                  int layout, int act, npvp_stream_t stream);
int npvp_act_bwd(const float* g, const float* y, float* dx, long long n, int act, npvp_stream_t stream);

/* ---- the data-parallel exchange: all-reduce(mean) of the parameter gradients, RCCL over xGMI, one process per GPU.
 * Replaces what the reference gets from Lightning's DDP strategy (ref/train_Predictor_lightning.py:40-42: strategy = 'ddp' over
 * `devices` GPUs) for a host without torch.distributed.  Protocol: rank 0 calls npvp_dp_unique_id and carries the 128
 * bytes to every rank (file, socket, MPI, a torch.distributed store ...); every rank, with its device current, calls npvp_dp_init
 * (collective).  Per step: order `side` after the producers of a bucket of the flat gradient buffer, npvp_dp_allreduce_async(bucket,
 * n, side) - in place, mean over the ranks, same buckets in the same order on every rank, the host never blocks - and before the
 * optimiser npvp_dp_wait(compute): the compute stream waits on the device for every reduction enqueued so far.  npvp_dp_finalize
 * drains and releases the communicator.  RCCL is found at run time (the copy already loaded in the process, else $NPVP_RCCL_LIB, else
 * librccl.so.1 on the loader's path); a process that never calls these never loads it.  One communicator per process. */
int npvp_dp_unique_id(void* id_out_128_bytes);
int npvp_dp_init(int rank, int world, const void* unique_id_128_bytes);
int npvp_dp_world(void); /* 0 before npvp_dp_init */
int npvp_dp_rank(void);  /* -1 before npvp_dp_init */
int npvp_dp_allreduce_async(float* bucket, size_t n, npvp_stream_t side);
int npvp_dp_wait(npvp_stream_t compute);
int npvp_dp_finalize(void);

#ifdef __cplusplus
}
#endif
#endif
