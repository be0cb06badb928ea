// Two-term fp16 split GEMMs ("f16x3"): the fp32-grade arithmetic of the large forward / dgrad / weight-gradient GEMMs at
// THREE matrix instructions per product instead of the six of the three-term bf16 split (gemm_wide.hip).
//
//     x * s = hi + lo + eps,   hi = rne_f16(x s),  lo = rne_f16(x s - hi),   |eps| <= 2^-23 |x s|  (while lo is a normal)
//     a b  ~=  (hi_a hi_b + hi_a lo_b + lo_a hi_b) / (s_a s_b)              on v_mfma_f32_32x32x16_f16, fp32 accumulate
//
// fp16 carries 11 significand bits, so two terms reach 2^-22 where bf16 needs three; the dropped lo_a lo_b term is
// <= 2^-22 relative.  Representation error of a whole product sum against fp64: 7.6e-8 rel-L2 (tools/f16_emulate.py; an fp32
// GEMM's own accumulation error is 3e-7).  What fp16 does NOT have is fp32's exponent range, so every operand tensor comes
// with an AMAX SLOT - 32 words (64 bytes apart) whose maximum bounds |x| over the tensor - and is multiplied by the power of two s that
// puts that bound into [2^14, 2^15): nothing overflows (max 65504), and an element 2^-18 below the bound still has a
// NORMAL low term.  Smaller elements lose low-term bits gradually (fp16 subnormals: gfx950 converts to them and its MFMA
// multiplies them exactly - tools/f16_probe.hip), i.e. they carry an ABSOLUTE error of 2^-40 of the tensor's bound, which
// no dot product that contains the larger elements can see.  The slots are filled by the producers of the tensors (an
// epilogue max + one atomic per wave, order independent and therefore deterministic) or by npvp_amax; weights are split
// once per optimiser step into scaled planes (F / D, laid out like the LDS image, copied HBM -> LDS by LDS-DMA).
//
// Kernel organisation = gemm_wide.hip's (128 x 256 or 128 x 128 tile, 4 waves, K-step 16, two LDS stages, two workgroups
// per CU) with two planes per operand: 24.8 KB per stage instead of 36.4, 24 MFMAs and ~20 VALU of splitting per wave and
// K-step instead of 48 and ~60.
#include "gemm.h"
#include <cstdlib>

namespace npvp {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// x, s -> (hi, lo) of x s:  hi = rne_f16(x s),  lo = rne_f16(x s - hi).  TWO VALU instructions per element, the mixed-precision FMAs
// (v_fma_mixlo_f16 / v_fma_mixhi_f16: an fp32 FMA whose operands may be fp16 halves and whose result is rounded into one half of the
// destination): hi = mix(x, s, 0), lo = mix(x, s, -hi).  Both FMAs are exact before the conversion (s is a power of two; x s - hi
// has at most 13 significant bits), so the halves are bit for bit those of "scale, convert, convert back, subtract, convert"
// (tools/f16_split_check.hip runs both forms over random and edge-case words on the GPU) - which cost 3 VALU per element as the
// compiler's best (pk_mul, cvt_pk, 2 cvt back, pk_fma, cvt_pk per pair; 8 - 9 in the scalar form of round 3).  Every VALU
// instruction of the K loop costs its 4 cycles on top of the MFMAs' (profiles/r03_gemm_f16_latency.txt); the split was 28 of the
// forward loop's 36 per step and ~76 of the weight-gradient loop's.
__device__ __forceinline__ void split_f16_scaled(const f32x4 v, const float s, f16x4& hi, f16x4& lo) {
  unsigned int h01, h23, l01, l23;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h01) : "v"(v[0]), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h23) : "v"(v[2]), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h01) : "v"(v[1]), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h23) : "v"(v[3]), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l01) : "v"(v[0]), "v"(s), "v"(h01));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l23) : "v"(v[2]), "v"(s), "v"(h23));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l01) : "v"(v[1]), "v"(s), "v"(h01));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l23) : "v"(v[3]), "v"(s), "v"(h23));
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 hp = {h01, h23}, lp = {l01, l23};
  hi = __builtin_bit_cast(f16x4, hp);
  lo = __builtin_bit_cast(f16x4, lp);
}

// workgroup-wide OR of a per-thread predicate (every thread calls it; NW waves; `flags` = NW ints of LDS nobody else uses)
template <int NW>
__device__ __forceinline__ bool block_any(bool pred, int* flags, int wave) {
  const bool w = __builtin_amdgcn_ballot_w64(pred) != 0ull;
  if ((threadIdx.x & 63) == 0) flags[wave] = w ? 1 : 0;
  __syncthreads();
  int any = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) any |= flags[i];
  __syncthreads();                                    // (the flags may be written again by a later call)
  return __builtin_amdgcn_readfirstlane(any) != 0;
}

// NST = LDS stages.  2: B tile kt+1 is DMA'd during step kt and must have landed at the step's end - fine when a step is long (the
// 128 x 256 tile: 24 MFMAs per wave and step, two workgroups per CU).  3 (the small tiles, round 4): B tile kt+2 is DMA'd during
// step kt, i.e. it has two steps to arrive: with 6 - 12 MFMAs per wave and step and one or two workgroups per CU a step was as
// long as the L2 -> LDS latency of its B pieces (0.42 us per K-step, three quarters of a shard-sized GEMM's loop).
// LDS of one workgroup of gemm_f16_body: the operand stages, reused as the epilogue's scratch
template <int TM, int TN, int WM, int WN, int NST>
constexpr int gemm_f16_lds_bytes() {
  constexpr int NW = WM * WN, BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int STAGE = 2 * 2 * (BM * 16 + 64) + 2 * 2 * (BN * 16);
  constexpr int EPI_BYTES = (4 + NW * EPI_FLOATS + BM) * 4;
  return NST * STAGE > EPI_BYTES ? NST * STAGE : EPI_BYTES;
}

// The kernel's body: workgroup `bid` of the `nwg` that cover the problem, on `lds` (gemm_f16_lds_bytes of it).  A function so that a
// launch can hold more than one problem (gemm_f16_group_kernel below: a dgrad and the weight gradient of the same dy in ONE launch).
template <int TM, int TN, int WM, int WN, bool ROWSTATS, int NST>
__device__ __forceinline__ void gemm_f16_body(const GemmParams& p, char* lds, const int bid, const int nwg) {
  constexpr int NW = WM * WN, THREADS = 64 * NW;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int APASS = BM / (THREADS / 4);
  static_assert((APASS == 1 || APASS == 2) && BN % 64 == 0, "A staging: a thread stages one or two rows of the tile");
  constexpr int KGS_A = BM * 16 + 64, A_PLANE = 2 * KGS_A, A_BYTES = 2 * A_PLANE;
  constexpr int KGS_B = BN * 16, B_PLANE = 2 * KGS_B, B_BYTES = 2 * B_PLANE;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int CPS = BN / 64;                       // 1 KB glds chunks per (term, k-group) slab
  constexpr int NCHUNK = 4 * CPS, CPW = (NCHUNK + NW - 1) / NW;
  static_assert(NST == 2 || NST == 3, "two or three LDS stages");
  static_assert(gemm_f16_lds_bytes<TM, TN, WM, WN, NST>() >= NST * STAGE, "lds size formula");

  int tile_m, tile_n;
  tile_of_block_unsplit(p, nwg, bid, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
  const int nk = p.K >> 4;
  const float sa = amax_scale(amax_slot_read(p.a_amax));
  const unsigned int cpeek = amax_peek_block(p.c_amax);

  f32x16 acc[TM][TN];                                 // (written by the first K-step: never zeroed)

  const int quad = t & 3, rl = t >> 2;
  // global addresses = wave-uniform 64-bit base (SGPRs, advanced per K-step by scalar adds) + a per-lane 32-bit byte offset that
  // never changes: no 64-bit vector adds inside the K loop (they were 6 of its 38 VALU per step)
  const char* a_base = reinterpret_cast<const char*>(p.A + (long long)m0 * p.lda);
  const unsigned int a_off0 = (unsigned int)(((long long)(min(m0 + rl, p.M - 1) - m0) * p.lda + 4 * quad) * 4);
  const unsigned int a_off1 = (unsigned int)(((long long)(min(m0 + rl + BM / 2, p.M - 1) - m0) * p.lda + 4 * quad) * 4);
  const int a_dst = (quad >> 1) * KGS_A + (quad & 1) * 8 + rl * 16;
  // a row-group mask on A (the backward of a DropPath site): this thread stages the same two rows in every K-step, so the
  // mask is folded into their scales once
  float sa0 = sa, sa1 = sa;
  if (p.adrop.thresh) {
    const unsigned long long aseed = *p.seed;
    sa0 *= drop_spec_scale(p.adrop, aseed, min(m0 + rl, p.M - 1), 0, 1);
    sa1 *= drop_spec_scale(p.adrop, aseed, min(m0 + rl + BM / 2, p.M - 1), 0, 1);
  }
  const char* b_base = reinterpret_cast<const char*>(p.b_pre);
  unsigned int b_off[CPW];
  int b_dst[CPW];
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int c = wave + NW * i;
    const int slab = c / CPS, part = c - slab * CPS, s = slab >> 1, kg = slab & 1;
    const int col = min(n0 + part * 64 + lane, p.N - 1);
    b_off[i] = (unsigned int)(((long long)s * (p.b_pre_plane >> 3) + (long long)kg * p.N + col) * 16);
    b_dst[i] = A_BYTES + s * B_PLANE + kg * KGS_B + part * 1024;
  }
  const long long b_step = 32ll * p.N;                // bytes per K-step
  const unsigned int lds_u32 = (unsigned int)(size_t)((lptr_t)lds);

  const int fa_off = h * KGS_A + (wm * TM * 32 + r) * 16;
  const int fb_off = A_BYTES + h * KGS_B + (wn * TN * 32 + r) * 16;

  // A tiles travel global -> registers -> (split) -> LDS planes, TWO tiles ahead in two register sets (ea / eb for the even /
  // odd step of the unrolled pair).  The loads are inline asm with hand-counted waits: beside an outstanding LDS-DMA hipcc
  // waits vmcnt(0) at the first use of any loaded register, which put the latency of the loads a step had just issued
  // in front of every barrier.  VMEM issue order per step: 4 (CPW) LDS-DMA pieces of B tile kt+1, then 2 loads of A tile
  // kt+3; "vmcnt(2)" at the step's end therefore waits for everything but those 2 loads - B kt+1 (one step of latency
  // budget) and A kt+2 (a step and a half) have landed, and the registers consumed in the next step are valid.
  f32x4 ea0, ea1 = {0.f, 0.f, 0.f, 0.f}, eb0, eb1 = {0.f, 0.f, 0.f, 0.f};
#define NPVP_H_ALOAD(R0, R1, KT)                                                                           \
  { const char* ab_ = a_base + ((long long)min((KT), nk - 1) << 6);                                        \
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(R0) : "v"(a_off0), "s"(ab_) : "memory");          \
    if constexpr (APASS == 2) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(R1) : "v"(a_off1), "s"(ab_) : "memory"); }
  // everything but the APASS loads just issued has landed
#define NPVP_H_WAIT_BUT_NEWEST(...)                                                                        \
  { if constexpr (APASS == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" : __VA_ARGS__ :: "memory");     \
    else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" : __VA_ARGS__ :: "memory"); }
  // three stages: everything but what THIS step issued has landed - its CPW LDS-DMA pieces (B tile kt+2) and its APASS loads
  // (A tile kt+3); vector-memory operations complete in issue order, so B tile kt+1 and A tile kt+2 (issued a step ago) are in
  // (two tied register operands: the count is %2)
#define NPVP_H_WAIT_BUT_THIS_STEP(...)                                                                     \
  asm volatile("s_waitcnt vmcnt(%2) lgkmcnt(0)" : __VA_ARGS__ : "n"(CPW + APASS) : "memory");
#define NPVP_H_ASTORE(ST, V, ROWOFF)                                                     \
  { if (ROWOFF) rm1 = fmaxf(fmaxf(rm1, fmaxf(fabsf((V)[0]), fabsf((V)[1]))), fmaxf(fabsf((V)[2]), fabsf((V)[3])));  \
    else rm0 = fmaxf(fmaxf(rm0, fmaxf(fabsf((V)[0]), fabsf((V)[1]))), fmaxf(fabsf((V)[2]), fabsf((V)[3])));         \
    f16x4 hi_, lo_; split_f16_scaled((V), (ROWOFF) ? sa1 : sa0, hi_, lo_);               \
    *reinterpret_cast<f16x4*>((ST) + a_dst + (ROWOFF)) = hi_;                            \
    *reinterpret_cast<f16x4*>((ST) + A_PLANE + a_dst + (ROWOFF)) = lo_; }
  // LDS-DMA with a scalar base + 32-bit lane offset, M0 = the piece's LDS address (the builtin takes a 64-bit per-lane pointer
  // and pays a 64-bit vector add per piece)
#define NPVP_H_BLOAD(ST, KT)                                                             \
  { const char* bb_ = b_base + (long long)min((KT), nk - 1) * b_step;                    \
    _Pragma("unroll") for (int i_ = 0; i_ < CPW; ++i_)                                   \
      if (NCHUNK % NW == 0 || wave + NW * i_ < NCHUNK) {                                 \
        const unsigned int m0_ = lds_u32 + (unsigned int)((ST) - lds) + (unsigned int)b_dst[i_];          \
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"                     \
                     :: "s"(m0_), "v"(b_off[i_]), "s"(bb_) : "memory", "m0"); } }
  static_assert(NCHUNK % NW == 0, "the hand-counted waits assume CPW LDS-DMA pieces per wave and step");

  // ROW GUARD.  One scale per TENSOR leaves a token row that lies far below the tensor's bound with few significant bits (its
  // low terms are fp16 subnormals: an absolute error of 2^-40 of the bound).  Every thread therefore keeps the running |max| of
  // the two rows it stages (rm0 / rm1: 4 v_max3 per K-step); when the K loop is done and some row of the tile turns out to lie
  // 2^18 or more below the bound (its largest low term was subnormal), the TILE is computed again with every row scaled by the
  // power of two of its OWN maximum - the staging already multiplies by a per-thread factor (sa0 / sa1), so the second pass is
  // the same loop - and the accumulators are brought back to the tensor's scale (exact: powers of two) before the epilogue.
  // Real training tensors never take the second pass (their smallest rows sit 2^15 below the bound, profiles/r04_f16_range_audit.txt);
  // a tile whose rows span more than 2^18 costs twice.  Zero rows (dropped samples) and rows below 2^-111 are left alone.
  __shared__ int guard_flags[NW];
  float rm0 = 0.f, rm1 = 0.f;
  bool rescued = false;
  const float zrow = 1.f;
  float rf0 = zrow, rf1 = zrow;                         // 2^-(distance of the row's exponent from the bound's), second pass only
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto run_tile = [&]() __attribute__((always_inline)) {
  // prologue: tile 0 -> stage 0 (B by DMA, A through the registers), A tiles 1 and 2 -> register sets
  NPVP_H_BLOAD(lds, 0)
  NPVP_H_ALOAD(ea0, ea1, 0)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(ea0), "+v"(ea1) :: "memory");
  NPVP_H_ASTORE(lds, ea0, 0)
  if constexpr (APASS == 2) NPVP_H_ASTORE(lds, ea1, (BM / 2) * 16)
  if constexpr (NST == 3) NPVP_H_BLOAD(lds + STAGE, 1)      // (three stages: B tile 1 leaves here, step kt DMAs tile kt + 2)
  NPVP_H_ALOAD(ea0, ea1, 1)
  NPVP_H_ALOAD(eb0, eb1, 2)
  NPVP_H_WAIT_BUT_NEWEST("+v"(ea0), "+v"(ea1))
  __builtin_amdgcn_s_barrier();

  // one K-step: MFMAs of tile KT on stage CUR; B tile KT+1 is DMA'd into NXT; A tile KT+1 (register set R, loaded two
  // steps ago) is split and written to NXT, then R is reloaded with A tile KT+3.  RN = the set the NEXT step consumes: the
  // step's closing wait is tied to it so that nothing that reads it can be scheduled above the wait.
#define NPVP_H_STEP(KT, CUR, NXT, R0, R1, RN0, RN1) NPVP_H_STEP_(KT, CUR, NXT, NXT, R0, R1, RN0, RN1, false)
  // three stages: B2 = the stage that receives B tile KT+2
#define NPVP_H_STEP3(KT, CUR, NXT, B2, R0, R1, RN0, RN1) NPVP_H_STEP_(KT, CUR, NXT, B2, R0, R1, RN0, RN1, false)
  // FIRST: the step's first MFMA per accumulator takes the constant 0 as its C operand (the accumulators are never zeroed: 128
  // v_mov less per wave and tile)
#define NPVP_H_STEP_(KT, CUR, NXT, B2, R0, R1, RN0, RN1, FIRST)                                            \
  {                                                                                                        \
    const char* st_ = lds + (CUR) * STAGE;                                                                 \
    char* nx_ = lds + (NXT) * STAGE;                                                                       \
    NPVP_H_BLOAD(lds + (B2) * STAGE, (KT) + (NST - 1))                                                     \
    f16x8 fb_[2][TN];                                                                                      \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                       \
      _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_)                                                    \
        fb_[s_][j_] = *reinterpret_cast<const f16x8*>(st_ + fb_off + s_ * B_PLANE + j_ * 512);             \
    _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) {                                                    \
      f16x8 fa_[2];                                                                                        \
      _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                     \
        fa_[s_] = *reinterpret_cast<const f16x8*>(st_ + fa_off + s_ * A_PLANE + i_ * 512);                 \
      if (i_ == 0) { NPVP_H_ASTORE(nx_, R0, 0) if constexpr (APASS == 2) NPVP_H_ASTORE(nx_, R1, (BM / 2) * 16) } \
      if (i_ == TM - 1) { NPVP_H_ALOAD(R0, R1, (KT) + 3) }                                                 \
      /* smallest terms first */                                                                           \
      _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_[1], fb_[0][j_], (FIRST) ? zero16 : acc[i_][j_], 0, 0, 0); \
      _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_[0], fb_[1][j_], acc[i_][j_], 0, 0, 0); \
      _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_[0], fb_[0][j_], acc[i_][j_], 0, 0, 0); \
    }                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);       /* every MFMA of the step is issued before the wave parks at the wait */ \
    if constexpr (NST == 3) { NPVP_H_WAIT_BUT_THIS_STEP("+v"(RN0), "+v"(RN1)) }                            \
    else { NPVP_H_WAIT_BUT_NEWEST("+v"(RN0), "+v"(RN1)) }                                                  \
    __builtin_amdgcn_s_barrier();                                                                          \
  }

  if constexpr (NST == 2) {
  NPVP_H_STEP_(0, 0, 1, 1, ea0, ea1, eb0, eb1, true)
  int kt = 1;
  for (; kt + 1 < nk; kt += 2) {
    NPVP_H_STEP(kt, 1, 0, eb0, eb1, ea0, ea1)
    NPVP_H_STEP(kt + 1, 0, 1, ea0, ea1, eb0, eb1)
  }
  if (kt < nk) NPVP_H_STEP(kt, 1, 0, eb0, eb1, ea0, ea1)
  } else {
  // step s: MFMAs on stage s % 3, A tile s+1 -> stage (s+1) % 3, B tile s+2 -> stage (s+2) % 3 (the stage step s-1 multiplied
  // from: every wave has passed that step's closing barrier); register sets alternate as above: period 6
  NPVP_H_STEP_(0, 0, 1, 2, ea0, ea1, eb0, eb1, true)
  int kt = 1;
  for (; kt + 5 < nk; kt += 6) {
    NPVP_H_STEP3(kt, 1, 2, 0, eb0, eb1, ea0, ea1)
    NPVP_H_STEP3(kt + 1, 2, 0, 1, ea0, ea1, eb0, eb1)
    NPVP_H_STEP3(kt + 2, 0, 1, 2, eb0, eb1, ea0, ea1)
    NPVP_H_STEP3(kt + 3, 1, 2, 0, ea0, ea1, eb0, eb1)
    NPVP_H_STEP3(kt + 4, 2, 0, 1, eb0, eb1, ea0, ea1)
    NPVP_H_STEP3(kt + 5, 0, 1, 2, ea0, ea1, eb0, eb1)
  }
  if (kt < nk) NPVP_H_STEP3(kt, 1, 2, 0, eb0, eb1, ea0, ea1)
  if (kt + 1 < nk) NPVP_H_STEP3(kt + 1, 2, 0, 1, ea0, ea1, eb0, eb1)
  if (kt + 2 < nk) NPVP_H_STEP3(kt + 2, 0, 1, 2, eb0, eb1, ea0, ea1)
  if (kt + 3 < nk) NPVP_H_STEP3(kt + 3, 1, 2, 0, ea0, ea1, eb0, eb1)
  if (kt + 4 < nk) NPVP_H_STEP3(kt + 4, 2, 0, 1, eb0, eb1, ea0, ea1)
  }
  // (the clamped loads past the last tile.  Both register sets are TIED to the wait: the last two steps' loads are never read,
  //  and a dead asm output may be given a register that the code between the load and this wait uses for something else -
  //  the load then lands in it.  Seen as intermittent wrong tiles once the row guard changed the register allocation.)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(ea0), "+v"(ea1), "+v"(eb0), "+v"(eb1) :: "memory");
  // (three stages: the last step's clamped B pieces were still in flight at its closing barrier - every wave's must be in before
  //  any wave turns the stages into the epilogue's scratch or a second pass rewrites stage 0)
  if constexpr (NST == 3) __builtin_amdgcn_s_barrier();
  };
  run_tile();
  {
    rm0 = fmaxf(rm0, __shfl_xor(rm0, 1)); rm0 = fmaxf(rm0, __shfl_xor(rm0, 2));       // the four lanes that stage one row
    rm1 = fmaxf(rm1, __shfl_xor(rm1, 1)); rm1 = fmaxf(rm1, __shfl_xor(rm1, 2));
    const int et = (int)((__float_as_uint(sa) >> 23) & 0xffu);                         // sa = 2^(141 - E_bound): exponent field 268 - E_bound
    const int e0 = (int)((__float_as_uint(rm0) >> 23) & 0xffu), e1 = (int)((__float_as_uint(rm1) >> 23) & 0xffu);
    // distance of the row's exponent below the bound's: d = E_bound - E_row = (268 - et) - e
    const int d0 = 268 - et - e0, d1 = 268 - et - e1;
    const bool ok0 = e0 >= 16 && e0 != 255 && d0 > 0 && d0 <= 120, ok1 = e1 >= 16 && e1 != 255 && d1 > 0 && d1 <= 120;
    // (a bound too small to scale - amax_scale returned 1 - reads as E_bound = 141 here; its rows have e < 16 and are left alone)
    if (block_any<NW>((ok0 && d0 >= 18) || (ok1 && d1 >= 18), guard_flags, wave)) {
      const float g0 = sa0 * pow2_recip(sa), g1 = sa1 * pow2_recip(sa);                // the row-group mask factors (1 without a_drop)
      rf0 = ok0 ? __uint_as_float((unsigned int)(127 - d0) << 23) : 1.f;
      rf1 = ok1 ? __uint_as_float((unsigned int)(127 - d1) << 23) : 1.f;
      sa0 = (ok0 ? amax_scale(rm0) : sa) * g0;
      sa1 = (ok1 ? amax_scale(rm1) : sa) * g1;
      rescued = true;
      run_tile();
    }
  }
#undef NPVP_H_STEP
#undef NPVP_H_STEP3
#undef NPVP_H_WAIT_BUT_THIS_STEP
#undef NPVP_H_STEP_
#undef NPVP_H_BLOAD
#undef NPVP_H_ASTORE
#undef NPVP_H_ALOAD
#undef NPVP_H_WAIT_BUT_NEWEST

  // Back to the operands' own scale: C = acc * (1/sa) * (1/sb) * alpha + bias.  The two scales are powers of two, so their product
  // times alpha is exact and ONE fused multiply-add per element (the epilogue's own alpha * acc + bias) rounds exactly like
  // scaling first - as long as that combined factor is a normal number.  Otherwise (bounds near the ends of fp32's range) the
  // accumulators are scaled in two exact steps, so that neither product of scales can under- or overflow.  Folding saves 32 VALU
  // per 32 x 32 tile (the scalings were a third of the epilogue's instructions).
  const float ia = pow2_recip(sa), ib = pow2_recip(amax_scale(amax_slot_read(p.b_amax)));
  float alpha = p.alpha;                 // (handed to the epilogue beside p: a modified COPY of the parameter block lives in scratch)
  {
    const float f = ia * ib, af = p.alpha * f, mag = fabsf(af);
    const bool fold = f >= 1.17549435e-38f && f <= 3.0e38f && ((mag >= 1.17549435e-38f && mag <= 3.0e38f) || p.alpha == 0.f);
    if (fold) alpha = af;
    else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int g = 0; g < 16; ++g) acc[i][j][g] = acc[i][j][g] * ia * ib;
    }
  }

  const int row_base = m0 + wm * TM * 32, col_base = n0 + wn * TN * 32;
  float cmax = 0.f;
  float* scr = reinterpret_cast<float*>(lds) + 4 + wave * EPI_FLOATS;      // per-wave transposition scratch (the stages are idle:
  // the K loop ended on a barrier); 4 floats: amax commit
  if (rescued) {                                                            // (workgroup-uniform) rows back to the tensor's scale
    float* rowfac = reinterpret_cast<float*>(lds) + 4 + NW * EPI_FLOATS;
    if (quad == 0) { rowfac[rl] = rf0; if (APASS == 2) rowfac[rl + BM / 2] = rf1; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const float f = rowfac[wm * TM * 32 + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * h];
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j][g] *= f;
      }
  }
  if constexpr (ROWSTATS) {
    static_assert(!ROWSTATS || (TN % 2 == 0 && TM % 2 == 0), "frame statistics ride on 64 x 64 accumulator blocks");
#pragma unroll
    for (int i = 0; i < TM; i += 2)
#pragma unroll
      for (int j = 0; j < TN; j += 2)
        epilogue_rowstats_block(p, acc[i][j], acc[i][j + 1], acc[i + 1][j], acc[i + 1][j + 1], row_base + i * 32, col_base + j * 32, lane, scr, cmax, alpha);
  } else {
    const unsigned long long seed = (p.seed && p.drop.thresh) ? *p.seed : 0ull;
    static_for<0, TM>([&](auto i) __attribute__((always_inline)) {
      const float4 rowsc = epilogue_row_scales(p, seed, row_base + i * 32, lane);
      static_for<0, TN>([&](auto j) __attribute__((always_inline)) {
        epilogue_tile(p, acc[i][j], row_base + i * 32, col_base + j * 32, lane, scr, 0, seed, cmax, rowsc, alpha);
      });
    });
  }
  amax_slot_commit_block(p.c_amax, cmax, reinterpret_cast<float*>(lds), cpeek);      // (the stages are idle after the K loop's last barrier)
}

// NST = LDS stages (see gemm_f16_body).
template <int TM, int TN, int WM, int WN, bool ROWSTATS, int MINB = 2, int NST = 2>
__global__ __launch_bounds__(64 * WM * WN, MINB) void gemm_f16_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) char lds[gemm_f16_lds_bytes<TM, TN, WM, WN, NST>()];
  gemm_f16_body<TM, TN, WM, WN, ROWSTATS, NST>(p, lds, blockIdx.x, gridDim.x);
}

// =====================================================================================================
// Weight gradients, dW[M,N] = A^T B with A = dy [K][M], B = x [K][N]: gemm_wgrad_wide_kernel's organisation (row-major
// staging, transposing LDS reads, split-K with one K-chunk per XCD, bias gradient from the staging registers) on two fp16
// planes per operand; both operands are split on the fly, each with the scale of its own amax slot.
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16x8 lds_read_tr_pair_h(const char* a, int second_off) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + second_off));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(f16x8, v);
}

constexpr int GEMM_WGRAD_F16_LDS_BYTES = 2 * (2 * 16 * 128 * 2 + 2 * 16 * 256 * 2);      // two stages of a 128 x 256 tile's two planes per operand

// workgroup (bx, by) of a (gx, splits + reduce rows) grid, on `lds` (GEMM_WGRAD_F16_LDS_BYTES): the kernel's body as a function, for
// launches that hold more than one problem (gemm_f16_group_kernel)
template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ void gemm_wgrad_f16_body(const GemmParams& p, char* lds, const int bx, const int by, const int gx) {
  constexpr int NW = WM * WN, THREADS = 64 * NW;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  static_assert(THREADS == 256 && BM == 128 && BN == 256, "staging maps are written for 256 threads on a 128 x 256 tile");
  constexpr int ROWA = BM * 2, ROWB = BN * 2;
  constexpr int A_PLANE = 16 * ROWA, B_PLANE = 16 * ROWB, A_BYTES = 2 * A_PLANE, B_BYTES = 2 * B_PLANE;
  constexpr int STAGE = A_BYTES + B_BYTES;
  static_assert(2 * STAGE == GEMM_WGRAD_F16_LDS_BYTES, "lds size formula");

  if (by >= p.splits) {
    // the extra grid rows: the PREVIOUS weight-gradient launch's split-K reduction (ReduceJob, gemm.h).  That launch is complete
    // (same stream), its partial slabs are at rest; these workgroups are HBM-bound and run beside this launch's MFMA-bound ones -
    // no launch of its own, no idle tail.  Same summation order as splitk_reduce_kernel: bit-identical.
    float cm = 0.f;
    splitk_reduce_body(p.prev, (by - p.splits) * gx + bx, p.prev.blocks, cm);
    return;
  }
  int z, tl;
  {
    const int tiles = gx;
    if (p.splits > 1 && (p.splits & 7) == 0) {
      const int lin = bx + tiles * by, xcd = lin & 7, slot = lin >> 3;
      const int g = slot / tiles;
      tl = slot - g * tiles; z = g * 8 + xcd;
    } else { z = by; tl = bx; }
  }
  const int tile_m = tl / p.tiles_n, tile_n = tl - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const float* A = p.A + (long long)z * p.K * p.lda;
  const float* B = p.B + (long long)z * p.K * p.ldb;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
  const int nk = p.K >> 4;
  const float sa = amax_scale(amax_slot_read(p.a_amax)), sb = amax_scale(amax_slot_read(p.b_amax));

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

  const int ka = t >> 5, cqa = t & 31, kb = t >> 6, cqb = t & 63;
  // global addresses = wave-uniform base (advanced per K-step and per row group by scalar adds) + one 32-bit lane offset per
  // operand: the loads take the scalar-base form and the loop has no 64-bit vector adds
  const char* a_base = reinterpret_cast<const char*>(A);
  const char* b_base = reinterpret_cast<const char*>(B);
  const unsigned int a_voff = (unsigned int)(((long long)ka * p.lda + min(m0 + 4 * cqa, p.M - 4)) * 4);
  const unsigned int b_voff = (unsigned int)(((long long)kb * p.ldb + min(n0 + 4 * cqb, p.N - 4)) * 4);
  const long long a_row8 = 32 * p.lda, b_row4 = 16 * p.ldb, a_step = 64 * p.lda, b_step = 64 * p.ldb;      // bytes
  const int a_dst = ka * ROWA + ((8 * cqa) ^ ((ka & 3) << 6));
  const int b_dst = A_BYTES + kb * ROWB + ((8 * cqb) ^ ((kb & 3) << 6));
  const int q = (lane >> 2) & 3, pp = lane & 3, cg = 16 * ((lane >> 4) & 1) + 4 * pp;
  int fa[TM], fb[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) fa[i] = (8 * h + q) * ROWA + (((wm * TM * 32 + i * 32 + cg) * 2) ^ (q << 6));
#pragma unroll
  for (int j = 0; j < TN; ++j) fb[j] = A_BYTES + (8 * h + q) * ROWB + (((wn * TN * 32 + j * 32 + cg) * 2) ^ (q << 6));

  const bool want_cs = p.colsum && tile_n == 0;
  f32x4 cs = {0.f, 0.f, 0.f, 0.f};
  // COLUMN WATCH.  A column m of dy is a ROW of dW; one scale per tensor leaves an output feature whose gradients lie 2^18 or more
  // below the tensor's bound with subnormal low terms.  Unlike gemm_f16_kernel's row guard this kernel does not repair such a
  // chunk in place (a second pass around this loop is not expressible without disturbing its hand-counted asynchronous loads);
  // it DETECTS it - every thread keeps the running |max| of the four columns of the A tile it stages (cm4) - and raises
  // p.range_flag, on which the host re-runs the weight gradient in the bf16x6 arithmetic (fp32's exponent range).
  f32x4 cm4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 ra[2], rb[4];
#define NPVP_G_LOAD(KT)                                                                                     \
  { const int kt_ = min((KT), nk - 1);                                                                      \
    const char* pa_ = a_base + (long long)kt_ * a_step; const char* pb_ = b_base + (long long)kt_ * b_step;  \
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ra[0]) : "v"(a_voff), "s"(pa_) : "memory");        \
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ra[1]) : "v"(a_voff), "s"(pa_ + a_row8) : "memory"); \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                        \
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rb[i_]) : "v"(b_voff), "s"(pb_ + i_ * b_row4) : "memory"); }
  // the loads are inline asm (scalar base + lane offset): waits by hand - issue order A0 A1 B0 B1 B2 B3
#define NPVP_G_WAIT(N, ...) asm volatile("s_waitcnt vmcnt(" #N ")" : __VA_ARGS__ :: "memory");
#define NPVP_G_STORE(DST, V, SC, PLANE)                                                                     \
  { f16x4 hi_, lo_; split_f16_scaled((V), (SC), hi_, lo_);                                                  \
    *reinterpret_cast<f16x4*>(DST) = hi_; *reinterpret_cast<f16x4*>((DST) + (PLANE)) = lo_; }
  // a row-group mask on A = dy [K][M]: the 16 rows of a K-step belong to ONE group (g1 % 16 == 0, checked by the launcher), so the
  // step's mask is one wave-uniform factor folded into the scale (and into the bias-gradient sums); am_ = mask of the tile in ra
  const unsigned long long aseed = p.adrop.thresh ? *p.seed : 0ull;
  const long long arow0 = (long long)z * p.K;
  float am_ = 1.f;
#define NPVP_G_AMASK(KT) { if (p.adrop.thresh) am_ = drop_spec_scale(p.adrop, aseed, arow0 + 16ll * min((KT), nk - 1), 0, 1); }
#define NPVP_G_STORE_A(ST) { const float sam_ = sa * am_;                                                                        \
    if (p.range_flag) cm4 = __builtin_elementwise_max(cm4, __builtin_elementwise_max(__builtin_elementwise_abs(ra[0]), __builtin_elementwise_abs(ra[1]))); \
    NPVP_G_STORE((ST) + a_dst, ra[0], sam_, A_PLANE) NPVP_G_STORE((ST) + a_dst + 8 * ROWA, ra[1], sam_, A_PLANE) }
#define NPVP_G_STORE_B(ST, I) NPVP_G_STORE((ST) + b_dst + (I) * 4 * ROWB, rb[I], sb, B_PLANE)

  NPVP_G_LOAD(0)
  NPVP_G_AMASK(0)
  NPVP_G_WAIT(0, "+v"(ra[0]), "+v"(ra[1]), "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]))
  if (want_cs) cs += (ra[0] + ra[1]) * am_;
  NPVP_G_STORE_A(lds)
  NPVP_G_STORE_B(lds, 0) NPVP_G_STORE_B(lds, 1) NPVP_G_STORE_B(lds, 2) NPVP_G_STORE_B(lds, 3)
  NPVP_G_LOAD(1)
  __syncthreads();

#define NPVP_G_STEP(KT, CUR, NXT)                                                                            \
  {                                                                                                          \
    const char* st_ = lds + (CUR) * STAGE;                                                                   \
    char* nx_ = lds + (NXT) * STAGE;                                                                         \
    NPVP_G_WAIT(3, "+v"(ra[0]), "+v"(ra[1]), "+v"(rb[0]))                                                    \
    NPVP_G_AMASK((KT) + 1)                            /* (ra holds tile KT+1) */                             \
    if (want_cs && (KT) + 1 < nk) { asm volatile("" ::: "memory"); cs += (ra[0] + ra[1]) * am_; }  /* (a real branch: only the first tile column sums) */ \
    f16x8 fa_[TM][2];                                                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_)                                                        \
      _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) fa_[i_][s_] = lds_read_tr_pair_h(st_ + fa[i_] + s_ * A_PLANE, 4 * ROWA); \
    _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) {                                                      \
      f16x8 fb_[2];                                                                                          \
      _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) fb_[s_] = lds_read_tr_pair_h(st_ + fb[j_] + s_ * B_PLANE, 4 * ROWB); \
      if (j_ == 0) { NPVP_G_STORE_A(nx_) NPVP_G_STORE_B(nx_, 0) }                                            \
      if (j_ == 1) { NPVP_G_WAIT(1, "+v"(rb[1]), "+v"(rb[2])) NPVP_G_STORE_B(nx_, 1) NPVP_G_STORE_B(nx_, 2) } \
      if (j_ == 2) { NPVP_G_WAIT(0, "+v"(rb[3])) NPVP_G_STORE_B(nx_, 3) NPVP_G_LOAD((KT) + 2) __builtin_amdgcn_sched_barrier(0); } \
      _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_[i_][1], fb_[0], acc[i_][j_], 0, 0, 0); \
      _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_[i_][0], fb_[1], acc[i_][j_], 0, 0, 0); \
      _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_[i_][0], fb_[0], acc[i_][j_], 0, 0, 0); \
    }                                                                                                        \
    /* the barrier orders LDS only: __syncthreads() would also wait (vmcnt(0)) for the loads of tile KT+2 issued above */ \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
    __builtin_amdgcn_s_barrier();                                                                            \
  }

  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    NPVP_G_STEP(kt, 0, 1)
    NPVP_G_STEP(kt + 1, 1, 0)
  }
  if (kt < nk) NPVP_G_STEP(kt, 0, 1)
  // (the clamped loads past the last tile still target ra / rb: TIED to the wait - a dead asm output may be given a register that
  //  the code between the load and this wait uses for something else, and the load then lands in it; see gemm_f16_kernel)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]) :: "memory");
  if (p.range_flag) {
    // column maxima of the chunk: 8 threads (ka = 0..7, all four waves) hold partial maxima of the same four columns
    float* red = reinterpret_cast<float*>(lds);            // (the stages are idle: the loop ended on a barrier)
    *reinterpret_cast<f32x4*>(red + ka * BM + 4 * cqa) = cm4;
    __syncthreads();
    bool flag = false;
    if (t < BM) {
      float m = red[t];
#pragma unroll
      for (int i = 1; i < 8; ++i) m = fmaxf(m, red[i * BM + t]);
      const int et = (int)((__float_as_uint(sa) >> 23) & 0xffu), e = (int)((__float_as_uint(m) >> 23) & 0xffu);
      const int d = 268 - et - e;                           // exponent distance below the tensor's bound (sa = 2^(141 - E_bound))
      flag = e >= 16 && e != 255 && d >= 18 && m0 + t < p.M;
    }
    const bool any = __builtin_amdgcn_ballot_w64(flag) != 0ull;
    if (any && lane == 0) atomicAdd(p.range_flag, 1u);
    __syncthreads();                                        // (`red` is about to be reused)
  }
#undef NPVP_G_STEP
#undef NPVP_G_WAIT
#undef NPVP_G_STORE_B
#undef NPVP_G_STORE_A
#undef NPVP_G_AMASK
#undef NPVP_G_STORE
#undef NPVP_G_LOAD

  if (want_cs) {
    float* red = reinterpret_cast<float*>(lds);
    *reinterpret_cast<f32x4*>(red + (t >> 5) * 128 + cqa * 4) = cs;
    __syncthreads();
    if (t < 128 && m0 + t < p.M) {
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) sum += red[i * 128 + t];
      store_colsum(p, (long long)z * p.M + m0 + t, sum);
    }
    __syncthreads();                     // `red` is about to become the epilogue's scratch
  }
  float* scr = reinterpret_cast<float*>(lds) + wave * EPI_FLOATS;
  static_assert(2 * STAGE >= NW * EPI_FLOATS * 4, "scratch");
  const float ia = pow2_recip(sa), ib = pow2_recip(sb);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = acc[i][j][g] * ia * ib;
  const unsigned long long seed = 0ull;
  const int row_base = m0 + wm * TM * 32, col_base = n0 + wn * TN * 32;
  float cmax = 0.f;
  const float4 no_rowsc = make_float4(1.f, 1.f, 1.f, 1.f);         // (a weight gradient's epilogue carries no mask)
  static_for<0, TM>([&](auto i) __attribute__((always_inline)) {
    static_for<0, TN>([&](auto j) __attribute__((always_inline)) {
      epilogue_tile(p, acc[i][j], row_base + i * 32, col_base + j * 32, lane, scr, z, seed, cmax, no_rowsc, p.alpha);
    });
  });
}

template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 2) void gemm_wgrad_f16_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) char lds[GEMM_WGRAD_F16_LDS_BYTES];
  gemm_wgrad_f16_body<TM, TN, WM, WN>(p, lds, blockIdx.x, blockIdx.y, gridDim.x);
}

// ---- a dgrad and the weight gradient of the same dy in ONE launch ------------------------------------------------------------------
// dx = dy W and dW += dy^T x both read dy and nothing of each other.  On an 8-clip shard (8 192 token rows) a dgrad is 256 tiles of
// 128 x 128 - one workgroup per CU, one wave per SIMD, the matrix pipes mostly idle behind barriers and LDS-DMA latency - and the
// weight gradient 256 split-K workgroups likewise.  The eager step overlaps them through the gradient stream; a step captured
// single-stream into a HIP graph (trainer.GraphedTrainStep: the host-independent form) has no second stream.  Here the two problems
// share a launch: workgroups [0, first[1]) run the dgrad body, the rest the weight-gradient body (its extra rows reduce the previous
// weight gradient's split-K slabs, as in gemm_wgrad_f16_kernel), so every CU holds one workgroup of each and the launch count of a
// shard's step drops by the 112 weight-gradient launches.  Block counts per problem are padded to multiples of 8: workgroups are
// dealt round-robin over the 8 XCDs, and both bodies map block index -> tile by XCD.
struct GemmGroup { GemmParams p[2]; int first[3]; int count[2]; int gx; };

template <int TM, int TN>
__global__ __launch_bounds__(256, 2) void gemm_f16_group_kernel(GemmGroup g) {
  constexpr int L0 = gemm_f16_lds_bytes<TM, TN, 2, 2, 3>();
  __shared__ __attribute__((aligned(16))) char lds[L0 > GEMM_WGRAD_F16_LDS_BYTES ? L0 : GEMM_WGRAD_F16_LDS_BYTES];
  const int b = blockIdx.x;
  if (b < g.first[1]) {
    if (b < g.count[0]) gemm_f16_body<TM, TN, 2, 2, false, 3>(g.p[0], lds, b, g.count[0]);
  } else {
    const int lb = b - g.first[1];
    if (lb < g.count[1]) {
      const int by = lb / g.gx;
      gemm_wgrad_f16_body<2, 4, 2, 2>(g.p[1], lds, lb - by * g.gx, by, g.gx);
    }
  }
}

// split count of the fp16 weight-gradient kernel: ~512 workgroups, >= 16 K-steps per split; 0 = shape not taken.  From 1 024 token
// rows (round 5; 4 096 before): the encoder of an 8-clip shard (1 024 - 2 048 rows) then takes the chained / fused route instead of
// the 128 x 128 bf16 kernel + a split-K reduction launch + a column-sum launch per weight gradient.
int f16_wgrad_splits(int M, int N, int K) {
  if ((K & 15) || M < 64 || N < 128 || K < 1024) return 0;
  const int tiles = ((M + 127) / 128) * ((N + 255) / 256);
  const int want = 512;                 // workgroups per launch (256 / 384 / 1024 measured within noise of it: DESIGN.md section 5)
  int s = (want + tiles - 1) / tiles;
  const int maxs = K / 256;
  if (s > maxs) s = maxs;
  if (s > 64) s = 64;
  if (s >= 8) s &= ~7;
  while (s > 1 && (K % (s * 16)) != 0) --s;
  return s < 1 ? 1 : s;
}

// blocks that reduce a [M][N] split-K result inside the next launch: a multiple of that launch's tile count (whole grid rows)
static int reduce_rows_for(const ReduceJob& j, int tiles) {
  if (!j.ws) return 0;
  const long long total4 = (long long)j.M * j.N / 4;
  long long want = (total4 + 255) / 256;
  if (want > 256) want = 256;
  if (want < 1) want = 1;
  return (int)((want + tiles - 1) / tiles);
}

bool launch_gemm_wgrad_f16(GemmParams& p, int splits, hipStream_t stream) {
  p.tiles_m = (p.M + 127) / 128;
  p.tiles_n = (p.N + 255) / 256;
  const int tiles = p.tiles_m * p.tiles_n, rr = reduce_rows_for(p.prev, tiles);
  p.prev.blocks = rr * tiles;
  dim3 grid(tiles, splits + rr), block(256);
  NPVP_LAUNCH((gemm_wgrad_f16_kernel<2, 4, 2, 2>), grid, block, 0, stream, p);
  return true;
}

// forward / dgrad with scaled fp16 planes: 1 = 128 x 256 tiles, 2 = 128 x 128 tiles (outputs that 128 x 256 tiles do not
// fill the chip with), 4 = 64 x 128 tiles (outputs of at most 256 tiles of 128 x 128), 3 = 128 x 64 tiles (the same for row counts
// that are not a multiple of 64, at most 128 tiles), 0 = shape not taken (the caller falls back to the three-term bf16 kernels WITHOUT planes)
int gemm_f16_variant(int M, int N, int K) {
  if ((K & 15) || (N & 7) || M < 128) return 0;
  const int tw = ((M + 127) / 128) * ((N + 255) / 256);
  if (N % 128 != 0 || tw >= 512) return 1;
  // up to 256 tiles of 128 x 128 (every [R x 512] output of an 8-clip shard, its whole encoder) half the chip or more would hold
  // one workgroup per CU - one wave per SIMD with nothing to hide a barrier or an LDS-DMA round trip behind: 64 x 128 tiles double
  // the workgroups at the SAME operand-split work per MFMA (the A rows a workgroup splits halve with its MFMAs; only the weight's
  // LDS-DMA, which costs no VALU, doubles).  Round 5, tools/gemm_bench.py over the five layer shapes, forward / dgrad TF: 8 192 rows
  // 259 -> 265, 4 096 rows 187 -> 212, 2 048 rows 114 -> 137 (there against round 4's 128 x 64 tiles, which double the A split
  // instead: profiles/r05_gemm_bench_tiles.txt); above 256 tiles no gain.
  if (M % 64 == 0 && ((M + 127) / 128) * (N / 128) <= 256) return 4;
  // up to 128 tiles of 128 x 128 (the encoder of an 8-clip shard: 1 024 .. 2 048 token rows) half the chip would idle: 128 x 64
  // tiles double the workgroups (R = 2 048: forward 18.8 -> 15.4 us at N = K = 512, 58 -> 49 us at K = 2 048; from 256 tiles on the
  // narrow tile loses - the A staging per MFMA doubles - profiles/r04_gemm_bench_narrow.txt)
  if (N % 64 == 0 && ((M + 127) / 128) * (N / 128) <= 128) return 3;
  return 2;
}

// the tiling of a forward / dgrad launch: -> variant (0 = not taken), p.tiles_* / p.colgroups set for it
static int prep_gemm_f16(GemmParams& p, int v128 = 0) {
  if (!p.b_pre || !p.a_amax || !p.b_amax || p.splits != 1 || p.colsum || ((uintptr_t)p.b_pre & 15) != 0) return 0;
  if (p.adrop.thresh && (p.adrop.mode != 1 || !p.seed)) return 0;
  int v = gemm_f16_variant(p.M, p.N, p.K);
  if (v == 4 && v128) v = 2;                         // (the caller wants 128 x 128 tiles where 64 x 128 would be taken)
  if (v == 0 || (p.rowstats && (p.N % 64 != 0 || p.M % 64 != 0))) return 0;
  const int bn = v == 1 ? 256 : (v == 3 ? 64 : 128);
  p.tiles_m = v == 4 ? (p.M + 63) / 64 : (p.M + 127) / 128;
  p.tiles_n = (p.N + bn - 1) / bn;
  p.colgroups = pick_colgroups((long long)p.N * p.K * 4, p.tiles_m, p.tiles_n);
  return v;
}

bool launch_gemm_f16(GemmParams& p, hipStream_t stream) {
  const int v = prep_gemm_f16(p);
  if (v == 0) return false;
  dim3 grid(p.tiles_m * p.tiles_n), block(256);
  if (v == 1) {
    if (p.rowstats) NPVP_LAUNCH((gemm_f16_kernel<2, 4, 2, 2, true>), grid, block, 0, stream, p);
    else NPVP_LAUNCH((gemm_f16_kernel<2, 4, 2, 2, false>), grid, block, 0, stream, p);
  } else if (v == 3 && !p.rowstats) {
    NPVP_LAUNCH((gemm_f16_kernel<2, 1, 2, 2, false, 2, 3>), grid, block, 0, stream, p);
  } else if (v == 4 && !p.rowstats) {
    NPVP_LAUNCH((gemm_f16_kernel<1, 2, 2, 2, false, 2, 3>), grid, block, 0, stream, p);
  } else {
    if (v == 4) { p.tiles_m = (p.M + 127) / 128; grid = dim3(p.tiles_m * p.tiles_n); p.colgroups = pick_colgroups((long long)p.N * p.K * 4, p.tiles_m, p.tiles_n); }
    if (v == 3) { p.tiles_n = (p.N + 127) / 128; grid = dim3(p.tiles_m * p.tiles_n); p.colgroups = pick_colgroups((long long)p.N * p.K * 4, p.tiles_m, p.tiles_n); }
    if (p.rowstats) NPVP_LAUNCH((gemm_f16_kernel<2, 2, 2, 2, true, 2, 3>), grid, block, 0, stream, p);
    else NPVP_LAUNCH((gemm_f16_kernel<2, 2, 2, 2, false, 2, 3>), grid, block, 0, stream, p);
  }
  return true;
}

// ---- amax of a tensor: slot[32] = max(slot, |x|) --------------------------------------------------------------------
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long long n4, long long ldq, long long cols4, float* slot) {
  __shared__ float ared[4];
  const unsigned int peek = amax_peek_block(slot);
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / cols4, c = i - row * cols4;
    m = amax4(m, ld4(x + 4 * (row * ldq + c)));
  }
  amax_slot_commit_block(slot, m, ared, peek);
}

// ---- weights -> scaled fp16 planes: F[term][K/8][N][8 over k], D[term][N/8][K][8 over n] -----------------------------
struct SplitDescH { const float* w; long long ld; long long N; long long K; _Float16* F; _Float16* D; float* amax; long long pad; };

__global__ __launch_bounds__(256) void weights_amax_kernel(const SplitDescH* __restrict__ desc, SplitDescH one) {
  const SplitDescH d = desc ? desc[blockIdx.y] : one;
  const int K4 = (int)d.K / 4;
  const long long total = d.N * K4;
  __shared__ float ared[4];
  const unsigned int peek = amax_peek_block(d.amax);
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / K4; const int k4 = (int)(i - n * K4);
    m = amax4(m, ld4(d.w + n * d.ld + 4 * k4));
  }
  amax_slot_commit_block(d.amax, m, ared, peek);
}

__global__ __launch_bounds__(256) void split_weights_f16_kernel(const SplitDescH* __restrict__ desc, SplitDescH one) {
  const SplitDescH d = desc ? desc[blockIdx.y] : one;
  const int N = (int)d.N, K = (int)d.K;
  const long long slots = (long long)N * K / 8, plane = (long long)N * K;
  const float s = amax_scale(amax_slot_read(d.amax));
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * slots; i += (long long)gridDim.x * blockDim.x) {
    float v[8];
    _Float16* out;
    if (i < slots) {
      if (!d.F) continue;
      const int n = (int)(i % N), kb = (int)(i / N);
      const float4 a = ld4(d.w + (long long)n * d.ld + kb * 8), b = ld4(d.w + (long long)n * d.ld + kb * 8 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      out = d.F + i * 8;
    } else {
      if (!d.D) continue;
      const long long j = i - slots;
      const int k = (int)(j % K), nb = (int)(j / K);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = d.w[(long long)(nb * 8 + e) * d.ld + k];
      out = d.D + j * 8;
    }
    f16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float x = v[e] * s; hi[e] = (_Float16)x; lo[e] = (_Float16)(x - (float)hi[e]); }
    *reinterpret_cast<f16x8*>(out) = hi;
    *reinterpret_cast<f16x8*>(out + plane) = lo;
  }
}

}  // namespace npvp

using namespace npvp;

// See include/npvp_hip.h.
extern "C" int npvp_amax(const float* x, long long rows, long long cols, long long ld, float* slot, hipStream_t stream) {
  NPVP_CHECK_ARG(x && slot && rows > 0 && cols > 0 && cols % 4 == 0 && ld % 4 == 0 && ld >= cols, "amax: cols and ld must be multiples of 4");
  NPVP_CHECK_ARG(((uintptr_t)x % 16) == 0, "amax: x must be 16-byte aligned");
  const long long n4 = rows * (cols / 4);
  long long blocks = (n4 + 256 * 8 - 1) / (256 * 8);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  NPVP_LAUNCH(amax_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, n4, ld / 4, cols / 4, slot);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

namespace npvp {
// The amax tables are zeroed by a KERNEL, not by hipMemsetAsync: inside a captured step a memset becomes a memset node, and
// memset nodes are what the ROCm 7.2 prepared-packet replay does not execute reliably (profiles/r06_graph_alloc_hazard.txt).
__global__ __launch_bounds__(256) void zero_words_kernel(unsigned* __restrict__ p, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = 0u;
}
}  // namespace npvp

static hipError_t zero_fill(void* p, size_t bytes, hipStream_t stream) {
  const long long n = (long long)(bytes / 4);
  long long blocks = (n + 255) / 256; if (blocks > 1024) blocks = 1024; if (blocks < 1) blocks = 1;
  NPVP_LAUNCH(zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (unsigned*)p, n);
  return hipGetLastError();
}

extern "C" int npvp_split_weights_f16(const void* desc, int count, void* amax_table, long long amax_bytes, hipStream_t stream) {
  NPVP_CHECK_ARG(desc && count > 0, "split_weights_f16: empty table");
  if (amax_table && amax_bytes > 0) {
    if (zero_fill(amax_table, (size_t)amax_bytes, stream) != hipSuccess) { npvp_set_error("split_weights_f16: zero fill failed"); return NPVP_ERR_LAUNCH; }
  }
  SplitDescH none = {};
  NPVP_LAUNCH(weights_amax_kernel, dim3(16, count), dim3(256), 0, stream, (const SplitDescH*)desc, none);
  NPVP_CHECK_LAUNCH();
  NPVP_LAUNCH(split_weights_f16_kernel, dim3(128, count), dim3(256), 0, stream, (const SplitDescH*)desc, none);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

extern "C" int npvp_split_weight_f16(const float* w, long long ld, int N, int K, void* F, void* D, float* amax_slot, hipStream_t stream) {
  NPVP_CHECK_ARG(N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0 && ld % 4 == 0, "split_weight_f16: N, K must be multiples of 8");
  NPVP_CHECK_ARG(((uintptr_t)w % 16) == 0 && amax_slot, "split_weight_f16: w must be 16-byte aligned, amax_slot non-null");
  if (zero_fill(amax_slot, AMAX_WORDS * AMAX_STRIDE * 4, stream) != hipSuccess) { npvp_set_error("split_weight_f16: zero fill failed"); return NPVP_ERR_LAUNCH; }
  SplitDescH one = {w, ld, N, K, (_Float16*)F, (_Float16*)D, amax_slot, 0};
  NPVP_LAUNCH(weights_amax_kernel, dim3(16, 1), dim3(256), 0, stream, (const SplitDescH*)nullptr, one);
  NPVP_CHECK_LAUNCH();
  NPVP_LAUNCH(split_weights_f16_kernel, dim3(128, 1), dim3(256), 0, stream, (const SplitDescH*)nullptr, one);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}

// ---- weight gradients whose split-K reduction rides in the NEXT weight-gradient launch (see include/npvp_hip.h) -------------------
static_assert(sizeof(ReduceJob) == 64, "npvp_reduce_job_t (include/npvp_hip.h) mirrors this layout");

extern "C" int npvp_wgrad_f16_chainable(int M, int N, int K) {
  return (M % 4 == 0 && N % 4 == 0 && f16_wgrad_splits(M, N, K) > 1) ? 1 : 0;
}

extern "C" long long npvp_wgrad_f16_chain_workspace_bytes(int M, int N, int K) {
  const int s = f16_wgrad_splits(M, N, K);
  return s > 1 ? ((long long)s * M * N + (long long)s * M) * 4 : 0;
}

extern "C" int npvp_wgrad_f16_chained(int M, int N, int K, const float* dy, long long lda, const float* x, long long ldb, float* dw,
                                      long long ldc, float* db, int accumulate, const float* a_amax, const float* b_amax,
                                      unsigned int* range_flag, float adrop_p, int adrop_g1, int adrop_g2, unsigned int adrop_salt,
                                      const unsigned long long* seed, const void* prev_job, void* my_job, void* workspace,
                                      long long ws_bytes, hipStream_t stream) {
  NPVP_CHECK_ARG(npvp_wgrad_f16_chainable(M, N, K), "wgrad_f16_chained: shape not taken (npvp_wgrad_f16_chainable tells)");
  NPVP_CHECK_ARG(dy && x && dw && a_amax && b_amax && my_job, "wgrad_f16_chained: null operand / amax slot / job");
  NPVP_CHECK_ARG(((uintptr_t)dy % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dw % 16) == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0,
                 "wgrad_f16_chained: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
  NPVP_CHECK_ARG(!db || ((uintptr_t)db % 16) == 0, "wgrad_f16_chained: the bias gradient must be 16-byte aligned");
  NPVP_CHECK_ARG(workspace && ws_bytes >= npvp_wgrad_f16_chain_workspace_bytes(M, N, K), "wgrad_f16_chained: workspace too small");
  NPVP_CHECK_ARG(adrop_p >= 0.f && adrop_p < 0.5f && (adrop_p == 0.f || (seed && adrop_g1 > 0 && adrop_g2 > 0 && adrop_g1 % 16 == 0)),
                 "wgrad_f16_chained: adrop needs a device seed and groups of a multiple of 16 rows");
  const int sh = f16_wgrad_splits(M, N, K);
  GemmParams p = {};
  p.A = dy; p.B = x; p.lda = lda; p.ldb = ldb; p.M = M; p.N = N; p.K = K / sh; p.alpha = 1.f;
  p.C = (float*)workspace; p.ldc = N; p.splits = sh; p.colgroups = 1; p.accum = accumulate ? 1 : 0;
  p.colsum = db ? (float*)workspace + (long long)sh * M * N : nullptr;
  p.seed = seed;
  p.drop = make_drop_spec(0.f, 0u, 0, 1, 1);
  p.adrop = make_drop_spec(adrop_p, adrop_salt, 1, adrop_g1, adrop_g2);
  p.a_amax = a_amax; p.b_amax = b_amax; p.range_flag = range_flag;
  if (prev_job) p.prev = *reinterpret_cast<const ReduceJob*>(prev_job);
  launch_gemm_wgrad_f16(p, sh, stream);
  NPVP_CHECK_LAUNCH();
  ReduceJob mine = {(const float*)workspace, dw, ldc, M, N, sh, accumulate ? 1 : 0, 1.f, 0, db ? p.colsum : nullptr, db};
  *reinterpret_cast<ReduceJob*>(my_job) = mine;
  return NPVP_OK;
}

// ---- dgrad + weight gradient of one linear layer in ONE launch (gemm_f16_group_kernel; include/npvp_hip.h) ---------------------------
extern "C" int npvp_linear_bwd_f16_takes(int R, int N, int K) {
  const int v = gemm_f16_variant(R, K, N);
  return ((v == 2 || v == 3 || v == 4) && npvp_wgrad_f16_chainable(N, K, R)) ? 1 : 0;
}

extern "C" int npvp_linear_bwd_f16(int R, int N, int K, const float* dy, long long ldy, const float* dy_amax, const void* w_planes_d,
                                   const float* w_amax, float* dx, long long ldx, int act, const float* aux_in, const float* residual,
                                   long long ldr, float drop_p, int drop_mode, int drop_g1, int drop_g2, unsigned int drop_salt,
                                   float* dx_amax, const float* x, long long ldxx, const float* x_amax, float* dw, long long ldw, float* db,
                                   unsigned int* range_flag, float adrop_p, int adrop_g1, int adrop_g2, unsigned int adrop_salt,
                                   const unsigned long long* seed, const void* prev_job, void* my_job, void* workspace,
                                   long long ws_bytes, hipStream_t stream) {
  NPVP_CHECK_ARG(npvp_linear_bwd_f16_takes(R, N, K), "linear_bwd_f16: shape not taken (npvp_linear_bwd_f16_takes tells)");
  NPVP_CHECK_ARG(dy && dy_amax && w_planes_d && w_amax && dx && x && x_amax && dw && my_job, "linear_bwd_f16: null operand / amax slot / job");
  NPVP_CHECK_ARG(((uintptr_t)dy % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dw % 16) == 0 && ((uintptr_t)dx % 16) == 0 &&
                 ((uintptr_t)w_planes_d % 16) == 0 && ldy % 4 == 0 && ldxx % 4 == 0 && ldw % 4 == 0 && ldx % 4 == 0,
                 "linear_bwd_f16: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
  NPVP_CHECK_ARG((!db || ((uintptr_t)db % 16) == 0) && (!residual || (((uintptr_t)residual % 16) == 0 && ldr % 4 == 0)) &&
                 (!aux_in || ((uintptr_t)aux_in % 16) == 0), "linear_bwd_f16: db / residual / aux_in must be 16-byte aligned");
  NPVP_CHECK_ARG(act == 0 || ((act == 3 || act == 4) && aux_in), "linear_bwd_f16: act must be 0, or 3 / 4 with aux_in");
  NPVP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed), "linear_bwd_f16: dropout p out of range or no seed");
  NPVP_CHECK_ARG(adrop_p >= 0.f && adrop_p < 0.5f && (adrop_p == 0.f || (seed && adrop_g1 > 0 && adrop_g2 > 0 && adrop_g1 % 16 == 0)),
                 "linear_bwd_f16: adrop needs a device seed and groups of a multiple of 16 rows");
  NPVP_CHECK_ARG(workspace && ws_bytes >= npvp_wgrad_f16_chain_workspace_bytes(N, K, R), "linear_bwd_f16: workspace too small");
  GemmGroup g = {};
  // problem 0: dx[R][K] = epilogue((mask) dy[R][N] . W[N][K])  - what npvp_gemm_f32(a_kc = 1, b_kc = 0, M = R, N = K, K = N) builds
  GemmParams& d = g.p[0];
  d.A = dy; d.lda = ldy; d.C = dx; d.ldc = ldx; d.M = R; d.N = K; d.K = N; d.alpha = 1.f; d.act = act; d.aux_in = aux_in;
  d.residual = residual; d.ldr = ldr; d.seed = seed; d.splits = 1;
  d.drop = make_drop_spec(drop_p, drop_salt, drop_mode, drop_g1, drop_g2);
  d.adrop = make_drop_spec(adrop_p, adrop_salt, 1, adrop_g1, adrop_g2);
  d.b_pre = w_planes_d; d.b_pre_plane = (long long)N * K;
  d.a_amax = dy_amax; d.b_amax = w_amax; d.c_amax = dx_amax;
  // problem 1: dW[N][K] (+)= dy^T x over R rows, split-K, reduction handed to the next launch (npvp_wgrad_f16_chained)
  const int sh = f16_wgrad_splits(N, K, R);
  // the dgrad's tiles: 64 x 128 where the launch then still fits the chip's 512 workgroup slots with the weight gradient's
  // workgroups beside it, else 128 x 128 (8 192 rows: 256 + 256 workgroups in one round beat 512 + 256 in one and a half -
  // profiles/r05_linear_bwd_bench.txt)
  const int t128 = ((R + 127) / 128) * ((K + 127) / 128), wwg = ((N + 127) / 128) * ((K + 255) / 256) * sh;
  const int v = prep_gemm_f16(d, 2 * t128 + wwg > 512 ? 1 : 0);
  NPVP_CHECK_ARG(v == 2 || v == 3 || v == 4, "linear_bwd_f16: the dgrad is not a small-tile launch");
  GemmParams& w = g.p[1];
  w.A = dy; w.B = x; w.lda = ldy; w.ldb = ldxx; w.M = N; w.N = K; w.K = R / sh; w.alpha = 1.f;
  w.C = (float*)workspace; w.ldc = K; w.splits = sh; w.colgroups = 1; w.accum = 1;
  w.colsum = db ? (float*)workspace + (long long)sh * N * K : nullptr;
  w.seed = seed;
  w.drop = make_drop_spec(0.f, 0u, 0, 1, 1);
  w.adrop = d.adrop;
  w.a_amax = dy_amax; w.b_amax = x_amax; w.range_flag = range_flag;
  if (prev_job) w.prev = *reinterpret_cast<const ReduceJob*>(prev_job);
  w.tiles_m = (N + 127) / 128; w.tiles_n = (K + 255) / 256;
  const int tiles = w.tiles_m * w.tiles_n, rr = reduce_rows_for(w.prev, tiles);
  w.prev.blocks = rr * tiles;
  g.count[0] = d.tiles_m * d.tiles_n;
  g.first[0] = 0; g.first[1] = (g.count[0] + 7) & ~7;
  g.count[1] = tiles * (sh + rr);
  g.first[2] = g.first[1] + ((g.count[1] + 7) & ~7);
  g.gx = tiles;
  if (v == 2) NPVP_LAUNCH((gemm_f16_group_kernel<2, 2>), dim3(g.first[2]), dim3(256), 0, stream, g);
  else if (v == 4) NPVP_LAUNCH((gemm_f16_group_kernel<1, 2>), dim3(g.first[2]), dim3(256), 0, stream, g);
  else NPVP_LAUNCH((gemm_f16_group_kernel<2, 1>), dim3(g.first[2]), dim3(256), 0, stream, g);
  NPVP_CHECK_LAUNCH();
  ReduceJob mine = {(const float*)workspace, dw, ldw, N, K, sh, 1, 1.f, 0, db ? w.colsum : nullptr, db};
  *reinterpret_cast<ReduceJob*>(my_job) = mine;
  return NPVP_OK;
}

__global__ __launch_bounds__(256) void splitk_reduce_job_kernel(ReduceJob j) {
  float cm = 0.f;
  splitk_reduce_body(j, blockIdx.x, gridDim.x, cm);
}

// many split-K reductions in one launch: the records travel in the kernel's argument block (cf. npvp_sum_rows_multi)
constexpr int RJ_MAX = 32;
struct ReduceBatch { ReduceJob j[RJ_MAX]; int first[RJ_MAX + 1]; int n; };
__global__ __launch_bounds__(256) void splitk_reduce_multi_kernel(ReduceBatch b) {
  int lo = 0, hi = b.n - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (b.first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
  float cm = 0.f;
  splitk_reduce_body(b.j[lo], (int)blockIdx.x - b.first[lo], b.first[lo + 1] - b.first[lo], cm);
}

extern "C" int npvp_splitk_reduce_multi(const void* jobs, int n, hipStream_t stream) {
  NPVP_CHECK_ARG(jobs && n > 0, "splitk_reduce_multi: no jobs");
  const ReduceJob* J = reinterpret_cast<const ReduceJob*>(jobs);
  for (int at = 0; at < n; at += RJ_MAX) {
    ReduceBatch b;
    b.n = n - at < RJ_MAX ? n - at : RJ_MAX;
    int blocks = 0;
    for (int i = 0; i < b.n; ++i) {
      b.j[i] = J[at + i];
      NPVP_CHECK_ARG(b.j[i].ws && b.j[i].out && b.j[i].M > 0 && b.j[i].N > 0 && b.j[i].splits > 0, "splitk_reduce_multi: empty job");
      const long long total4 = (long long)b.j[i].M * b.j[i].N / 4;
      int nb = (int)((total4 + 255) / 256); if (nb > 256) nb = 256;
      b.first[i] = blocks; blocks += nb;
    }
    b.first[b.n] = blocks;
    NPVP_LAUNCH(splitk_reduce_multi_kernel, dim3(blocks), dim3(256), 0, stream, b);
    NPVP_CHECK_LAUNCH();
  }
  return NPVP_OK;
}

extern "C" int npvp_splitk_reduce_job(const void* job, hipStream_t stream) {
  NPVP_CHECK_ARG(job, "splitk_reduce_job: null job");
  const ReduceJob j = *reinterpret_cast<const ReduceJob*>(job);
  NPVP_CHECK_ARG(j.ws && j.out && j.M > 0 && j.N > 0 && j.splits > 0, "splitk_reduce_job: empty job");
  const long long total4 = (long long)j.M * j.N / 4;
  int blocks = (int)((total4 + 255) / 256); if (blocks > 2048) blocks = 2048;
  NPVP_LAUNCH(splitk_reduce_job_kernel, dim3(blocks), dim3(256), 0, stream, j);
  NPVP_CHECK_LAUNCH();
  return NPVP_OK;
}
