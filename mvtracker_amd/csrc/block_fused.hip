// Fused "rest of a transformer block" for the updater on the bf16 matrix cores (cotracker2/blocks.py:297-300,
// 334-337): everything that follows the attention of an AttnBlock / CrossAttnBlock, in ONE kernel, in place:
//     x += att . Wo^T + bo                                    (attention output projection + residual)
//     x += W2 . gelu_tanh(W1 . LayerNorm(x) + b1) + b2        (MLP + residual)
//     y  = LayerNorm(x)[* lnw + lnb] . Wn^T + bn              (optional: the NEXT block's q / kv / qkv projection)
// Unfused this is 6-7 launches (GEMM, LayerNorm, GEMM, GEMM, LayerNorm, GEMM) whose intermediates round-trip through
// HBM; for the 768 virtual-token rows each of them is pure launch / fill latency.
//
// Structure: a workgroup owns 128 token rows (four 32-row MFMA blocks) and runs 8 waves.  Activations live in LDS
// as bf16 ([128][K+8] images, conflict-free ds_read_b128); every weight matrix is streamed from global / L2 exactly
// once per workgroup, straight into MFMA A-operand registers (lane (r,h) reads 16 B of row n0+r at k0+8h: the eight
// k-steps that share a 128-B line hit L1).  All GEMMs are computed transposed, out^T (n x m) = W (n x k) . act^T, so a
// weight fragment feeds four MFMAs (the four token blocks) and the accumulator holds token m on the lane and
// channel n in the registers: LayerNorm statistics are per-lane sums + one cross-wave exchange through LDS.
//   out-proj / MLP-fc2 / next-proj : wave w owns output channel block(s) n = w (, w+8, ...) x 4 token blocks
//   MLP-fc1 (per 128 hidden units) : wave (jb = w&3, mp = w>>2) owns hidden block jb x token blocks 2mp, 2mp+1;
//                                    gelu(H) is written to LDS as bf16 [m][j] for fc2 (double buffered, one barrier
//                                    per chunk)
#include "common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Workgroup barrier that orders LDS only.  The waves of a workgroup exchange data through LDS alone, so a barrier has to wait for
// the wave's LDS operations -- not, as __syncthreads() does (s_waitcnt vmcnt(0) lgkmcnt(0) before s_barrier), for every global load
// in flight: the 16 weight fragments of the queue, the tile's x rows and the affine parameters are requested ahead of barriers on
// purpose, and draining them at each one put their memory round trips back on the critical path.
__device__ __forceinline__ void lds_barrier() {
#ifdef MVT_FULL_BARRIERS  // (A/B builds)
  __syncthreads();
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#endif
}

constexpr int NT = 512;   // threads (8 waves)
constexpr int C = 256;    // hidden size of the updater
constexpr int LDX = C + 8;
// NMB = 32-row token blocks per workgroup: 2 (64 rows) for the 12k point rows, 1 (32 rows) for the 768 virtual-token
// rows, where the work has to be spread over as many CUs as possible.  (128-row workgroups leave most CUs idle at
// M = 12288 and measured 1.6x slower.)
template <int NMB> struct Cfg {
  static constexpr int BM = 32 * NMB;
  static constexpr int HC = 256;                         // MLP hidden units per chunk
  static constexpr int JW = HC / 32;                     // waves across the hidden blocks of a chunk (4 or 8)
  static constexpr int NM1 = NMB / (8 / JW);             // token blocks per wave in fc1 (2 or 1)
  static constexpr int LDH = HC + 8;                     // bf16 row stride of the H buffers
  static constexpr int LDA = 296;                        // attention tile row stride (Ko = 288)
  static_assert(BM * LDA <= 2 * BM * LDH, "attention tile fits the H buffers");
};
constexpr int YLD = 40;   // bf16 row stride of the projection staging tile (32 columns + 8 pad)
constexpr int FS = 512;   // elements per (32-row block, k-step) weight fragment: [64 lanes][8 bf16], see mvt_pack_frag_bf16

// Diagnostic build only (-DMVT_STAMPS, tools/stamp_block.py): s_memtime at the phase boundaries of workgroups 0 and 100, one
// slot per (workgroup, wave, stamp), in a buffer of its own.  No stamp executes in the shipped library.
#ifdef MVT_STAMPS
__device__ unsigned long long mvt_stamp_buf[2 * 8 * 64];
// -DMVT_STAMP_SEL="(NMB == 1 && MODE == 2 && ATT == 0)" restricts the stamps to one instantiation (a whole updater call runs many)
#ifndef MVT_STAMP_SEL
#define MVT_STAMP_SEL true
#endif
#ifndef MVT_STAMP_WG1
#define MVT_STAMP_WG1 100
#endif
#define STAMP(i)                                                                                                  \
  do {                                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    unsigned long long t_;                                                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    if (lane == 0 && MVT_STAMP_SEL && (bx == 0 || bx == MVT_STAMP_WG1) && by == 0 && bz == 0) /* (logical block indices) */ \
      mvt_stamp_buf[((bx ? 1 : 0) * 8 + wave) * 64 + (i)] = t_;                                                   \
  } while (0)
#else
#define STAMP(i) \
  do {           \
  } while (0)
#endif

#if defined(MVT_STAMPS) && defined(MVT_ASTAMPS)  // (stamps inside attn_compute: only with the default MVT_STAMP_SEL)
#define ASTAMP(i)                                       \
  do {                                                  \
    const int wave = threadIdx.x >> 6;                  \
    const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z; \
    STAMP(i);                                           \
  } while (0)
#else
#define ASTAMP(i) \
  do {            \
  } while (0)
#endif

struct BlockArgs {
  float* x;                  // [M][ldx] tokens, updated in place
  int ldx;
  const float* att;          // [M][ldatt] attention output (Ko columns) or null: skip the output projection
  int ldatt, Ko;
  const unsigned short* wo;  // fragment-major bf16 (mvt_pack_frag_bf16) of [C][Ko]
  const float* bo;
  int ldwo;
  const unsigned short* w1;  // fragment-major bf16 of [H][C]
  const float* b1;
  const unsigned short* w2;  // fragment-major bf16 of [C][H]
  const float* b2;
  int ldw1, ldw2, H;
  mvt_block_next next[MVT_BLOCK_MAX_NEXT];    // follow-up projections of LayerNorm(x)
  int n_next;
  int att_bf16;              // att is a bf16 tensor
  long long M;
  float* ws;                 // split path (small M): [M][C] x after the projection, then [H/256][M][C] partial MLP outputs
  // in-kernel attention (ATT != 0): bf16 q / k / v operands, see mvt_block_attn
  const unsigned short* aq;
  const unsigned short* ak;
  const unsigned short* av;
  int ldaq, ldakv, S, nkeys, bmv;
  const float* parts;   // ATT 3: key-split attention partials of mvt_attention_bf16(MVT_ATTN_PARTIALS_ONLY)
  int nsplit;
  // ATT 4 (MODE 3): x itself is produced here -- rows < Mp: tokens . Win^T + bin; rows >= Mp: the learned virtual token
  const float* tokx;          // [Mp][ldtok] fp32 token matrix
  const unsigned short* win;  // fragment-major bf16 of [C][37 * 16]
  const float* bin;
  const float* virt;          // [n_virtual][C]
  int ldtok;
  long long Mp;
  // ... or the token rows are ASSEMBLED here (tokx == null; mvtracker.py:374-387, the arithmetic of token_assemble_kernel):
  const float* t_coords;    // [n][S][3]
  const float* t_fcorr;     // [n][S][Fc]
  const float* t_ffeats;    // [n][S][Cf]
  const float* t_maskvis;   // [n][S][2]
  const float* t_pos;       // [n][D]
  const float* t_time;      // [S][D]
  int t_E, t_Fc, t_Cf, t_D;
  // ATT 6: the 64 context tokens of the per-frame attention are the rows of a split-path block whose pass 2 was DEFERRED to this
  // kernel (mvt_block_ctx): their x is finished here from the partial sums, LayerNorm + k|v projection feed the attention from LDS
  const float* c_ws;        // split-path workspace of the context block: [c_nch + 1][S * 2 tiles][8 waves][4][64 lanes][4]
  const float* c_b2;        // its fc2 bias
  float* c_x;               // the context rows of the residual stream (row = token * S + frame), written by tile 0 of every frame
  int c_ldx, c_nch;
  mvt_block_next c_kv;      // LayerNorm + k|v projection of the context (N = 576; y unused: k|v never leave the CU)
  mvt_block_next c_next;    // optional (w == null: none) further projection of the context rows, written to y
};

// gelu_tanh on a pair: x * sigmoid(2k), k = sqrt(2/pi) (x + 0.044715 x^3), as x / (1 + exp2(x (c0 + c1 x^2))) with
// c0 = -2 sqrt(2/pi) log2(e), c1 = 0.044715 c0.  Packed fp32 ops (v_pk_mul / v_pk_fma / v_pk_add) carry two values per
// instruction; only v_exp_f32 / v_rcp_f32 are per value: ~26 instead of ~46 issue cycles per value -- the fc1 epilogue was VALU-bound
// (in-kernel stamps: 3.2 k of the 8.7 k cycles of an MLP chunk).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_tanh_pk(f32x2 x) {
  const f32x2 c0 = {-2.3022082f, -2.3022082f}, c1 = {-0.10294324f, -0.10294324f}, one = {1.0f, 1.0f};
  const f32x2 a = (x * x * c1 + c0) * x;
  f32x2 e;
  e[0] = __builtin_amdgcn_exp2f(a[0]);
  e[1] = __builtin_amdgcn_exp2f(a[1]);
  const f32x2 d = e + one;
  f32x2 rc;
  rc[0] = __builtin_amdgcn_rcpf(d[0]);
  rc[1] = __builtin_amdgcn_rcpf(d[1]);
  return x * rc;
}

__device__ __forceinline__ bf16x8 ldg_frag(const unsigned short* p) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
}
__device__ __forceinline__ bf16x8 lds_frag(const unsigned short* p) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
}

// acc[mb] += W[n0 + r][k] * Act[mb*32 + r'][k] for k in [0, 16*KS): W from global (wrow -> W[n0+r][8h]),
// Act from LDS (arow -> Act[r][8h], row stride lda).  Weight fragments are fetched four k-steps ahead.
template <int KS, int NMB>
__device__ __forceinline__ void gemm_wt(f32x16 (&acc)[NMB], const unsigned short* wrow, const unsigned short* arow, int lda,
                                        int mb0) {
  constexpr int PF = (KS % 8 == 0) ? 8 : 4;
  bf16x8 wq[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) wq[i] = ldg_frag(wrow + (i < KS ? i : 0) * FS);
  // groups of four k-steps (rolled: a fully unrolled body makes the scheduler hoist every LDS read and spill)
#pragma unroll 1
  for (int g = 0; g < KS / PF; ++g) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int ks = g * PF + j;
      const bf16x8 wa = wq[j];
      wq[j] = ldg_frag(wrow + (ks + PF < KS ? ks + PF : KS - 1) * FS);
#pragma unroll
      for (int i = 0; i < NMB; ++i) {
        const bf16x8 xb = lds_frag(arow + (mb0 + i) * 32 * lda + ks * 16);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, xb, acc[i], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < KS % PF; ++j) {
    const int ks = (KS / PF) * PF + j;
#pragma unroll
    for (int i = 0; i < NMB; ++i) {
      const bf16x8 xb = lds_frag(arow + (mb0 + i) * 32 * lda + ks * 16);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[j], xb, acc[i], 0, 0, 0);
    }
  }
}

// (MVT_ABL_NOLOAD / NOLDS / NOMFMA / NOGELU: ablation builds -- a phase compiled out to see what it costs, tools/time_block.py;
//  never defined in the shipped library.)
// Same product with a persistent weight-fragment queue: wq holds the first PFQ fragments of this stream on entry and the
// first PFQ fragments of the NEXT stream (nxt -> its row, 8h already applied) on exit, so the global-load latency of
// every GEMM call hides under the previous call instead of being exposed at its start.  KS % PFQ == 0.
constexpr int PFQ = 16;
template <int KS, int NMB>
__device__ __forceinline__ void gemm_wq(f32x16 (&acc)[NMB], bf16x8 (&wq)[PFQ], const unsigned short* wrow, const unsigned short* nxt,
                                        const unsigned short* arow, int lda, int mb0) {
  static_assert(KS == PFQ, "every GEMM segment is exactly one queue length");
  // Activation fragments are read from LDS one group of GK k-steps ahead of the MFMAs that use them (double-buffered
  // registers): the ~130-cycle LDS latency hides under the previous group's MFMAs instead of being paid per group.
  constexpr int GK = NMB >= 2 ? 2 : 4;
  bf16x8 xa[GK][NMB], xn[GK][NMB];
#pragma unroll
  for (int j = 0; j < GK; ++j)
#pragma unroll
    for (int i = 0; i < NMB; ++i) xa[j][i] = lds_frag(arow + (mb0 + i) * 32 * lda + j * 16);
#pragma unroll
  for (int g = 0; g < KS / GK; ++g) {
    if (g + 1 < KS / GK) {
#pragma unroll
      for (int j = 0; j < GK; ++j)
#pragma unroll
        for (int i = 0; i < NMB; ++i) xn[j][i] = lds_frag(arow + (mb0 + i) * 32 * lda + ((g + 1) * GK + j) * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < GK; ++j) {
      const int ks = g * GK + j;
      const bf16x8 wa = wq[ks];
#ifndef MVT_ABL_NOLOAD
      wq[ks] = ldg_frag(nxt + ks * FS);
#endif
#pragma unroll
      for (int i = 0; i < NMB; ++i) {
#ifdef MVT_ABL_NOLDS
        const bf16x8 xb = wa;
#else
        const bf16x8 xb = xa[j][i];
#endif
#ifdef MVT_ABL_NOMFMA
        acc[i][0] += (float)xb[0] + (float)wa[1];
#else
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, xb, acc[i], 0, 0, 0);
#endif
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (g + 1 < KS / GK) {
#pragma unroll
      for (int j = 0; j < GK; ++j)
#pragma unroll
        for (int i = 0; i < NMB; ++i) xa[j][i] = xn[j][i];
    }
  }
}
// The attention output projection (Ko = 288: 18 k-steps) on the same queue: fragments 0..15 in wq, 16 and 17 in wx (requested by the caller just ahead of the tile's x rows) -- the queue is requested long
// before (ahead of / inside the attention phase), so the projection starts on resident operands instead of paying a memory round
// trip per four k-steps -- and, with REFILL, the queue leaves holding the first 16 fragments of the stream nxt (the MLP's fc1).
template <int NMB, bool REFILL>
__device__ __forceinline__ void gemm_wq18(f32x16 (&acc)[NMB], bf16x8 (&wq)[PFQ], const bf16x8 (&wx)[2], const unsigned short* nxt,
                                          const unsigned short* arow, int lda) {
  constexpr int KS = 18, GK = 2;
  bf16x8 xa[GK][NMB], xn[GK][NMB];
#pragma unroll
  for (int j = 0; j < GK; ++j)
#pragma unroll
    for (int i = 0; i < NMB; ++i) xa[j][i] = lds_frag(arow + i * 32 * lda + j * 16);
#pragma unroll
  for (int g = 0; g < KS / GK; ++g) {
    if (g + 1 < KS / GK) {
#pragma unroll
      for (int j = 0; j < GK; ++j)
#pragma unroll
        for (int i = 0; i < NMB; ++i) xn[j][i] = lds_frag(arow + i * 32 * lda + ((g + 1) * GK + j) * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < GK; ++j) {
      const int ks = g * GK + j;
      const bf16x8 wa = ks < PFQ ? wq[ks < PFQ ? ks : 0] : wx[ks >= PFQ ? ks - PFQ : 0];
      if (REFILL && ks < PFQ) wq[ks] = ldg_frag(nxt + ks * FS);
#pragma unroll
      for (int i = 0; i < NMB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, xa[j][i], acc[i], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (g + 1 < KS / GK) {
#pragma unroll
      for (int j = 0; j < GK; ++j)
#pragma unroll
        for (int i = 0; i < NMB; ++i) xa[j][i] = xn[j][i];
    }
  }
}
__device__ __forceinline__ void fill_wq(bf16x8 (&wq)[PFQ], const unsigned short* wrow) {
#pragma unroll
  for (int j = 0; j < PFQ; ++j) wq[j] = ldg_frag(wrow + j * FS);
}

// ---------------------------------------------------------------------------------------------------------------------
// In-kernel attention (head width 48): ONE wave computes softmax(Q K^T / sqrt(48)) V for up to 32 * NMBQ queries and
// 32 * NKB keys of one head and leaves the bf16 result in the workgroup's attention tile As[row][hd*48 + d] -- the B operand
// of the output projection -- instead of a separate attention launch writing it to HBM and this kernel reading it back.
// Same arithmetic, in the same order, as attention_mfma_kernel<1> (attention_mfma.hip): S^T = K . Q^T on the MFMA, online
// softmax over the accumulator registers, O^T += V^T . P^T with P^T taken from the accumulator and V^T from a small
// transposed LDS image, so the fused and the unfused paths agree bit for bit.
//   qrow(i) / krow(j): element row of query i / key j in q / k|v;  vt: this wave's V^T image [48][LDVA];  zrow: >= 40 zero
//   elements (the d padding rows 48..63 of the second 32-row block);  arow0: tile row of query 0.
constexpr int LDVA = 40, DHA = 48, VTA = DHA * LDVA;
constexpr int LDKV = 584;  // bf16 row stride of the in-LDS context k|v tile of ATT 6 (2 * 288 columns + 8)
// operands of one wave_attention unit, as loaded from global memory (NMBQ query blocks, NKB key blocks)
template <int NMBQ, int NKB> struct AttnFrags {
  u32x4 q[NMBQ][3];   // lane (r, h) of block mb: Q[mb*32 + r][ks*16 + 8h .. +7]
  u32x4 k[NKB][3];    // K[kb*32 + r][ks*16 + 8h .. +7]
  u32x2 v[NKB][6];    // V[kb*32 + r][24h + 4i .. +3]
};
template <int NMBQ, int NKB, class QR, class KR>
__device__ __forceinline__ void attn_load(AttnFrags<NMBQ, NKB>& f, const unsigned short* __restrict__ q, int ldq, QR qrow, int nq,
                                          const unsigned short* __restrict__ k, const unsigned short* __restrict__ v, int ldkv, KR krow,
                                          int nk, int hd, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int mb = 0; mb < NMBQ; ++mb) {
    const int qi = mb * 32 + r;
    const long long qo = qrow(qi < nq ? qi : nq - 1) * (long long)ldq + hd * DHA;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) f.q[mb][ks] = *reinterpret_cast<const u32x4*>(q + qo + ks * 16 + 8 * h);
  }
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const int key = kb * 32 + r;
    const long long ko = krow(key < nk ? key : nk - 1) * (long long)ldkv + hd * DHA;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) f.k[kb][ks] = *reinterpret_cast<const u32x4*>(k + ko + ks * 16 + 8 * h);
#pragma unroll
    for (int i = 0; i < 6; ++i) f.v[kb][i] = *reinterpret_cast<const u32x2*>(v + ko + 24 * h + 4 * i);
  }
}
__device__ __forceinline__ f32x4 bf4_to_f32(unsigned lo, unsigned hi) {
  return (f32x4){__uint_as_float(lo << 16), __uint_as_float(lo & 0xFFFF0000u), __uint_as_float(hi << 16), __uint_as_float(hi & 0xFFFF0000u)};
}
// Reductions over the lane pair (l, l ^ 32) without an LDS round trip (gfx950 v_permlane32_swap): with both operands = v the two
// results hold, in every lane, the values of the pair's lower and of its upper lane.
__device__ __forceinline__ float pair_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
  return fmaxf(__int_as_float(r[0]), __int_as_float(r[1]));
}
__device__ __forceinline__ float pair_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
  return __int_as_float(r[0]) + __int_as_float(r[1]);
}

// bd > 0: BLOCK-DIAGONAL attention -- query i only sees the keys j with j / bd == i / bd.  That is the time attention of a tile of
// whole tracks (bd = S frames per track, rows track-major): all tracks of the tile go through ONE unit per head (32 * NMBQ queries x
// 32 * NKB keys, the cross-track scores masked) instead of one 32 x 32 unit per (track, head) -- the S x S attention of a single
// track fills 14 % of a 32 x 32 MFMA tile, and a unit's cost is its dependent chain (operand fetch, three chained MFMAs, softmax,
// V^T staging, P.V), not its arithmetic.
// Softmax in ONE pass over all NKB key blocks (scores of every key block first, then max / exp / sum once): no running maximum,
// no rescaling of the output accumulators between key blocks.
template <int NMBQ, int NKB>
__device__ __forceinline__ void attn_compute(const AttnFrags<NMBQ, NKB>& f, int nq, int nk, int hd, unsigned short* vt,
                                             const unsigned short* zrow, unsigned short* As, int lda, int arow0, int lane, int bd = 0,
                                             int vstride = 0) {
  const int r = lane & 31, h = lane >> 5;
  const float scale = 1.0f / sqrtf((float)DHA);
  // V^T images: vstride > 0: one image per key block (vt + kb * vstride), all staged HERE, once, ahead of everything else (their
  // LDS round trip passes under the score MFMAs); vstride == 0: a single image, restaged per key block inside the query-block loop
  // (only sensible with one query block)
  auto stage_vt = [&](int kb, unsigned short* img) {
    const int key = kb * 32 + r;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      u32x2 w = f.v[kb][i];
      if (key >= nk) w = (u32x2){0u, 0u};
#pragma unroll
      for (int e = 0; e < 4; ++e) img[(24 * h + 4 * i + e) * LDVA + r] = (unsigned short)((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xFFFFu));
    }
  };
  if (vstride > 0) {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) stage_vt(kb, vt + kb * vstride);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
  __builtin_amdgcn_sched_barrier(0);  // (phase boundaries pinned: left free, the scheduler interleaves the V^T stores, the score
  ASTAMP(49);                         //  MFMAs and the softmax into a stream that measured 10 k cycles slower per unit)
  bf16x8 qf[NMBQ][3];
#pragma unroll
  for (int mb = 0; mb < NMBQ; ++mb) {
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      f32x4 a = bf4_to_f32(f.q[mb][ks][0], f.q[mb][ks][1]), b = bf4_to_f32(f.q[mb][ks][2], f.q[mb][ks][3]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[e] *= scale;
        b[e] *= scale;
      }
      const bf16x4 x = __builtin_convertvector(a, bf16x4), y = __builtin_convertvector(b, bf16x4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        qf[mb][ks][e] = x[e];
        qf[mb][ks][4 + e] = y[e];
      }
    }
  }
  // key blocks whose second half (keys 16..31 of the block: accumulator registers 8..15, the second P.V k-step) does not exist --
  // the S = 12 keys of a lone track: their scores are -inf, their probabilities exactly 0 and their products add +0, so leaving
  // them out (wave-uniform) changes no bit of the result and saves half of the exponentials and P.V MFMAs
  const unsigned bdm = bd > 0 ? (65536u / (unsigned)bd + 1u) : 0u;  // j / bd == (j * bdm) >> 16 for j < 64 <= 2^16 / bd
#pragma unroll
  for (int mb = 0; mb < NMBQ; ++mb) {
    const int qi = mb * 32 + r;
    const unsigned qt = ((unsigned)qi * bdm) >> 16;
    f32x16 sc[NKB];
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sc[kb][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.k[kb][ks]), qf[mb][ks], sc[kb], 0, 0, 0);
    }
    // (registers 8..15 of a block = its keys 16..31: left out, wave-uniformly, when they do not exist -- see above; static register
    //  indices everywhere: a run-time trip count inside these unrolled loops would turn them into indexed register accesses)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const bool half = nk <= kb * 32 + 16;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        // (branch-free: as `a && (b || c)` every element became its own divergent branch -- 32 x ~25 instructions with exec
        //  save / restore; bdm == 0 without a block-diagonal mask, so both track numbers are 0 and the compare always holds)
        const unsigned kk = (unsigned)(kb * 32 + (e & 3) + 8 * (e >> 2)) + 4u * (unsigned)h;
        const bool ok = (kk < (unsigned)nk) & (((kk * bdm) >> 16) == qt);
        sc[kb][e] = ok ? sc[kb][e] : -INFINITY;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = fmaxf(mx, sc[kb][e]);
      if (!half) {
#pragma unroll
        for (int e = 8; e < 16; ++e) mx = fmaxf(mx, sc[kb][e]);
      }
    }
    mx = pair_max(mx);
    if (mx == -INFINITY) mx = 0.f;  // (a padding query past nq: every key masked; its column is never stored)
    float ps = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const bool half = nk <= kb * 32 + 16;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sc[kb][e] = __expf(sc[kb][e] - mx);
        ps += sc[kb][e];
      }
      if (!half) {
#pragma unroll
        for (int e = 8; e < 16; ++e) {
          sc[kb][e] = __expf(sc[kb][e] - mx);
          ps += sc[kb][e];
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    ASTAMP(50 + 2 * mb);
    f32x16 oacc[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[db][e] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const bool half = nk <= kb * 32 + 16;
      const unsigned short* vimg = vt + kb * vstride;
      if (vstride == 0 && (mb == 0 || NKB > 1)) {  // single image: restaged per key block (NKB 1: staged once)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        stage_vt(kb, vt);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      }
#pragma unroll
      for (int sk = 0; sk < 2; ++sk) {
        if (sk == 1 && half) break;
        f32x4 lo4, hi4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          lo4[e] = sc[kb][8 * sk + e];
          hi4[e] = sc[kb][8 * sk + 4 + e];
        }
        const bf16x4 bl = __builtin_convertvector(lo4, bf16x4), bh = __builtin_convertvector(hi4, bf16x4);
        bf16x8 pb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pb[e] = bl[e];
          pb[4 + e] = bh[e];
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const unsigned short* vr = (db * 32 + r < DHA) ? &vimg[(db * 32 + r) * LDVA + 16 * sk + 4 * h] : zrow + 16 * sk + 4 * h;
          const u32x2 a0 = *reinterpret_cast<const u32x2*>(vr), a1 = *reinterpret_cast<const u32x2*>(vr + 8);
          const bf16x8 va = __builtin_bit_cast(bf16x8, (u32x4){a0[0], a0[1], a1[0], a1[1]});
          oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb, oacc[db], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    ASTAMP(51 + 2 * mb);
    const float lt = pair_sum(ps);
    const float inv = 1.0f / lt;
    if (qi >= nq) continue;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int d = db * 32 + 8 * gq + 4 * h;
        if (d < DHA) {
          f32x4 t;
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = oacc[db][4 * gq + e] * inv;
          *reinterpret_cast<u32x2*>(&As[(arow0 + qi) * lda + hd * DHA + d]) = __builtin_bit_cast(u32x2, __builtin_convertvector(t, bf16x4));
        }
      }
    }
  }
}

// LayerNorm of the 128 x C tile held as accumulators v[4][16] (wave w: channels w*32 .. +31 of every token), written
// as bf16 into Xs.  st: LDS scratch [8 waves][128 tokens][2].  Optional affine (wave's channel slice).
// ms: (mean, biased variance) of every token block of this lane's token -- computed here unless `have` (the follow-up projections
// of a block normalise the same x with different eps / affine: the statistics exchange and its barrier run once).
template <int NMB>
__device__ __forceinline__ void ln_to_lds(const f32x16 (&v)[NMB], unsigned short* Xs, float* st, int wave, int lane, float eps,
                                          const float* lnw, const float* lnb, float (&ms)[NMB][2], bool have) {
  constexpr int BM = 32 * NMB;
  const int r = lane & 31, h = lane >> 5;
  // the affine parameters of this wave's channel slice are requested first: their memory round trip passes under the statistics
  // exchange and its barrier (read after it, they cost a LayerNorm with affine twice the time of one without)
  f32x4 gw[4], gb[4];
  if (lnw) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      gw[g] = *reinterpret_cast<const f32x4*>(lnw + wave * 32 + 8 * g + 4 * h);
      gb[g] = *reinterpret_cast<const f32x4*>(lnb + wave * 32 + 8 * g + 4 * h);
    }
  }
  if (!have) {  // (workgroup-uniform)
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        s1 += v[mb][e];
        s2 = fmaf(v[mb][e], v[mb][e], s2);
      }
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (h == 0) {
        st[(wave * BM + mb * 32 + r) * 2] = s1;
        st[(wave * BM + mb * 32 + r) * 2 + 1] = s2;
      }
    }
    lds_barrier();
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        s1 += st[(w * BM + mb * 32 + r) * 2];
        s2 += st[(w * BM + mb * 32 + r) * 2 + 1];
      }
      const float mean = s1 / (float)C;
      ms[mb][0] = mean;
      ms[mb][1] = fmaxf(s2 / (float)C - mean * mean, 0.f);
    }
  }
#pragma unroll
  for (int mb = 0; mb < NMB; ++mb) {
    const float mean = ms[mb][0];
    const float rstd = 1.0f / sqrtf(ms[mb][1] + eps);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = wave * 32 + 8 * g + 4 * h;
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (v[mb][4 * g + e] - mean) * rstd;
        if (lnw) o[e] = o[e] * gw[g][e] + gb[g][e];
      }
      const bf16x4 b = __builtin_convertvector(o, bf16x4);
      *reinterpret_cast<u32x2*>(&Xs[(mb * 32 + r) * LDX + n]) = __builtin_bit_cast(u32x2, b);
    }
  }
  lds_barrier();
}

// MODE 0: the whole block in one workgroup per row tile.
// MODE 1 / 2: the same block cut over more workgroups for the 768 virtual-track rows, where 24 row tiles each streaming
// every weight through one CU were pure latency.  Pass 1 (grid.y = MLP chunk): projection + residual + LayerNorm (recomputed
// per chunk, it is cheap), ONE 256-unit chunk of the MLP, partial fc2 output -> ws.  Pass 2 (grid.y = column slice):
// x = x_mid + b2 + sum of the partials in fixed order (deterministic), then the follow-up projections, 8 column blocks per
// workgroup.  Each workgroup streams a quarter of the weights, four times as many CUs pull them.
// ATT: the attention that precedes the block runs INSIDE the kernel (wave_attention) instead of as its own launch:
//   1  time attention (cotracker2/blocks.py:464-467): a tile = p.bmv / S whole tracks (60 rows = 5 tracks at S = 12); the
//      S x S attention of every (track, head) is one wave_attention unit, units dealt round-robin to the 8 waves;
//   2  per-frame attention against p.nkeys <= 64 context tokens of the same frame (point<-virtual cross attention,
//      virtual self attention; blocks.py:477-483): the tile is FRAME-MAJOR -- tokens blockIdx.x*BM .. +BM-1 of frame blockIdx.z,
//      i.e. rows token*S + frame -- so that all its queries share one K / V set; one wave per head.
template <int NMB, int MODE, int ATT>
__global__ __launch_bounds__(NT) void block_fused_bf16(BlockArgs p) {
  using K_ = Cfg<NMB>;
  constexpr int BM = K_::BM, HC = K_::HC, LDH = K_::LDH, LDA = K_::LDA;
  // ATT 6: ATT 2 on the 64-row point tiles with the attention's context (the virtual tokens' k|v) derived in the kernel from a
  // DEFERRED pass 2 (step 0 below)
  constexpr bool CTX = ATT == 6;
  static_assert(!CTX || (NMB == 2 && MODE == 0), "ATT 6 runs on the 64-row tiles of the point rows");
  // (ATT 6: the H buffers also hold the context's k|v tile [64][LDKV] during the prologue -- 74.8 KB instead of 67.6)
  constexpr int HSZ = CTX && (64 * LDKV + 1) / 2 > BM * LDH ? (64 * LDKV + 1) / 2 : BM * LDH;
  __shared__ __attribute__((aligned(16))) unsigned short Xs[BM * LDX];
  __shared__ __attribute__((aligned(16))) unsigned short Hs[2][HSZ];
  __shared__ float st[8 * BM * 2];
  __shared__ __attribute__((aligned(16))) float b1s[4 * C];
  __shared__ __attribute__((aligned(16))) float cbs[CTX ? 2 * 288 : 4];  // bias of the context's k|v projection

  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int r = lane & 31, h = lane >> 5;
  // Logical block indices.  The frame-major grids (x = token tile, y = MLP chunk / column slice, z = frame) are placed XCD-AWARE:
  // the hardware deals consecutive linear workgroup ids round-robin to the 8 XCDs, so with the plain mapping every XCD works on every
  // frame and pulls every frame's hand-over data (split-path partial sums, attention partials, k|v rows -- all written by OTHER
  // XCDs in the previous launch) through the fabric into its own L2: 3.9 MB of partial sums per XCD and launch instead of 0.65 MB.
  // Remapped, XCD c owns the contiguous run [c, c+1) * total/8 of the frame-major order: 1.5 frames at S = 12, the same frames in
  // every kernel of the chain.
  int bx = (int)blockIdx.x, by = (int)blockIdx.y, bz = (int)blockIdx.z;
  if constexpr (ATT == 2 || ATT == 3 || ATT == 5 || ATT == 6) {  // (the frame-major forms, FM below)
    const unsigned total = gridDim.x * gridDim.y * gridDim.z;
    if (total % 8 == 0) {
      const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      const unsigned V = (L % 8) * (total / 8) + L / 8;
      bx = (int)(V % gridDim.x);
      by = (int)((V / gridDim.x) % gridDim.y);
      bz = (int)(V / (gridDim.x * gridDim.y));
    }
  }
  static_assert(ATT == 0 || ATT == 5 || MODE != 2, "pass 2 of the split path has no attention");
  static_assert(ATT != 5 || MODE == 2, "ATT 5 = pass 2 on the frame-major tiles of an ATT 2 / ATT 3 pass 1");
  static_assert((MODE != 1 && MODE != 2) || NMB == 1, "the split path runs on 32-row tiles");
  constexpr bool FM = ATT == 2 || ATT == 3 || ATT == 5 || ATT == 6;  // frame-major tile: tokens bx*BM.. of frame bz
  const int bmv = ATT == 1 ? p.bmv : BM;                      // rows of the tile that hold tokens
  const long long m0 = FM ? 0 : (long long)bx * bmv;
  const long long ntok = FM ? p.M / p.S : 0;                   // tokens per frame
  // global row of tile row i (-1: none)
  auto grow = [&](int i) -> long long {
    if (FM) {
      const long long tk = (long long)bx * BM + i;
      return tk < ntok ? tk * p.S + (long long)bz : -1;
    }
    const long long m = m0 + i;
    return (i < bmv && m < p.M) ? m : -1;
  };
  // rows [rlo, rhi) bound the tile (workgroup-uniform)
  const long long rlo = FM ? (long long)bx * BM * p.S : m0;
  const long long rhi = FM ? ((long long)bx * BM + BM) * p.S : m0 + bmv;
  constexpr bool HAS_MLP = MODE == 0 || MODE == 1;  // MODE 2: pass 2 of the split path; MODE 3: LayerNorm + projections of a
                                                    // read-only x in whole 64-row tiles (mvt_ln_proj_bf16 at large M)
  if (HAS_MLP)
    for (int i = t; i < p.H; i += NT) b1s[i] = p.b1[i];
  // Split-path workspace: [chunk 0 = x after the projection, 1.. = MLP partials][tile][wave][g][lane][4 floats] -- the accumulator
  // layout itself, so that pass 1's stores and pass 2's loads are contiguous 1-KiB pieces per wave instruction (as [M][C] rows
  // every access of a wave touched 32 different rows and pass 2 spent half its time on these loads).  Both passes use the same
  // tiling (grid.x, grid.z), hence the same (tile, wave, g, lane) <-> element map.
  const long long ws_tile = FM ? (long long)bz * gridDim.x + bx : (long long)bx;
  const long long ws_ntile = FM ? (long long)gridDim.z * gridDim.x : (long long)gridDim.x;
  auto ws_off = [&](int chunk, int g) -> long long { return ((((long long)chunk * ws_ntile + ws_tile) * 8 + wave) * 4 + g) * 256 + lane * 4; };
  // a projection runs in this workgroup when its row range meets the workgroup's rows (workgroup-uniform)
  auto active = [&](int q) { return q < p.n_next && rlo < p.next[q].row_hi && rhi > p.next[q].row_lo; };
  bf16x8 wq[PFQ];  // the weight-fragment queue, chained through the MLP and the follow-up projections
  // 32-row tiles (the 768 virtual-token rows, pure latency chains) have registers to spare: the first 16 weight fragments the
  // wave will consume are requested HERE, so their memory round trip (the ~40 MB of updater weights do not live in L2) overlaps
  // the token / attention-operand loads and the LayerNorm instead of following them.
  constexpr bool EARLY = NMB == 1 && MODE != 0;
  bool early_have = false;
  // The attention output projection (every form but pass 2 and the input transform, when there is an attention) takes its 18
  // weight fragments from the queue too (gemm_wq18): they are requested ahead of / inside the attention phase and the queue is
  // refilled with the first fc1 fragments while the projection runs, so neither GEMM starts on a memory round trip.
  const bool outp = MODE != 2 && ATT != 4 && (ATT != 0 || p.att != nullptr);
  const unsigned short* wo_row = p.wo + ((long long)wave * 18 * 64 + lane) * 8;
  const unsigned short* fc1_first = p.w1 + ((long long)(wave % K_::JW) * (C / 16) * 64 + lane) * 8 +
                                    (long long)(MODE == 1 ? by : 0) * K_::JW * (C / 16) * FS;
  auto prefetch_wo = [&]() { fill_wq(wq, wo_row); };
  if (HAS_MLP && !outp) {
    fill_wq(wq, fc1_first);
  } else if (CTX) {
    // the queue starts with the context's k|v projection (18 column blocks, every wave has a first one), requested ahead of
    // the partial sums: the first barrier of step 0 (a __syncthreads waits for EVERY outstanding load) then finds it resident --
    // requested after the sums, the fragments' whole memory round trip sat in front of that barrier (in-kernel stamps: 6-9 k cycles)
    fill_wq(wq, p.c_kv.w + ((long long)wave * (C / 16) * 64 + lane) * 8);
  } else if (outp) {
    prefetch_wo();
  } else if (EARLY && MODE == 2) {
    const int nb0e = by * 8 + wave;
    if (active(0) && nb0e < (p.next[0].N + 31) / 32) {
      fill_wq(wq, p.next[0].w + ((long long)nb0e * (C / 16) * 64 + lane) * 8);
      early_have = true;
    }
  }

  STAMP(0);
  // token values of this wave's 32-channel slice: v[mb][e] = x[m0 + mb*32 + r][wave*32 + (e&3) + 8*(e>>2) + 4h]
  f32x16 v[NMB];
#pragma unroll
  for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
    for (int e = 0; e < 16; ++e) v[mb][e] = 0.f;

  // the tile's own x rows (added after the projection) are requested BEFORE the projection GEMM, which hides their round trip
  f32x4 xr[NMB][4];
  auto load_x = [&]() {
    if (ATT == 4 || (MODE == 2 && p.ws)) return;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      const long long m = grow(mb * 32 + r);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        xr[mb][g] = m >= 0 ? *reinterpret_cast<const f32x4*>(p.x + m * (long long)p.ldx + wave * 32 + 8 * g + 4 * h) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };

  // ahead of the barrier that ends the attention / staging phase: the last two projection fragments, then the x rows (in this order:
  // loads return in order, and the fragments are needed first) -- the waves that arrive last at the barrier are the critical path,
  // and for them the x rows' round trip now passes under the projection GEMM instead of following it
  bf16x8 wx[2];
  auto pre_gemm = [&]() {
    wx[0] = ldg_frag(wo_row + 16 * FS);
    wx[1] = ldg_frag(wo_row + 17 * FS);
    load_x();
  };

  // ---- 0. (ATT 6) the DEFERRED PASS 2 of the context block.  The 64 context tokens of this frame (the virtual tracks) went through
  // pass 1 of the split path in the previous launch; instead of a pass-2 launch finishing them (x = x_mid + b2 + partial sums, then
  // LayerNorm + k|v projection into a tensor that this kernel reads back) every workgroup of the frame finishes them itself: same
  // loads, the same fixed summation order, LayerNorm and GEMM arithmetic as pass 2 (bit-identical), the k|v tile stays in LDS.
  // One launch and one dependent read of freshly written data less per layer; tile 0 of the frame stores the context's x, the first
  // workgroups of the frame evaluate the optional further projection of the context rows (the next layer's time q|k|v), 8 column
  // blocks each.
  if constexpr (CTX) {
    constexpr int KVB = 2 * 288 / 32;  // column blocks of the projection: k | v
    unsigned short* KV = &Hs[0][0];
    unsigned short* XN = &Xs[0];  // the 64 normalised context rows (the GEMM's activation operand)
    for (int i = t; i < KVB * 32; i += NT) cbs[i] = p.c_kv.b[i];
    const long long ctile = 2LL * gridDim.z;
    auto cws_off = [&](int chunk, int mb, int g) -> long long {
      return ((((long long)chunk * ctile + (long long)bz * 2 + mb) * 8 + wave) * 4 + g) * 256 + lane * 4;
    };
    constexpr int MAXCH = 4;
    const int nch = p.c_nch;
    f32x16 cv[2];
    // (two halves of the channel quads: 80 registers of loads in flight each, next to the 64 of the weight queue -- the transfer is
    //  bound by the CU's 64 B / clk from L2, 320 KB per workgroup, not by the number of loads in flight)
#pragma unroll
    for (int gh = 0; gh < 2; ++gh) {
      f32x4 part[2][2][MAXCH + 1], b2v[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int sidx = 0; sidx <= MAXCH; ++sidx)
            part[mb][g][sidx] = sidx <= nch ? *reinterpret_cast<const f32x4*>(p.c_ws + cws_off(sidx, mb, 2 * gh + g)) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 2; ++g) b2v[g] = *reinterpret_cast<const f32x4*>(p.c_b2 + wave * 32 + 8 * (2 * gh + g) + 4 * h);
      __builtin_amdgcn_sched_barrier(0);  // (all loads of the half ahead of its first add)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          f32x4 xv = part[mb][g][0];
#pragma unroll
          for (int sidx = 1; sidx <= MAXCH; ++sidx)
            if (sidx <= nch) {
#pragma unroll
              for (int e = 0; e < 4; ++e) xv[e] += part[mb][g][sidx][e];
            }
#pragma unroll
          for (int e = 0; e < 4; ++e) cv[mb][4 * (2 * gh + g) + e] = 0.f + (xv[e] + b2v[g][e]);  // (pass 2 accumulates into a zero: same bits)
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    STAMP(56);
    auto krow_w = [&](int nb) { return p.c_kv.w + ((long long)nb * (C / 16) * 64 + lane) * 8; };
    auto nrow_w = [&](int nb) { return p.c_next.w + ((long long)nb * (C / 16) * 64 + lane) * 8; };
    if (bx == 0) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const long long m = (long long)(mb * 32 + r) * p.S + (long long)bz;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = cv[mb][4 * g + e];
          *reinterpret_cast<f32x4*>(p.c_x + m * (long long)p.c_ldx + wave * 32 + 8 * g + 4 * h) = o;
        }
      }
    }
    float cms[2][2];
    ln_to_lds<2>(cv, XN, st, wave, lane, p.c_kv.eps, p.c_kv.lnw, p.c_kv.lnb, cms, false);
    STAMP(57);
    const int nnb = p.c_next.w ? (p.c_next.N + 31) / 32 : 0;
    const bool do_next = bx * 8 < nnb;  // (workgroup-uniform)
    const int nbn = bx * 8 + wave;
    const bool my_next = nbn < nnb;
#pragma unroll 1
    for (int nb = wave; nb < KVB; nb += 8) {
      f32x16 acc[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mb][e] = 0.f;
      gemm_wq<C / 16, 2>(acc, wq, krow_w(nb), nb + 8 < KVB ? krow_w(nb + 8) : (my_next ? nrow_w(nbn) : wo_row), &XN[r * LDX + 8 * h], LDX, 0);
      const int kc = nb * 32;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = acc[mb][4 * g + e] + cbs[nb * 32 + 8 * g + 4 * h + e];
          *reinterpret_cast<u32x2*>(&KV[(mb * 32 + r) * LDKV + kc + 8 * g + 4 * h]) = __builtin_bit_cast(u32x2, __builtin_convertvector(o, bf16x4));
        }
    }
    STAMP(58);
    if (do_next) {
      lds_barrier();  // every wave is done reading Xs; st (the statistics scratch) is idle: it carries the projection's bias
      for (int i = t; i < p.c_next.N; i += NT) st[i] = p.c_next.b[i];
      ln_to_lds<2>(cv, XN, st, wave, lane, p.c_next.eps, p.c_next.lnw, p.c_next.lnb, cms, true);
      if (my_next) {
        f32x16 acc[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[mb][e] = 0.f;
        gemm_wq<C / 16, 2>(acc, wq, nrow_w(nbn), wo_row, &XN[r * LDX + 8 * h], LDX, 0);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const long long m = (long long)(mb * 32 + r) * p.S + (long long)bz;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = nbn * 32 + 8 * g + 4 * h;
            if (n + 3 < p.c_next.N) {
              f32x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = acc[mb][4 * g + e] + st[n + e];
              store_act4(p.c_next.y, m * (long long)p.c_next.ldy + n, o, p.c_next.y_bf16);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (n + e < p.c_next.N) store_act(p.c_next.y, m * (long long)p.c_next.ldy + n + e, acc[mb][4 * g + e] + st[n + e], p.c_next.y_bf16);
            }
          }
        }
      }
      lds_barrier();  // st is about to become the attention's zero row
    }
    STAMP(59);
    // (the barrier that publishes the k|v tile and frees Xs is the one at the head of the attention phase below)
  }

  // ---- 1. attention output projection (accumulated into v, x is added afterwards)
  if (MODE == 3 && ATT == 4) {
    // Input transform (cotracker2/blocks.py:456-459): the residual stream x is BORN here.  The 581-wide token rows (padded to
    // 37 k-steps) are staged as bf16 in two halves (19 + 18 k-steps: one half fits the H buffers), x = tokens . Win^T + bin for
    // the point rows; the rows of the virtual tracks start from their learned tokens.  x is then stored and projected like in
    // the plain MODE 3 -- one launch instead of GEMM + broadcast + LayerNorm/projection.
    unsigned short* As = &Hs[0][0];
    constexpr int LDT = 312, KSA = 19, KSB = 18;
    static_assert(BM * LDT <= 2 * BM * LDH, "a token half fits the H buffers");
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      const int k0 = half ? KSA * 16 : 0, kq = (half ? KSB : KSA) * 4;  // float4 per row of this half
      if (half) lds_barrier();  // every wave is done with the first half
      if (p.tokx) {
        for (int f = t; f < BM * kq; f += NT) {
          const int row = f / kq, c = (f - row * kq) * 4;
          const long long m = grow(row);
          u32x2 w = (u32x2){0u, 0u};
          if (m >= 0 && m < p.Mp && k0 + c < p.ldtok)
            w = __builtin_bit_cast(u32x2, __builtin_convertvector(*reinterpret_cast<const f32x4*>(p.tokx + m * (long long)p.ldtok + k0 + c), bf16x4));
          *reinterpret_cast<u32x2*>(&As[row * LDT + c]) = w;
        }
      } else {
        // The token rows are ASSEMBLED here (no token matrix in HBM at all).  One element of the updater's input token
        // (mvtracker.py:379-387): [sin|cos flow embedding 3E | flow 3 | fcorr Fc | ffeats Cf | track mask, visibility 2] + positional
        // + time embedding -- the same expressions, in the same order, as token_assemble_kernel: bit-identical
        // (test_in_kernel_token_assembly_bit_identical).  A wave takes whole rows (8 each), its lanes sweep the row's element PAIRS:
        // everything per row (track, frame, flow) is wave-uniform, a wave's loads are contiguous along the row, the six loads of
        // a pair are issued before its arithmetic, and a sin | cos pair shares one sincosf.  (As quads of four branchy per-element
        // calls -- a dozen dependent, 16-byte-strided 4-byte loads per quad -- this staging was 110 k of the kernel's 148 k cycles:
        // bound by the NUMBER of scattered load instructions, 2.7 k per workgroup.)
        const int E = p.t_E, Fc = p.t_Fc, Cf = p.t_Cf, D = p.t_D;
        const int o1 = 3 * E + 3, o2 = o1 + Fc, o3 = o2 + Cf;  // first element of fcorr / ffeats / (mask, visibility)
        const int wp = kq * 2;                                  // element pairs per row of this half
        const float step = 1000.0f / (float)E;
#pragma unroll 1
        for (int rr = wave; rr < BM; rr += 8) {
          const long long mm = grow(rr);
          const bool okr = mm >= 0 && mm < p.Mp;  // (wave-uniform)
          const long long m = okr ? mm : 0;
          const int n = (int)(m / p.S), sidx = (int)(m - (long long)n * p.S);
          const float* cc = p.t_coords + m * 3;
          const float* c0 = p.t_coords + (long long)n * p.S * 3;
          const float fl[3] = {cc[0] - c0[0], cc[1] - c0[1], cc[2] - c0[2]};
          const float* pp = p.t_pos + (long long)n * D;
          const float* tp = p.t_time + (long long)sidx * D;
          const float* fcr = p.t_fcorr + m * Fc - o1;
          const float* ffr = p.t_ffeats + m * Cf - o2;
          const float* mvr = p.t_maskvis + m * 2 - o3;
          constexpr int SW = 3;  // sweeps of 64 pairs cover a half row (152 / 144 pairs)
          float pe[SW][2], te[SW][2], sv[SW][2];
#pragma unroll
          for (int u = 0; u < SW; ++u) {
            const int d0 = k0 + 2 * (lane + 64 * u);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int de = d0 + e < D ? d0 + e : D - 1;
              pe[u][e] = pp[de];
              te[u][e] = tp[de];
              const int ds = de < o1 ? o1 : de;  // (flow / embedding elements take their value from fl: any valid address will do)
              const float* src = ds < o2 ? fcr + ds : (ds < o3 ? ffr + ds : mvr + ds);
              sv[u][e] = *src;
            }
          }
          __builtin_amdgcn_sched_barrier(0);  // (every load of the row ahead of the first sincos)
#pragma unroll
          for (int u = 0; u < SW; ++u) {
            const int j = lane + 64 * u, d0 = k0 + 2 * j;
            if (j >= wp) continue;
            unsigned w = 0u;
            if (okr && d0 < D) {
              if (d0 + 1 < 3 * E) {  // sine and cosine of one argument (d0 and E are even)
                const int a = (d0 >= E) + (d0 >= 2 * E), w0 = d0 - a * E;
                sincosf(fl[a] * ((float)w0 * step), &sv[u][0], &sv[u][1]);
              } else {
#pragma unroll
                for (int e = 0; e < 2; ++e)
                  if (d0 + e < o1) sv[u][e] = fl[d0 + e - 3 * E];  // (the three flow components)
              }
              const float v0 = (sv[u][0] + pe[u][0]) + te[u][0];
              const float v1 = d0 + 1 < D ? (sv[u][1] + pe[u][1]) + te[u][1] : 0.f;
              w = (unsigned)mvt_bf16_bits(v0) | ((unsigned)mvt_bf16_bits(v1) << 16);
            }
            *reinterpret_cast<unsigned*>(&As[rr * LDT + 2 * j]) = w;
          }
        }
      }
      STAMP(1 + 4 * half);
      lds_barrier();
      STAMP(2 + 4 * half);
      if (half == 0) gemm_wt<KSA, NMB>(v, p.win + ((long long)wave * (KSA + KSB) * 64 + lane) * 8, &As[r * LDT + 8 * h], LDT, 0);
      else gemm_wt<KSB, NMB>(v, p.win + (((long long)wave * (KSA + KSB) + KSA) * 64 + lane) * 8, &As[r * LDT + 8 * h], LDT, 0);
      STAMP(3 + 4 * half);
    }
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      const long long m = grow(mb * 32 + r);
      const bool virt = m >= p.Mp;
      const long long tk = virt ? (m - p.Mp) / p.S : 0;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int c = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        v[mb][e] = virt ? p.virt[tk * C + c] : v[mb][e] + p.bin[c];
      }
    }
  } else if (MODE != 2 && ATT == 3) {
    // The attention tile from the key-split partials of the virtual<-point attention (one (m, l, O^T) state per split and
    // (frame, head) chunk, in attention_mfma_kernel's accumulator layout).  The tile is frame-major -- the 32 virtual tokens
    // mb = bx of frame bz -- so wave hd reads the records of its (frame, head) chunk in their NATIVE lane layout
    // (lane (r, h): query mb*32 + r, rows d = db*32 + (e&3) + 8(e>>2) + 4h) and combines the splits with the same sequential
    // arithmetic, in the same order, as attention_merge_kernel -- which this replaces: bit-identical results.
    unsigned short* As = &Hs[0][0];
    static_assert(ATT != 3 || NMB == 1, "partials tiles are 32 tokens");
    if (wave < 6) {
      const int mb = bx;
      const long long nchunk = (long long)p.S * 6;
      const long long cid = (long long)bz * 6 + wave;
      // the records of ALL splits are requested before the first is combined (they were written by other CUs: as a load-combine
      // loop over the splits every iteration paid its own trip to the memory side); MVT_ATTN_NSPLIT splits at most
      constexpr int MAXS = MVT_ATTN_NSPLIT;
      float ms[MAXS], ls[MAXS];
      f32x4 os[MAXS][2][4];
#pragma unroll
      for (int w = 0; w < MAXS; ++w) {
        const long long rec = (w < p.nsplit ? w : 0) * nchunk + cid;
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(p.parts + mvt_part_off(rec, 0, lane));
        ms[w] = q0[mb];
        ls[w] = q0[2 + mb];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            os[w][db][g] = *reinterpret_cast<const f32x4*>(p.parts + mvt_part_off(rec, 1 + (mb * 2 + db) * 4 + g, lane));
      }
      __builtin_amdgcn_sched_barrier(0);  // (keeps the scheduler from sinking the loads back to their uses)
      float mm = ms[0], ll = ls[0];
      f32x16 oa[2];
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) oa[db][4 * g + e] = os[0][db][g][e];
#pragma unroll
      for (int w = 1; w < MAXS; ++w) {
        if (w >= p.nsplit) break;
        const float mw = ms[w];
        const float mn = fmaxf(mm, mw);
        const float ca = (mm == -INFINITY) ? 0.f : __expf(mm - mn);
        const float cb = (mw == -INFINITY) ? 0.f : __expf(mw - mn);
        ll = fmaf(ll, ca, ls[w] * cb);
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) oa[db][4 * g + e] = fmaf(oa[db][4 * g + e], ca, os[w][db][g][e] * cb);
        mm = mn;
      }
      const float inv = 1.0f / (ll + __shfl_xor(ll, 32, 64));
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = db * 32 + 8 * g + 4 * h;
          if (d < DHA) {
            f32x4 t4;
#pragma unroll
            for (int e = 0; e < 4; ++e) t4[e] = oa[db][4 * g + e] * inv;
            *reinterpret_cast<u32x2*>(&As[r * LDA + wave * DHA + d]) = __builtin_bit_cast(u32x2, __builtin_convertvector(t4, bf16x4));
          }
        }
    }
    pre_gemm();
    lds_barrier();
    gemm_wq18<NMB, HAS_MLP>(v, wq, wx, fc1_first, &As[r * LDA + 8 * h], LDA);
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int e = 0; e < 16; ++e) v[mb][e] += p.bo[wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h];
  } else if (MODE != 2 && ATT != 0) {
    unsigned short* As = &Hs[0][0];
    // V^T images of the waves: Xs is idle until the first LayerNorm (NMB 2: all eight fit; NMB 1: four there, two behind As);
    // the zero row (d padding) lives in the LayerNorm scratch, equally idle
    unsigned short* zrow = reinterpret_cast<unsigned short*>(st);
    if (t < 48) zrow[t] = 0;
    constexpr int XW = (BM * LDX) / VTA;  // images that fit Xs
    // NMB 2 (64-row tiles): TWO images per attention wave (one per key block, staged once): waves 0..3 in Xs, waves 4, 5 behind As;
    // NMB 1: one image per wave (four in Xs, two behind As), restaged per key block
    constexpr int VPW = NMB == 2 ? 2 : 1;
    constexpr int XWV = XW / VPW;  // waves whose images live in Xs
    unsigned short* vt = wave < XWV ? &Xs[wave * VPW * VTA] : &Hs[0][0] + ((BM * LDA + 7) & ~7) + (wave - XWV) * VPW * VTA;
    constexpr int vstride = VPW == 2 ? VTA : 0;
    static_assert(XWV + ((2 * BM * LDH - BM * LDA - 8) / VTA) / VPW >= 6, "the V^T images of six waves fit beside the attention tile");
    lds_barrier();
    STAMP(1);
    if (ATT == 1) {
      // Time attention of the tile's whole tracks (rows track-major: row = track * S + frame): one BLOCK-DIAGONAL unit per head --
      // 64 queries x 64 keys, query i sees the keys of its own track only -- on waves 0..5.  (Round 2 ran one 32 x 32 unit per
      // (track, head): 30 units dealt to the 8 waves, four dependent chains per wave, 384 operand loads per workgroup; in-kernel
      // stamps: 24-30 k of the kernel's 102 k cycles.)
      const long long left = p.M - m0;
      const int nrow = left < bmv ? (int)left : bmv;
      if (wave < 6 && nrow > 0) {
        auto row = [&](int j) { return m0 + j; };
        AttnFrags<NMB, 2> fr;
        attn_load<NMB, 2>(fr, p.aq, p.ldaq, row, nrow, p.ak, p.av, p.ldakv, row, nrow, wave, lane);
        attn_compute<NMB, 2>(fr, nrow, nrow, wave, vt, zrow, As, LDA, 0, lane, p.S, vstride);
      }
    } else {
      const bool on = wave < 6 && grow(0) >= 0;
      const long long left = ntok - (long long)bx * BM;
      const int nq = left < BM ? (int)left : BM;
      auto qrow = [&](int i) { return ((long long)bx * BM + i) * p.S + (long long)bz; };
      AttnFrags<NMB, 2> fr;
      if (CTX) {
        // k|v of the frame's context tokens from the LDS tile of step 0 (tile row = context token); the tile lives where the
        // attention output and the V^T images go: every wave holds its operands in registers before the first of them is written
        const unsigned short* KV = &Hs[0][0];
        auto krow = [&](int j) { return (long long)j; };
        if (on) attn_load<NMB, 2>(fr, p.aq, p.ldaq, qrow, nq, KV, KV + 288, LDKV, krow, p.nkeys, wave, lane);
        lds_barrier();
      } else {
        auto krow = [&](int j) { return (long long)j * p.S + (long long)bz; };
        if (on) attn_load<NMB, 2>(fr, p.aq, p.ldaq, qrow, nq, p.ak, p.av, p.ldakv, krow, p.nkeys, wave, lane);
      }
      if (on) attn_compute<NMB, 2>(fr, nq, p.nkeys, wave, vt, zrow, As, LDA, 0, lane, 0, vstride);
    }
    pre_gemm();
    STAMP(2);
    lds_barrier();
    STAMP(3);
    gemm_wq18<NMB, HAS_MLP>(v, wq, wx, fc1_first, &As[r * LDA + 8 * h], LDA);
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int e = 0; e < 16; ++e) v[mb][e] += p.bo[wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h];
  } else if (MODE != 2 && p.att) {
    // att tile (fp32) -> bf16 [128][296], overlaid on the two (still unused) H buffers
    unsigned short* As = &Hs[0][0];
    const int q4 = p.Ko / 4;  // float4 per row
    for (int f = t; f < BM * q4; f += NT) {
      const int row = f / q4, c = (f - row * q4) * 4;
      const long long m = grow(row);
      u32x2 w = (u32x2){0u, 0u};
      if (m >= 0) {
        if (p.att_bf16) {
          w = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(p.att) + m * (long long)p.ldatt + c);
        } else {
          const f32x4 a = *reinterpret_cast<const f32x4*>(p.att + m * (long long)p.ldatt + c);
          w = __builtin_bit_cast(u32x2, __builtin_convertvector(a, bf16x4));
        }
      }
      *reinterpret_cast<u32x2*>(&As[row * LDA + c]) = w;
    }
    pre_gemm();
    STAMP(2);
    lds_barrier();
    STAMP(3);
    gemm_wq18<NMB, HAS_MLP>(v, wq, wx, fc1_first, &As[r * LDA + 8 * h], LDA);
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int e = 0; e < 16; ++e) v[mb][e] += p.bo[wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h];
  } else {
    load_x();
  }
  STAMP(4);
  if (MODE == 2 && p.ws) {
    // pass 2 of the split path: x after the projection (pass 1) + b2 + the MLP partials, in chunk order.  ALL the workspace
    // loads (up to 5 per element quad, written by other CUs: every one a trip to the memory side) are issued before the first
    // add -- as a load-add loop over the chunks they ran one round trip after the other, 13 k cycles of this 24 k-cycle kernel.
    constexpr int MAXCH = 4;  // H <= 4 C, 256 hidden units per chunk
    const int nch = p.H / Cfg<NMB>::HC;
    f32x4 part[NMB][4][MAXCH + 1];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      const long long m = grow(mb * 32 + r);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int sidx = 0; sidx <= MAXCH; ++sidx)
          part[mb][g][sidx] = (m >= 0 && sidx <= nch) ? *reinterpret_cast<const f32x4*>(p.ws + ws_off(sidx, g)) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_sched_barrier(0);  // (keeps the scheduler from sinking the loads back to their uses)
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 xv = part[mb][g][0];
#pragma unroll
        for (int sidx = 1; sidx <= MAXCH; ++sidx)
          if (sidx <= nch) {
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[e] += part[mb][g][sidx][e];
          }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[mb][4 * g + e] += xv[e] + p.b2[wave * 32 + 8 * g + 4 * h + e];
      }
  } else if (ATT != 4) {  // (MODE 2 without a workspace = mvt_ln_proj_bf16: x is final and only read; ws / b2 are null there)
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[mb][4 * g + e] += xr[mb][g][e];
  }

  // ---- 2. MLP: LayerNorm -> Xs, then chunks of HC hidden units.  x stays in registers: parking it in global memory
  //         (a store plus two dependent re-reads per token) cost more HBM traffic than the whole GEMM loop took.
  auto store_x = [&]() {
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      const long long m = grow(mb * 32 + r);
      if (m >= 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = v[mb][4 * g + e];
          *reinterpret_cast<f32x4*>(p.x + m * (long long)p.ldx + wave * 32 + 8 * g + 4 * h) = o;
        }
      }
    }
  };
  STAMP(5);
  float ln_ms[NMB][2];
  if (HAS_MLP) ln_to_lds<NMB>(v, Xs, st, wave, lane, 1e-6f, nullptr, nullptr, ln_ms, false);  // ends with a barrier: Hs is free again
  STAMP(6);
  const bool tail_next = MODE == 0 && active(0) && wave < (p.next[0].N + 31) / 32;
  if (HAS_MLP) {
    f32x16 acc2[NMB];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc2[mb][e] = 0.f;
    constexpr int JW = K_::JW, NM1 = K_::NM1;
    const int jb = wave % JW, mp = wave / JW;
    const int nchunk = p.H / HC;
    const unsigned short* w1row = p.w1 + ((long long)jb * (C / 16) * 64 + lane) * 8;                 // + c * 4 blocks
    const unsigned short* w2row = p.w2 + ((long long)wave * (p.H / 16) * 64 + lane) * 8;             // + c * (HC/16) k-steps
    static_assert(HC / 16 == PFQ && C / 16 == PFQ, "every GEMM segment is exactly one queue length");
    const unsigned short* tail = tail_next ? p.next[0].w + ((long long)wave * (C / 16) * 64 + lane) * 8 : w1row;
    // ONE fragment queue for the whole MLP: the wave consumes fc1(c0), fc2(c0), fc1(c1), ... strictly in this order, 16
    // k-steps each, so the queue always holds the next 16 fragments of that sequence (>= 16 MFMA k-steps of lookahead,
    // which is what an L2 round trip needs; two half-depth queues left every fragment ~250 cycles short)
    const int c_lo = MODE == 1 ? by : 0, c_hi = MODE == 1 ? c_lo + 1 : nchunk;
    // (the queue already holds fc1(c_lo): filled at kernel start, or by the output projection while it ran)
#pragma unroll 1
    for (int c = c_lo; c < c_hi; ++c) {
      unsigned short* Hb = Hs[c & 1];
      // fc1: H^T block (hidden jb of this chunk) x token blocks NM1*mp ..
      f32x16 ha[NM1];
#pragma unroll
      for (int i = 0; i < NM1; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) ha[i][e] = 0.f;
      STAMP(8 + 5 * c);
      gemm_wq<C / 16, NM1>(ha, wq, w1row + (long long)c * JW * (C / 16) * FS, w2row + (long long)c * (HC / 16) * FS, &Xs[r * LDX + 8 * h], LDX,
                           NM1 * mp);
      STAMP(9 + 5 * c);
#pragma unroll
      for (int i = 0; i < NM1; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 bq = *reinterpret_cast<const f32x4*>(&b1s[c * HC + jb * 32 + 8 * g + 4 * h]);
          f32x4 o;
#ifdef MVT_ABL_NOGELU
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = ha[i][4 * g + e] + bq[e];
#else
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const f32x2 y = gelu_tanh_pk((f32x2){ha[i][4 * g + e] + bq[e], ha[i][4 * g + e + 1] + bq[e + 1]});
            o[e] = y[0];
            o[e + 1] = y[1];
          }
#endif
          const bf16x4 b = __builtin_convertvector(o, bf16x4);
          *reinterpret_cast<u32x2*>(&Hb[((NM1 * mp + i) * 32 + r) * LDH + jb * 32 + 8 * g + 4 * h]) = __builtin_bit_cast(u32x2, b);
        }
      }
      STAMP(10 + 5 * c);
      lds_barrier();
      STAMP(11 + 5 * c);
      // fc2 partial: out^T block (channels of this wave) += W2[:, chunk] . H^T; refills the queue with the next chunk's fc1
      // fragments (after the last chunk: with the first follow-up projection's, or harmlessly with fc1(c0) again)
      const unsigned short* after = c + 1 < c_hi ? w1row + (long long)(c + 1) * JW * (C / 16) * FS : tail;
      gemm_wq<HC / 16, NMB>(acc2, wq, w2row + (long long)c * (HC / 16) * FS, after, &Hb[r * LDH + 8 * h], LDH, 0);
      STAMP(12 + 5 * c);
    }
    if (MODE == 1) {  // partial fc2 output of this chunk (+ x after the projection, once) -> workspace; pass 2 finishes
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) {
        const long long m = grow(mb * 32 + r);
        if (m < 0) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 a4, x4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a4[e] = acc2[mb][4 * g + e];
            x4[e] = v[mb][4 * g + e];
          }
          *reinterpret_cast<f32x4*>(p.ws + ws_off(by + 1, g)) = a4;
          if (by == 0) *reinterpret_cast<f32x4*>(p.ws + ws_off(0, g)) = x4;
        }
      }
      return;
    }
    // x += MLP output
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[mb][4 * g + e] += acc2[mb][4 * g + e] + p.b2[wave * 32 + 8 * g + 4 * h + e];
  }
  STAMP(30);
  if (MODE == 0 || ATT == 4 || (p.ws && by == 0)) store_x();  // (never in the projection-only form: x is read-only there)
  STAMP(31);

  // ---- 3. optional follow-up projections: y_i = LayerNorm_i(x) . Wn_i^T + bn_i
  bool have = tail_next || early_have;  // the queue already holds this wave's first block of the projection
  bool have_ms = false;
#pragma unroll
  for (int q = 0; q < MVT_BLOCK_MAX_NEXT; ++q) {  // static indices: a dynamically indexed kernel-argument array would live in scratch
    if (q >= p.n_next) break;
    if (!active(q)) continue;  // (have is false here: a chain is only set up towards an active projection)
    const mvt_block_next nx = p.next[q];
    lds_barrier();  // every wave is done reading Xs / Hs
    // the projection's bias goes through LDS (b1s is free after the MLP; N <= 4C): a global load in the epilogue would sit
    // behind the 16 queued fragment loads in the in-order vmcnt counter and expose their whole latency every block
    for (int i = t; i < nx.N; i += NT) b1s[i] = nx.b[i];
    STAMP(32 + 8 * q);
    ln_to_lds<NMB>(v, Xs, st, wave, lane, nx.eps, nx.lnw, nx.lnb, ln_ms, have_ms);
    have_ms = true;  // (x does not change any more: the next projection reuses the statistics)
    STAMP(33 + 8 * q);
    const int nblocks = (nx.N + 31) / 32;
    auto nrow_of = [&](int nb) { return nx.w + ((long long)nb * (C / 16) * 64 + lane) * 8; };
    const int nb0 = MODE == 2 ? by * 8 + wave : wave, nbstep = MODE == 2 ? 8 * (int)gridDim.y : 8;
    if (nb0 < nblocks && !have) fill_wq(wq, nrow_of(nb0));
    // after this wave's last block the queue moves on to its first block of the next projection, if that one runs here
    const int qn = q + 1 < MVT_BLOCK_MAX_NEXT ? q + 1 : MVT_BLOCK_MAX_NEXT - 1;
    const unsigned short* chain = nullptr;
    if ((MODE == 0 || MODE == 3) && q + 1 < MVT_BLOCK_MAX_NEXT && active(q + 1) && wave < (p.next[qn].N + 31) / 32)
      chain = p.next[qn].w + ((long long)wave * (C / 16) * 64 + lane) * 8;
    have = chain != nullptr && wave < nblocks;
    for (int nb = nb0; nb < nblocks; nb += nbstep) {
      f32x16 acc[NMB];
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mb][e] = 0.f;
      gemm_wq<C / 16, NMB>(acc, wq, nrow_of(nb), nb + nbstep < nblocks ? nrow_of(nb + nbstep) : (chain ? chain : nrow_of(nb)), &Xs[r * LDX + 8 * h], LDX, 0);
      STAMP(34 + 8 * q + (nb - nb0) / nbstep);
      if (nx.y_bf16 && nb * 32 + 31 < nx.N && (nx.ldy & 7) == 0 && ((uintptr_t)nx.y & 15) == 0) {
        // bf16 projection output through a wave-private LDS tile (the H buffers are idle now): the accumulator layout (token on
        // the lane) stores 8 bytes per lane into 32 different rows; staged, every lane stores 16 bytes and a row's 32 columns
        // leave as one 64-B piece.
        unsigned short* tile = &Hs[0][0] + wave * (NMB * 32 * YLD);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = acc[mb][4 * g + e] + b1s[nb * 32 + 8 * g + 4 * h + e];
            *reinterpret_cast<u32x2*>(&tile[(mb * 32 + r) * YLD + 8 * g + 4 * h]) = __builtin_bit_cast(u32x2, __builtin_convertvector(o, bf16x4));
          }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        unsigned short* yb = reinterpret_cast<unsigned short*>(nx.y) + nb * 32;
#pragma unroll
        for (int k = 0; k < NMB * 2; ++k) {
          const int pc = lane + 64 * k, row = pc >> 2, c = pc & 3;
          const long long m = grow(row);
          if (m >= nx.row_lo && m < nx.row_hi)
            *reinterpret_cast<u32x4*>(yb + m * (long long)nx.ldy + c * 8) = *reinterpret_cast<const u32x4*>(&tile[row * YLD + c * 8]);
        }
        continue;
      }
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) {
        const long long m = grow(mb * 32 + r);
        if (m < nx.row_lo || m >= nx.row_hi) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = nb * 32 + 8 * g + 4 * h;
          if (n + 3 < nx.N) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = acc[mb][4 * g + e] + b1s[n + e];
            store_act4(nx.y, m * (long long)nx.ldy + n, o, nx.y_bf16);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < nx.N) store_act(nx.y, m * (long long)nx.ldy + n + e, acc[mb][4 * g + e] + b1s[n + e], nx.y_bf16);
          }
        }
      }
    }
    STAMP(39 + 8 * q);
  }
  STAMP(63);
}

// [N][ld] row-major bf16 -> fragment-major [ceil(N/32)][K/16][64 lanes][8]: lane (r = l&31, h = l>>5) of fragment
// (nb, ks) holds W[nb*32 + r][ks*16 + 8h .. +7] (zero rows past N), i.e. exactly the MFMA A operand, so that a wave
// reads each fragment as one coalesced 1-KiB load and every byte of a cache line is consumed by a single instruction.
__global__ void pack_frag_kernel(const unsigned short* __restrict__ w, int ld, int N, int K, unsigned short* __restrict__ out) {
  const long long total = (long long)((N + 31) / 32) * (K / 16) * 64;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const long long f = i >> 6;
    const int ks = (int)(f % (K / 16));
    const int nb = (int)(f / (K / 16));
    const int n = nb * 32 + (lane & 31), k = ks * 16 + 8 * (lane >> 5);
    u32x4 v = (u32x4){0u, 0u, 0u, 0u};
    if (n < N) v = *reinterpret_cast<const u32x4*>(w + (long long)n * ld + k);
    *reinterpret_cast<u32x4*>(out + i * 8) = v;
  }
}

}  // namespace

#ifdef MVT_STAMPS
extern "C" int mvt_debug_read_stamps(unsigned long long* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(mvt_stamp_buf), sizeof(unsigned long long) * 2 * 8 * 64) == hipSuccess ? MVT_OK : MVT_ERR_HIP_BASE;
}
extern "C" int mvt_debug_clear_stamps() {
  static unsigned long long z[2 * 8 * 64];
  return hipMemcpyToSymbol(HIP_SYMBOL(mvt_stamp_buf), z, sizeof(z)) == hipSuccess ? MVT_OK : MVT_ERR_HIP_BASE;
}
#endif

extern "C" int mvt_pack_frag_bf16(const unsigned short* w, int ld, int N, int K, unsigned short* out, void* stream) {
  MVT_REQUIRE(w && out && N > 0 && K > 0 && K % 16 == 0 && ld % 8 == 0 && ld >= K);
  MVT_REQUIRE(((uintptr_t)w % 16 == 0) && ((uintptr_t)out % 16 == 0));
  const long long total = (long long)((N + 31) / 32) * (K / 16) * 64;
  hipLaunchKernelGGL(pack_frag_kernel, dim3((unsigned)mvt_cdiv(total, 256)), dim3(256), 0, mvt_stream(stream), w, ld, N, K, out);
  return mvt_launch_status();
}

extern "C" int mvt_block_fused_bf16(float* x, int ldx, const void* att, int att_bf16, int ldatt, int Ko, const unsigned short* wo, int ldwo,
                                    const float* bo, const unsigned short* w1, int ldw1, const float* b1, const unsigned short* w2,
                                    int ldw2, const float* b2, int H, const mvt_block_next* next, int n_next, long long M, int Cc,
                                    float* workspace, void* stream) {
  MVT_REQUIRE(x && w1 && b1 && w2 && b2 && M > 0 && Cc == C && H > 0 && H % 256 == 0 && H <= 4 * C);
  MVT_REQUIRE(ldx % 4 == 0 && ldx >= C);
  MVT_REQUIRE(!att || (wo && bo && Ko == 288 && ldatt % 4 == 0 && ldatt >= Ko));
  MVT_REQUIRE(n_next >= 0 && n_next <= MVT_BLOCK_MAX_NEXT && (n_next == 0 || next));
  MVT_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)att % 16 == 0) && ((uintptr_t)wo % 16 == 0) && ((uintptr_t)w1 % 16 == 0) &&
              ((uintptr_t)w2 % 16 == 0));
  BlockArgs a{};
  a.x = x; a.ldx = ldx; a.att = (const float*)att; a.att_bf16 = att_bf16 ? 1 : 0; a.ldatt = ldatt; a.Ko = Ko; a.wo = wo; a.bo = bo; a.ldwo = ldwo;
  a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.ldw1 = ldw1; a.ldw2 = ldw2; a.H = H; a.M = M; a.n_next = n_next;
  for (int q = 0; q < n_next; ++q) {
    const mvt_block_next& nx = next[q];
    MVT_REQUIRE(nx.w && nx.b && nx.y && nx.N > 0 && nx.N <= 4 * C && nx.ldy % 4 == 0 && nx.ldy >= nx.N);
    MVT_REQUIRE(nx.y_bf16 == 0 || nx.y_bf16 == 1);
    MVT_REQUIRE((nx.lnw == nullptr) == (nx.lnb == nullptr) && ((uintptr_t)nx.lnw % 16 == 0) && ((uintptr_t)nx.lnb % 16 == 0) && ((uintptr_t)nx.w % 16 == 0) && ((uintptr_t)nx.y % 16 == 0));
    MVT_REQUIRE(nx.row_lo >= 0 && (nx.row_hi == 0 || nx.row_hi > nx.row_lo));
    a.next[q] = nx;
    if (nx.row_hi == 0) a.next[q].row_hi = M;
  }
  static const char* force = getenv("MVT_BLOCK_NMB");  // tuning override
  const int nmb = force ? atoi(force) : (M >= 4096 ? 2 : 1);  // 64-row workgroups measured 1.6x faster than 128-row ones at M = 12288
  static const bool no_split = getenv("MVT_BLOCK_NOSPLIT") != nullptr;
  a.ws = workspace;
  if (nmb == 2) {
    hipLaunchKernelGGL((block_fused_bf16<2, 0, 0>), dim3((unsigned)mvt_cdiv(M, 64)), dim3(NT), 0, mvt_stream(stream), a);
  } else if (workspace && !no_split && M <= 2048 && M % 32 == 0) {  // (whole 32-row tiles: the workspace is tile-native)
    MVT_REQUIRE((uintptr_t)workspace % 16 == 0);
    const unsigned tiles = (unsigned)mvt_cdiv(M, 32);
    hipLaunchKernelGGL((block_fused_bf16<1, 1, 0>), dim3(tiles, (unsigned)(H / 256)), dim3(NT), 0, mvt_stream(stream), a);
    int maxblk = 1;
    for (int q = 0; q < n_next; ++q) maxblk = (next[q].N + 31) / 32 > maxblk ? (next[q].N + 31) / 32 : maxblk;
    const unsigned slices = n_next ? (unsigned)mvt_cdiv(maxblk, 8) : 1u;
    hipLaunchKernelGGL((block_fused_bf16<1, 2, 0>), dim3(tiles, slices), dim3(NT), 0, mvt_stream(stream), a);
  } else {
    hipLaunchKernelGGL((block_fused_bf16<1, 0, 0>), dim3((unsigned)mvt_cdiv(M, 32)), dim3(NT), 0, mvt_stream(stream), a);
  }
  return mvt_launch_status();
}

// The block with its attention inside (ATT 1 / 2 of block_fused_bf16).
extern "C" int mvt_attn_block_fused_bf16(float* x, int ldx, const mvt_block_attn* attn, const unsigned short* wo, const float* bo,
                                         const unsigned short* w1, const float* b1, const unsigned short* w2, const float* b2, int H,
                                         const mvt_block_next* next, int n_next, long long M, int Cc, float* workspace, void* stream) {
  MVT_REQUIRE(x && attn && wo && bo && w1 && b1 && w2 && b2 && M > 0 && Cc == C && H > 0 && H % 256 == 0 && H <= 4 * C);
  MVT_REQUIRE(ldx % 4 == 0 && ldx >= C && n_next >= 0 && n_next <= MVT_BLOCK_MAX_NEXT && (n_next == 0 || next));
  MVT_REQUIRE(attn->heads == 6 && attn->dim_head == DHA && attn->S >= 1 && M % attn->S == 0);
  MVT_REQUIRE(attn->kind == MVT_ATTN_PARTIALS || (attn->kind == MVT_ATTN_FRAME_CTX && attn->q && attn->ldq % 8 == 0 && attn->ldq >= 288) ||
              (attn->q && attn->k && attn->v && attn->ldq % 8 == 0 && attn->ldkv % 8 == 0 && attn->ldq >= 288 && attn->ldkv >= 288));
  MVT_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)wo % 16 == 0) && ((uintptr_t)w1 % 16 == 0) && ((uintptr_t)w2 % 16 == 0));
  MVT_REQUIRE(((uintptr_t)attn->q % 16 == 0) && ((uintptr_t)attn->k % 16 == 0) && ((uintptr_t)attn->v % 16 == 0));
  BlockArgs a{};
  a.x = x; a.ldx = ldx; a.att = nullptr; a.Ko = 288; a.wo = wo; a.bo = bo; a.ldwo = 288;
  a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.ldw1 = C; a.ldw2 = H; a.H = H; a.M = M; a.n_next = n_next; a.ws = workspace;
  a.aq = attn->q; a.ak = attn->k; a.av = attn->v; a.ldaq = attn->ldq; a.ldakv = attn->ldkv; a.S = attn->S; a.nkeys = attn->n_keys;
  for (int q = 0; q < n_next; ++q) {
    const mvt_block_next& nx = next[q];
    MVT_REQUIRE(nx.w && nx.b && nx.y && nx.N > 0 && nx.N <= 4 * C && nx.ldy % 4 == 0 && nx.ldy >= nx.N);
    MVT_REQUIRE((nx.y_bf16 == 0 || nx.y_bf16 == 1) && (nx.lnw == nullptr) == (nx.lnb == nullptr) && ((uintptr_t)nx.lnw % 16 == 0) && ((uintptr_t)nx.lnb % 16 == 0));
    MVT_REQUIRE(((uintptr_t)nx.w % 16 == 0) && ((uintptr_t)nx.y % 16 == 0) && nx.row_lo >= 0 && (nx.row_hi == 0 || nx.row_hi > nx.row_lo));
    a.next[q] = nx;
    if (nx.row_hi == 0) a.next[q].row_hi = M;
  }
  const int S = attn->S;
  if (attn->kind == MVT_ATTN_TIME) {
    // tiles of whole tracks: S <= 32 keys per track (one MFMA key block), 64 / S tracks per 64-row tile
    MVT_REQUIRE(S <= 32 && !workspace);
    // Round 4, the small-M form: 32-row tiles (S = 12: two whole tracks, 24 rows) while they fit ONE round of workgroups at one per
    // CU -- twice the weight bytes in total, half the work per workgroup: a whole updater call 903 -> 819 us at 342 tracks, 932 -> 861
    // at 448; at 512 tracks (288 workgroups) 929 -> 977: not there.  MVT_TIME_NMB1=0: the 64-row tiles everywhere.
    static const bool small_tiles = !(getenv("MVT_TIME_NMB1") && atoi(getenv("MVT_TIME_NMB1")) == 0);
    if (small_tiles && mvt_cdiv(M, (32 / S) * S) <= 256) {
      a.bmv = (32 / S) * S;
      hipLaunchKernelGGL((block_fused_bf16<1, 0, 1>), dim3((unsigned)mvt_cdiv(M, a.bmv)), dim3(NT), 0, mvt_stream(stream), a);
      return mvt_launch_status();
    }
    a.bmv = (64 / S) * S;
    hipLaunchKernelGGL((block_fused_bf16<2, 0, 1>), dim3((unsigned)mvt_cdiv(M, a.bmv)), dim3(NT), 0, mvt_stream(stream), a);
  } else if (attn->kind == MVT_ATTN_PARTIALS) {
    // attention tile from the key-split partials (64 queries = the virtual tokens, frame = group): split path only
    MVT_REQUIRE(attn->partials && attn->n_splits >= 1 && attn->n_splits <= MVT_ATTN_NSPLIT && attn->n_keys == 64 && M == 64LL * S);
    MVT_REQUIRE(workspace && (uintptr_t)workspace % 16 == 0 && M <= 2048 && (uintptr_t)attn->partials % 16 == 0);
    a.parts = attn->partials; a.nsplit = attn->n_splits;
    const unsigned tiles = (unsigned)mvt_cdiv(M, 32);
    a.S = S;
    hipLaunchKernelGGL((block_fused_bf16<1, 1, 3>), dim3(2, (unsigned)(H / 256), (unsigned)S), dim3(NT), 0, mvt_stream(stream), a);
    int maxblk = 1;
    for (int q = 0; q < n_next; ++q) maxblk = (next[q].N + 31) / 32 > maxblk ? (next[q].N + 31) / 32 : maxblk;
    const unsigned slices = n_next ? (unsigned)mvt_cdiv(maxblk, 8) : 1u;
    (void)tiles;  // pass 2 on the SAME frame-major tiles (the workspace is tile-native)
    if (!attn->defer_pass2)
      hipLaunchKernelGGL((block_fused_bf16<1, 2, 5>), dim3(2, slices, (unsigned)S), dim3(NT), 0, mvt_stream(stream), a);
  } else if (attn->kind == MVT_ATTN_FRAME) {
    MVT_REQUIRE(attn->n_keys >= 1 && attn->n_keys <= 64);
    const long long ntok = M / S;
    // Round 4, the small-M form (BASELINE config C5: 512 tracks per GPU = 6 144 point rows): 32-token tiles.  At 64 tokens per tile
    // the launch is 96 workgroups on 256 CUs; 192 workgroups of 32 tokens stream twice the weight bytes in total but finish the launch
    // sooner -- a whole updater call 929 -> 882 us (tools/time_updater.py 512).  MVT_FRAME_NMB1=0: the 64-token tiles.
    // Only while the 32-token tiles still fit ONE round of workgroups at one per CU (680 tracks: 264 workgroups, 927 -> 1 026 us).
    static const bool small_tiles = !(getenv("MVT_FRAME_NMB1") && atoi(getenv("MVT_FRAME_NMB1")) == 0);
    if (ntok * S >= 4096 && small_tiles && mvt_cdiv(ntok, 32) * S <= 256) {
      MVT_REQUIRE(!workspace);
      hipLaunchKernelGGL((block_fused_bf16<1, 0, 2>), dim3((unsigned)mvt_cdiv(ntok, 32), 1, (unsigned)S), dim3(NT), 0, mvt_stream(stream), a);
    } else if (ntok * S >= 4096) {
      MVT_REQUIRE(!workspace);
      hipLaunchKernelGGL((block_fused_bf16<2, 0, 2>), dim3((unsigned)mvt_cdiv(ntok, 64), 1, (unsigned)S), dim3(NT), 0, mvt_stream(stream), a);
    } else {
      // few rows (the 64 virtual tracks): two-launch split path; pass 1 (frame-major tiles, attention recomputed by each of the
      // H / 256 chunk workgroups: it is tiny) leaves x and the MLP partials in the workspace by GLOBAL row, pass 2 is unchanged
      MVT_REQUIRE(workspace && (uintptr_t)workspace % 16 == 0 && M <= 2048 && ntok % 32 == 0);  // (whole tiles: tile-native workspace)
      hipLaunchKernelGGL((block_fused_bf16<1, 1, 2>), dim3((unsigned)(ntok / 32), (unsigned)(H / 256), (unsigned)S), dim3(NT), 0,
                         mvt_stream(stream), a);
      int maxblk = 1;
      for (int q = 0; q < n_next; ++q) maxblk = (next[q].N + 31) / 32 > maxblk ? (next[q].N + 31) / 32 : maxblk;
      const unsigned slices = n_next ? (unsigned)mvt_cdiv(maxblk, 8) : 1u;
      if (!attn->defer_pass2)
        hipLaunchKernelGGL((block_fused_bf16<1, 2, 5>), dim3((unsigned)(ntok / 32), slices, (unsigned)S), dim3(NT), 0, mvt_stream(stream), a);
    }
  } else if (attn->kind == MVT_ATTN_FRAME_CTX) {
    // per-frame attention whose 64 context tokens are the rows of a split-path block with a deferred pass 2 (ATT 6)
    const mvt_block_ctx* cx = attn->ctx;
    const long long ntok = M / S;
    MVT_REQUIRE(cx && attn->n_keys == 64 && ntok * S >= 4096 && !workspace && !attn->defer_pass2);
    MVT_REQUIRE(cx->ws && cx->b2 && cx->x && cx->chunks >= 1 && cx->chunks <= 4 && cx->ldx % 4 == 0 && cx->ldx >= C);
    MVT_REQUIRE(((uintptr_t)cx->ws % 16 == 0) && ((uintptr_t)cx->b2 % 16 == 0) && ((uintptr_t)cx->x % 16 == 0));
    MVT_REQUIRE(cx->kv.w && cx->kv.b && cx->kv.N == 2 * 288 && (cx->kv.lnw == nullptr) == (cx->kv.lnb == nullptr));
    MVT_REQUIRE(((uintptr_t)cx->kv.w % 16 == 0) && ((uintptr_t)cx->kv.lnw % 16 == 0) && ((uintptr_t)cx->kv.lnb % 16 == 0));
    const unsigned tiles = (unsigned)mvt_cdiv(ntok, 64);
    if (cx->next.w) {
      const mvt_block_next& nx = cx->next;
      MVT_REQUIRE(nx.b && nx.y && nx.N > 0 && nx.N <= 4 * C && nx.ldy % 4 == 0 && nx.ldy >= nx.N && (nx.y_bf16 == 0 || nx.y_bf16 == 1));
      MVT_REQUIRE((nx.lnw == nullptr) == (nx.lnb == nullptr) && ((uintptr_t)nx.lnw % 16 == 0) && ((uintptr_t)nx.lnb % 16 == 0));
      MVT_REQUIRE(((uintptr_t)nx.w % 16 == 0) && ((uintptr_t)nx.y % 16 == 0) && (nx.N + 31) / 32 <= 8 * (int)tiles);
    }
    a.c_ws = cx->ws; a.c_b2 = cx->b2; a.c_x = cx->x; a.c_ldx = cx->ldx; a.c_nch = cx->chunks; a.c_kv = cx->kv; a.c_next = cx->next;
    hipLaunchKernelGGL((block_fused_bf16<2, 0, 6>), dim3(tiles, 1, (unsigned)S), dim3(NT), 0, mvt_stream(stream), a);
  } else {
    return MVT_ERR_ARG;
  }
  return mvt_launch_status();
}

// Input transform + virtual tokens + first projections in one launch (ATT 4 of MODE 3).
extern "C" int mvt_input_proj_bf16(const float* tokens, int ldtok, int token_dim, long long Mp, const unsigned short* win, const float* bin,
                                   const float* virtual_tokens, int S, float* x, int ldx, const mvt_block_next* next, int n_next, long long M,
                                   int Cc, void* stream) {
  MVT_REQUIRE(tokens && win && bin && virtual_tokens && x && next && n_next >= 1 && n_next <= MVT_BLOCK_MAX_NEXT && Cc == C);
  MVT_REQUIRE(M > 0 && Mp >= 0 && Mp <= M && S >= 1 && (M - Mp) % S == 0 && token_dim >= 1 && token_dim <= 37 * 16 && ldtok >= token_dim);
  MVT_REQUIRE(ldtok % 4 == 0 && ldx % 4 == 0 && ldx >= C && ((uintptr_t)tokens % 16 == 0) && ((uintptr_t)win % 16 == 0) && ((uintptr_t)x % 16 == 0));
  BlockArgs a{};
  a.x = x; a.ldx = ldx; a.M = M; a.n_next = n_next; a.H = 4 * C; a.ws = nullptr; a.S = S;
  a.tokx = tokens; a.ldtok = ldtok; a.win = win; a.bin = bin; a.virt = virtual_tokens; a.Mp = Mp;
  for (int q = 0; q < n_next; ++q) {
    const mvt_block_next& nx = next[q];
    MVT_REQUIRE(nx.w && nx.b && nx.y && nx.N > 0 && nx.N <= 4 * C && nx.ldy % 4 == 0 && nx.ldy >= nx.N);
    MVT_REQUIRE((nx.lnw == nullptr) == (nx.lnb == nullptr) && ((uintptr_t)nx.lnw % 16 == 0) && ((uintptr_t)nx.lnb % 16 == 0) && ((uintptr_t)nx.w % 16 == 0) && ((uintptr_t)nx.y % 16 == 0));
    MVT_REQUIRE(nx.row_lo >= 0 && (nx.row_hi == 0 || nx.row_hi > nx.row_lo) && (nx.y_bf16 == 0 || nx.y_bf16 == 1));
    a.next[q] = nx;
    if (nx.row_hi == 0) a.next[q].row_hi = M;
  }
  hipLaunchKernelGGL((block_fused_bf16<2, 3, 4>), dim3((unsigned)mvt_cdiv(M, 64)), dim3(NT), 0, mvt_stream(stream), a);
  return mvt_launch_status();
}

// ... the same with the token rows assembled in the kernel (no token matrix): mvt_token_assemble + mvt_input_proj_bf16 in one launch.
extern "C" int mvt_token_input_proj_bf16(const float* coords, const float* fcorr, int Fc, const float* ffeats, int Cf, const float* mask_vis,
                                         const float* pos, const float* time_embed, int n_tracks, int S, int E,
                                         const unsigned short* win, const float* bin, const float* virtual_tokens, float* x, int ldx,
                                         const mvt_block_next* next, int n_next, long long M, int Cc, void* stream) {
  MVT_REQUIRE(coords && fcorr && ffeats && mask_vis && pos && time_embed && win && bin && virtual_tokens && x && next);
  MVT_REQUIRE(n_next >= 1 && n_next <= MVT_BLOCK_MAX_NEXT && Cc == C && n_tracks > 0 && S >= 1 && E > 0 && E % 2 == 0 && Fc > 0 && Cf > 0);
  const int D = 3 * E + 3 + Fc + Cf + 2;
  const long long Mp = (long long)n_tracks * S;
  MVT_REQUIRE(D <= 37 * 16 && M >= Mp && (M - Mp) % S == 0 && ldx % 4 == 0 && ldx >= C);
  MVT_REQUIRE(((uintptr_t)win % 16 == 0) && ((uintptr_t)x % 16 == 0));
  BlockArgs a{};
  a.x = x; a.ldx = ldx; a.M = M; a.n_next = n_next; a.H = 4 * C; a.ws = nullptr; a.S = S;
  a.tokx = nullptr; a.win = win; a.bin = bin; a.virt = virtual_tokens; a.Mp = Mp;
  a.t_coords = coords; a.t_fcorr = fcorr; a.t_ffeats = ffeats; a.t_maskvis = mask_vis; a.t_pos = pos; a.t_time = time_embed;
  a.t_E = E; a.t_Fc = Fc; a.t_Cf = Cf; a.t_D = D;
  for (int q = 0; q < n_next; ++q) {
    const mvt_block_next& nx = next[q];
    MVT_REQUIRE(nx.w && nx.b && nx.y && nx.N > 0 && nx.N <= 4 * C && nx.ldy % 4 == 0 && nx.ldy >= nx.N);
    MVT_REQUIRE((nx.lnw == nullptr) == (nx.lnb == nullptr) && ((uintptr_t)nx.lnw % 16 == 0) && ((uintptr_t)nx.lnb % 16 == 0) && ((uintptr_t)nx.w % 16 == 0) && ((uintptr_t)nx.y % 16 == 0));
    MVT_REQUIRE(nx.row_lo >= 0 && (nx.row_hi == 0 || nx.row_hi > nx.row_lo) && (nx.y_bf16 == 0 || nx.y_bf16 == 1));
    a.next[q] = nx;
    if (nx.row_hi == 0) a.next[q].row_hi = M;
  }
  hipLaunchKernelGGL((block_fused_bf16<2, 3, 4>), dim3((unsigned)mvt_cdiv(M, 64)), dim3(NT), 0, mvt_stream(stream), a);
  return mvt_launch_status();
}

// LayerNorm + projections only: y_i = LayerNorm_i(x) . Wn_i^T + bn_i (the first time-attention q|k|v of an updater call, which no
// preceding block can produce): pass 2 of the split path without a workspace -- x is read, never written.
extern "C" int mvt_ln_proj_bf16(const float* x, int ldx, const mvt_block_next* next, int n_next, long long M, int Cc, void* stream) {
  MVT_REQUIRE(x && next && n_next >= 1 && n_next <= MVT_BLOCK_MAX_NEXT && M > 0 && Cc == C && ldx % 4 == 0 && ldx >= C);
  MVT_REQUIRE((uintptr_t)x % 16 == 0);
  BlockArgs a{};
  a.x = const_cast<float*>(x); a.ldx = ldx; a.M = M; a.n_next = n_next; a.H = 4 * C; a.ws = nullptr;
  int maxblk = 1;
  for (int q = 0; q < n_next; ++q) {
    const mvt_block_next& nx = next[q];
    MVT_REQUIRE(nx.w && nx.b && nx.y && nx.N > 0 && nx.N <= 4 * C && nx.ldy % 4 == 0 && nx.ldy >= nx.N);
    MVT_REQUIRE((nx.lnw == nullptr) == (nx.lnb == nullptr) && ((uintptr_t)nx.lnw % 16 == 0) && ((uintptr_t)nx.lnb % 16 == 0) && ((uintptr_t)nx.w % 16 == 0) && ((uintptr_t)nx.y % 16 == 0));
    MVT_REQUIRE(nx.row_lo >= 0 && (nx.row_hi == 0 || nx.row_hi > nx.row_lo) && (nx.y_bf16 == 0 || nx.y_bf16 == 1));
    a.next[q] = nx;
    if (nx.row_hi == 0) a.next[q].row_hi = M;
    maxblk = (nx.N + 31) / 32 > maxblk ? (nx.N + 31) / 32 : maxblk;
  }
  if (M >= 4096)  // whole 64-row tiles, every column block in the workgroup: x and its LayerNorm are read / computed once per tile
    hipLaunchKernelGGL((block_fused_bf16<2, 3, 0>), dim3((unsigned)mvt_cdiv(M, 64)), dim3(NT), 0, mvt_stream(stream), a);
  else
    hipLaunchKernelGGL((block_fused_bf16<1, 2, 0>), dim3((unsigned)mvt_cdiv(M, 32), (unsigned)mvt_cdiv(maxblk, 8)), dim3(NT), 0,
                       mvt_stream(stream), a);
  return mvt_launch_status();
}
