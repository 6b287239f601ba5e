/*
 * mvtracker_hip.h -- C ABI of libmvtracker_hip.so, the MI355X (gfx950) implementation of the
 * multi-view point-tracking forward path.
 *
 * The reference (samiazirar/mvtracker) is pure Python on torch ATen; it has no FFI of its own
 * for this path.  The entry points below are the set of operations its hot path performs
 * (SURVEY.md section 8a/8b); each one names the reference code it replaces (paths relative to
 * the reference root).  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 / int32 / uint64 data unless marked "host";
 *     tensors are dense row-major with the layouts stated per function;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued
 *     asynchronously on it; no entry point allocates, frees or synchronises;
 *   - return value: 0 = ok, MVT_ERR_ARG = rejected arguments (nothing launched),
 *     1000 + hipError_t = launch failure; no exceptions, no global state, re-entrant per stream;
 *   - inputs are borrowed and never written; outputs are fully overwritten unless stated.
 *
 * Internal clip layout ("frame store"): features of pyramid level l are kept frame-major as
 *   fvec_l [T][V][h_l][w_l][C]  (fp32, channels last), xyz_l [T][V][h_l][w_l][4] (x,y,z,0),
 * so the fused point cloud of frame t (reference: init_pointcloud_from_rgbd,
 * mvtracker/models/core/model_utils.py:420-482, layout (B*S, V*H*W, C)) is the contiguous slice
 * [t] with P_l = V*h_l*w_l points.  Track state is track-major: coords [N][S][3],
 * ffeats [N][S][C], tokens [N][S][D].
 */
#ifndef MVTRACKER_HIP_H
#define MVTRACKER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVT_OK 0
#define MVT_ERR_ARG 1
#define MVT_ERR_HIP_BASE 1000

/* activation codes of mvt_gemm / mvt_conv2d epilogues */
#define MVT_ACT_NONE 0
#define MVT_ACT_RELU 1
#define MVT_ACT_GELU_TANH 2 /* nn.GELU(approximate="tanh"), cotracker2/blocks.py:289 */
#define MVT_ACT_GELU_ERF 3  /* nn.GELU(), mvtracker.py:179 */

/* library / device introspection (host) */
int mvt_abi_version(void);
const char* mvt_build_arch(void); /* "gfx950" */

/* ---------------------------------------------------------------------------------------------
 * Dense GEMM on the matrix cores (fp32 MFMA 32x32x2, exact fp32 FMA chains).
 *   C[M][ldc] = R[M][ldr] (optional) + act(A[M][lda] . Wt[N][ldw]^T + bias[N] (optional))
 * A rows must be readable for round_up(K,4) floats; Wt is the torch nn.Linear weight layout
 * [out][in] repacked so that ldw = round_up(K,32) with zero padding.  Replaces every
 * nn.Linear of the updater (cotracker2/blocks.py:55-67, 254-271, 364-382, 456, 489) and
 * ffeats_updater (mvtracker.py:179, 396).  R may alias C.
 * --------------------------------------------------------------------------------------------- */
int mvt_gemm(const float* A, int lda, const float* Wt, int ldw, const float* bias, const float* R, int ldr,
             float* C, int ldc, int M, int N, int K, int act, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 2-D convolution as implicit GEMM on the matrix cores, channels-last.
 *   in  [n][H][W][Cin]   (Cin % 32 == 0, or Cin == 4 for the 7x7 stem with kw*4+c packing)
 *   wt  [Cout][round_up(K,64)], row = [KH][KW][Cin] (K = KH*KW*Cin) repacked from torch's
 *       [Cout][Cin][KH][KW] and zero padded; for the stem row = [KH][32] (K = KH*32) with element
 *       kw*4+c (c<3, kw<7) and zeros elsewhere
 *   out [n][Ho][Wo][ldo] (ldo >= Cout; lets a conv write into a channel slice / the frame store)
 * Zero padding `pad`, stride `stride`.  Replaces nn.Conv2d in BasicEncoder / ResidualBlock
 * (mvtracker/models/core/spatracker/blocks.py:74-82, 116, 157-193).
 * --------------------------------------------------------------------------------------------- */
int mvt_conv2d(const float* in, const float* wt, const float* bias, float* out, int n, int H, int W, int Cin,
               int Cout, int KH, int KW, int stride, int pad, int ldo, int act, void* stream);

/* ---------------------------------------------------------------------------------------------
 * bf16 matrix-core variants of the two entry points above (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
 * Activations stay fp32 in memory and are converted while staged into LDS; weights are pre-split with
 * mvt_split_bf16 into row-major bf16 [N][ldw], ldw = round_up(K,64), zero padded (conv: K = KH*KW*Cin in
 * [kh][kw][cin] order, stem K = KH*32).
 *   w_lo == NULL : plain bf16 operands (the arithmetic torch autocast gives the reference's convs/linears).
 *   w_lo != NULL : split precision "bf16x3": x = hi + lo, products hi*hi + hi*lo + lo*hi -> fp32-grade
 *                  results (dropped term 2^-16 relative) at 3/16 of the fp32-MFMA cost.
 * --------------------------------------------------------------------------------------------- */
/* hi[i] = bf16(src[i]) (round to nearest even), lo[i] = bf16(src[i] - hi[i]) (lo optional); n % 4 == 0. */
int mvt_split_bf16(const float* src, unsigned short* hi, unsigned short* lo, long long n, void* stream);
int mvt_gemm_bf16(const float* A, int lda, const unsigned short* w_hi, const unsigned short* w_lo, int ldw,
                  const float* bias, const float* R, int ldr, void* C, int ldc, int M, int N, int K, int act,
                  int io_flags /* MVT_IO_OUT_BF16: C is a bf16 tensor */, void* stream);
/* InstanceNorm fusion of the bf16 convolutions (saves the separate statistics pass and the normalise pass between
 * the two convs of a ResidualBlock, blocks.py:119-122):
 *   in_stats    [n][Cin][2] (mean, rstd) or NULL: the input is read as relu((x - mean) * rstd) (3x3 stride-1 pad-1 only);
 *   out_partial [n][slots][Cout][2] or NULL: per-channel (sum, sum of squares) of the outputs of every 32-pixel block,
 *               slots = mvt_conv2d_stat_slots(...) (0 = not available for this shape; the 3x3 kernels of the two
 *               precisions cut the image differently); one writer per slot, reduced in a
 *               fixed order by mvt_instnorm_finish_slots -> deterministic.
 *
 * io_flags: element type of the activation tensors.  In bf16 mode (wt_lo == NULL) the encoder keeps its intermediate
 * activations in bf16 -- its big layers are HBM-bound, and the MFMA operands are bf16 anyway:
 *   MVT_IO_IN_BF16  `in` is bf16 [n][H][W][Cin]   (Cin % 32 == 0; the stem always reads fp32)
 *   MVT_IO_OUT_BF16 `out` is bf16 [n][Ho][Wo][ldo] (rounded to nearest even from the fp32 accumulator; the statistics
 *                   in out_partial are taken before the rounding)
 *   MVT_IO_SHORT_WG keep every workgroup short-lived: the wide 3x3 layers (Cout % 256 == 0, no normalise-on-load) otherwise run as
 *                   one 512-thread, 144-KiB-LDS workgroup per CU and pixel tile (conv3x3_big_bf16, ~100 us each), which starves
 *                   kernels of OTHER streams of CU slots while it runs -- set it for launches that share the GPU with latency-bound
 *                   work (the encoder blocks that run beside the refinement windows).  Results are bit-identical either way. */
#define MVT_IO_IN_BF16 1
#define MVT_IO_OUT_BF16 2
#define MVT_IO_SHORT_WG 4
int mvt_conv2d_stat_slots(int H, int W, int Cin, int KH, int KW, int stride, int pad, int split /* wt_lo != NULL */);
/* The first two convolutions of a strided ResidualBlock that read the block input (spatracker/blocks.py:84-128: conv1 = 3x3 / stride 2
 * / pad 1 and downsample[0] = 1x1 / stride 2 of the SAME x) in ONE launch, bf16 mode, bf16 tensors: the downsample samples input
 * pixel (2y, 2x) = the centre tap of the 3x3 window around output (y, x), so it is computed from the patch conv1 has staged, with its
 * own weights wd [Cout][ldwd] (rows = cin), bias, output tensor and statistics -- one launch, one staging of the input and one
 * InstanceNorm-finish fewer per block.  out3 / outd [n][Ho][Wo][ldo] bf16; part3 / partd [n][slots][Cout][2] with
 * slots = mvt_conv2d_stat_slots(H, W, Cin, 3, 3, 2, 1, 0) for BOTH (both NULL: no statistics).  Cin % 32 == 0, Cout % 32 == 0.
 * Values identical to the two separate mvt_conv2d_bf16 launches (same accumulation order); the downsample's statistics are cut
 * into this kernel's 4-row tiles instead of the 1x1 kernel's 8-row tiles (another fixed summation order). */
int mvt_conv3x3s2_down_bf16(const void* in, const unsigned short* w3, const float* b3, const unsigned short* wd, const float* bd,
                            void* out3, void* outd, int n, int H, int W, int Cin, int Cout, int ldo, float* part3, float* partd,
                            void* stream);
int mvt_conv2d_bf16(const void* in, const unsigned short* wt_hi, const unsigned short* wt_lo, const float* bias,
                    void* out, int n, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int ldo,
                    int act, int io_flags, const float* in_stats, float* out_partial, void* stream);
int mvt_instnorm_finish_slots(const float* partial, int slots, float* mean_rstd, int n, long long HW, int C, void* stream);

/* mvt_gemm_bf16 with LayerNorm (biased variance, eps, optional affine ln_w / ln_b of length K) applied to every A
 * row on the fly: C = R + act(LayerNorm(A) . W^T + bias).  K % 4 == 0, K <= 1024.  Replaces the norm1 / norm_context
 * + to_q / to_kv pairs of AttnBlock / CrossAttnBlock (cotracker2/blocks.py:298, 335) without materialising the
 * normalised tokens. */
int mvt_ln_gemm_bf16(const float* A, int lda, const float* ln_w, const float* ln_b, float ln_eps,
                     const unsigned short* w_hi, const unsigned short* w_lo, int ldw, const float* bias, const float* R,
                     int ldr, float* C, int ldc, int M, int N, int K, int act, void* stream);

/* Fused transformer MLP on the bf16 matrix cores, in place:
 *   x[m][0:C] += W2 . gelu_tanh(W1 . LayerNorm(x[m][0:C]) + b1) + b2      (LayerNorm without affine, given eps)
 * (AttnBlock / CrossAttnBlock second half, cotracker2/blocks.py:299-300, 336-337).  w1 [H][ldw1] and
 * w2 [C][ldw2] are bf16 (mvt_split_bf16 hi parts); C == 256, H % 64 == 0.  The hidden activations stay in
 * registers (the first GEMM's accumulator is the second GEMM's operand). */
int mvt_mlp_fused_bf16(float* x, int ldx, const unsigned short* w1, int ldw1, const float* b1,
                       const unsigned short* w2, int ldw2, const float* b2, long long M, int C, int H, float eps,
                       void* stream);

/* Fused remainder of an AttnBlock / CrossAttnBlock on the bf16 matrix cores (cotracker2/blocks.py:297-300, 334-337),
 * in place on the token rows x [M][ldx] (C == 256):
 *     x += att . Wo^T + bo                                  (att [M][ldatt], Ko == 288 columns; att == NULL skips it)
 *     x += W2 . gelu_tanh(W1 . LayerNorm(x) + b1) + b2      (LayerNorm eps 1e-6, no affine; H % 128 == 0, H <= 1024)
 *     y_i = LayerNorm_i(x) . Wn_i^T + bn_i                  (n_next <= 2 follow-up projections: the q / kv / qkv
 *                                                            of the blocks that consume x next)
 * All weights bf16 in the fragment-major layout of mvt_pack_frag_bf16 (ldwo / ldw1 / ldw2 / next.ldw are ignored).
 * One launch replaces GEMM, LayerNorm, GEMM, GEMM, LayerNorm, GEMM [, LayerNorm, GEMM]. */
/* Row-major bf16 w [N][ld] -> fragment-major out [ceil(N/32)][K/16][64][8]: lane (r = l&31, h = l>>5) of fragment
 * (nb, ks) holds w[nb*32 + r][ks*16 + 8h .. +7] (zero rows past N) -- the MFMA 32x32x16 A operand, one coalesced 1-KiB
 * load per wave and k-step.  K % 16 == 0. */
int mvt_pack_frag_bf16(const unsigned short* w, int ld, int N, int K, unsigned short* out, void* stream);
typedef struct mvt_block_next {
  const unsigned short* w; /* fragment-major bf16 of [N][256] */
  const float* b;          /* [N] */
  const float* lnw;        /* LayerNorm affine [256] or NULL (both) */
  const float* lnb;
  float* y;                /* [M][ldy] fp32, or bf16 when y_bf16 */
  int ldw, N, ldy;
  float eps;
  long long row_lo, row_hi; /* the projection is evaluated for rows [row_lo, row_hi) only (y is indexed by the global row);
                               row_hi == 0: every row */
  int y_bf16;               /* 1: y is a bf16 tensor (the q / k / v operands of mvt_attention_bf16) */
} mvt_block_next;
#define MVT_BLOCK_MAX_NEXT 3
int mvt_block_fused_bf16(float* x, int ldx, const void* att /* fp32, or bf16 when att_bf16 */, int att_bf16, int ldatt, int Ko,
                         const unsigned short* wo, int ldwo,
                         const float* bo, const unsigned short* w1, int ldw1, const float* b1, const unsigned short* w2,
                         int ldw2, const float* b2, int H, const mvt_block_next* next, int n_next, long long M, int C,
                         float* workspace /* NULL, or (H/256 + 1) * M * C floats: lets small M (<= 2048 rows, M % 32 == 0) run
                                             as two launches cut over 4x more workgroups (deterministic partial sums); scratch
                                             in the kernels' own tile order, no state between calls */,
                         void* stream);

/* mvt_block_fused_bf16 with the attention that precedes the block computed INSIDE the kernel (no attention launch, no
 * attention tensor in HBM); heads = 6, dim_head = 48.  x rows are track-major, row = token * S + frame.
 *   MVT_ATTN_TIME  (AttnBlock over time, cotracker2/blocks.py:464-467): every token attends over its own S <= 32 frames;
 *                  q, k, v [M][ld] bf16 indexed by the block's rows.
 *   MVT_ATTN_FRAME (CrossAttnBlock / AttnBlock over space, blocks.py:477-483): the tokens of frame t attend over the
 *                  n_keys <= 64 context tokens of frame t; q [M][ldq] indexed by the block's rows, k / v [n_keys * S][ldkv]
 *                  with context token j of frame t at row j * S + t.  M < 4096: two-launch split path, workspace required,
 *                  (M / S) % 32 == 0. */
#define MVT_ATTN_TIME 1
#define MVT_ATTN_FRAME 2
/*   MVT_ATTN_PARTIALS: the attention was computed key-split by mvt_attention_bf16(MVT_ATTN_PARTIALS_ONLY) (64 queries per
 *                  frame, the virtual tokens: M = 64 * S); the kernel combines the n_splits partial states itself -- same
 *                  arithmetic and order as the merge launch it replaces -- instead of reading a merged attention tensor. */
#define MVT_ATTN_PARTIALS 3
/*   MVT_ATTN_FRAME_CTX: MVT_ATTN_FRAME (M >= 4096, n_keys == 64) whose context tokens are the rows of a split-path block called
 *                  with defer_pass2: that block's second launch is skipped and every workgroup here finishes the context rows of
 *                  its frame itself (x = x_mid + b2 + partial sums in pass 2's order, LayerNorm, k|v projection -- bit-identical
 *                  to the pass-2 launch), keeps k|v in LDS, and the context's x / one further projection are written once per
 *                  frame.  k, v are unused.  One launch and one write-then-read of k|v less per updater layer. */
#define MVT_ATTN_FRAME_CTX 4
typedef struct mvt_block_ctx {
  const float* ws;     /* the `workspace` of the deferred block: (chunks + 1) * 64 * S * C floats */
  int chunks;          /* its H / 256 */
  const float* b2;     /* its fc2 bias */
  float* x;            /* its x rows [64 * S][ldx] (row = token * S + frame): final values are written here */
  int ldx;
  mvt_block_next kv;   /* LayerNorm + k|v projection of the context rows: N = 576 (k | v, 6 heads x 48); y is not written */
  mvt_block_next next; /* optional (w == NULL: none): one more projection of the context rows, written to y [64 * S][ldy] */
} mvt_block_ctx;
typedef struct mvt_block_attn {
  int kind, S, n_keys, heads, dim_head, ldq, ldkv;
  const unsigned short* q;
  const unsigned short* k;
  const unsigned short* v;
  const float* partials; /* MVT_ATTN_PARTIALS: the workspace of mvt_attention_bf16 */
  int n_splits;
  int defer_pass2;       /* split-path forms (workspace given, M < 4096): launch pass 1 only; x and the follow-up projections are
                            finished by the consumer (MVT_ATTN_FRAME_CTX) */
  const mvt_block_ctx* ctx; /* MVT_ATTN_FRAME_CTX */
} mvt_block_attn;
int mvt_attn_block_fused_bf16(float* x, int ldx, const mvt_block_attn* attn, const unsigned short* wo, const float* bo,
                              const unsigned short* w1, const float* b1, const unsigned short* w2, const float* b2, int H,
                              const mvt_block_next* next, int n_next, long long M, int C, float* workspace, void* stream);

/* The updater's input in one launch (cotracker2/blocks.py:456-459 + the first projection): rows < Mp of x [M][ldx] =
 * tokens [Mp][ldtok] (token_dim <= 592 columns used) . Win^T + bin, rows >= Mp = virtual_tokens[(row - Mp) / S]; x is written,
 * then y_i = LayerNorm_i(x) . Wn_i^T + bn_i as in mvt_ln_proj_bf16.  win: fragment-major bf16 of [256][592] (zero padded). */
int mvt_input_proj_bf16(const float* tokens, int ldtok, int token_dim, long long Mp, const unsigned short* win, const float* bin,
                        const float* virtual_tokens, int S, float* x, int ldx, const mvt_block_next* next, int n_next, long long M, int C,
                        void* stream);

/* LayerNorm + projections only (x is read, never written): y_i = LayerNorm_i(x) . Wn_i^T + bn_i with the mvt_block_next
 * descriptors of mvt_block_fused_bf16 -- the first q|k|v projection of an updater call. */
int mvt_ln_proj_bf16(const float* x, int ldx, const mvt_block_next* next, int n_next, long long M, int C, void* stream);

/* rgbs [V][T][3][H][W] (values 0..255) -> x [T_sel][V][H][W][4] = (2*(rgb/255)-1, 0) for frames
 * t0..t0+nt-1 (mvtracker.py:565-567 normalisation + channels-last repack). */
int mvt_rgb_to_nhwc4(const float* rgbs, float* out, int V, int T, int H, int W, int t0, int nt, void* stream);
/* same from uint8 frames, the storage type of the sample files (demo.py:650, 922-929): the clip stays 1 byte per value in HBM
 * and over PCIe; (float)u8 is exact, so the result is bit-identical to converting first. */
int mvt_rgb_u8_to_nhwc4(const unsigned char* rgbs, float* out, int V, int T, int H, int W, int t0, int nt, void* stream);
/* same for an arbitrary run of images numbered frame-major (image t * V + v = view v of frame t, the order of the frame store):
 * images img0 .. img0+nimg-1 -> out [nimg][H][W][4].  The multi-GPU encoder split cuts the V*T images -- not frames -- evenly
 * across ranks.  is_u8: rgbs holds bytes. */
int mvt_rgb_images_to_nhwc4(const void* rgbs, int is_u8, float* out, int V, int T, int H, int W, long long img0, int nimg, void* stream);

/* nearest-neighbour resize of [n][C][Hi][Wi] planes to [n][C][Ho][Wo] with torch's index rule
 * (evaluation_predictor_3dpt.py:76-81, F.interpolate(mode="nearest")). */
int mvt_resize_nearest(const float* in, float* out, long long planes, int Hi, int Wi, int Ho, int Wo, void* stream);

/* InstanceNorm2d (eps 1e-5, biased variance, no affine; blocks.py:99-103,150-152) on
 * channels-last x [n][HW][C]:  stats -> mean_rstd [n][C][2]; `partial` is scratch of
 * n*MVT_IN_SLABS*C*2 doubles. */
#define MVT_IN_SLABS 64
int mvt_instnorm_stats(const void* x, int ldx, double* partial, float* mean_rstd, int n, long long HW, int C,
                       int io_flags /* MVT_IO_IN_BF16: x is bf16 */, void* stream);
/* y = relu((x-mean)*rstd)                                  (skip == NULL)
 * y = relu(skip' + relu((x-mean)*rstd)), skip' = skip or (skip-mean_s)*rstd_s when skip_stats
 * is given (ResidualBlock.forward, blocks.py:119-128); with MVT_APPLY_SKIP_RELU in io_flags skip' = relu((skip-mean_s)*rstd_s),
 * i.e. the skip tensor is itself a raw conv output whose norm + ReLU was never materialised (the stem).  y may alias x. */
#define MVT_APPLY_SKIP_RELU 4
int mvt_instnorm_apply(const void* x, const float* mean_rstd, const void* skip, const float* skip_stats, void* y,
                       int n, long long HW, int C, int io_flags /* 0, or IN|OUT: x, skip and y are all bf16 */, void* stream);

/* bilinear resize, align_corners=True, of channels-last src [n][Hs][Ws][C] into the channel
 * slice [c_off, c_off+C) of dst [n][Hd][Wd][ldd] (blocks.py:254-280, the 416-channel concat). */
int mvt_resize_bilinear_ac(const void* src, void* dst, int n, int Hs, int Ws, int C, int Hd, int Wd, int ldd,
                           int c_off, int io_flags /* 0, or IN|OUT: src and dst are bf16 */, void* stream);
/* The encoder's concat in ONE launch (spatracker/blocks.py:266-276): nsrc <= 4 stage outputs [n][Hs_k][Ws_k][C_k], each resized
 * like mvt_resize_bilinear_ac, written side by side (channel offsets 0, C_0, C_0 + C_1, ...) into dst [n][Hd][Wd][ldd]; a wave
 * writes whole contiguous pixel rows.  srcs / Hs / Ws / Cs: HOST arrays.  io_flags: 0 (fp32) or IN_BF16 | OUT_BF16. */
int mvt_concat_resize_bilinear_ac(int nsrc, const void* const* srcs, const int* Hs, const int* Ws, const int* Cs, void* dst, int n,
                                  int Hd, int Wd, int ldd, int io_flags, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Frame store construction (reference: init_pointcloud_from_rgbd, model_utils.py:420-482, and
 * the depth resampling of mvtracker.py:558-562).
 * --------------------------------------------------------------------------------------------- */
/* intrs [n][3][3], extrs [n][3][4] (world->camera) -> kinv [n][9], einv [n][12] (rows 0..2 of the
 * inverse 4x4), closed form in fp64 rounded to fp32 (model_utils.py:453-457). */
int mvt_invert_cameras(const float* intrs, const float* extrs, float* kinv, float* einv, int n, void* stream);
/* depths [V][T][H][W] -> level-0 strided depth [T][V][H/s][W/s], nearest: source pixel (s*i, s*j)
 * (mvtracker.py:558-562). */
int mvt_depth_subsample(const float* depths, float* out, int V, int T, int H, int W, int s, void* stream);
/* View assignment of MonocularToMultiViewAdapter (monocular_baselines.py:630-680): for every query
 * (t, x, y, z) the view whose depth map, sampled bilinearly at the query's projection at frame t, exceeds the query's camera
 * depth the most (-1e4 for projections outside the image or behind the camera); the first maximum wins.
 * depths [V][T][H][W], intrs [V][T][3][3], extrs [V][T][3][4], query_points [N][4]; view_out [N] int32;
 * xyz_out NULL or [V][N][3] = (pixel x, pixel y, camera z) of every query in every view. */
int mvt_adapter_best_view(const float* depths, const float* intrs, const float* extrs, const float* query_points, int V, int T,
                          int H, int W, int N, int* view_out, float* xyz_out, void* stream);
/* 2x2 average pool of channels-last [n][h][w][C] -> [n][h/2][w/2][C] (model_utils.py:440).  io_flags: 0 (fp32 rows) or
 * MVT_IO_IN_BF16 | MVT_IO_OUT_BF16 (the bf16 frame store of bf16 mode -- under autocast the reference's fmaps are bf16 and
 * every pooled level is rounded to bf16 again, model_utils.py:436-440; the 2x2 mean itself is taken in fp32). */
int mvt_avgpool2(const void* in, void* out, long long n, int h, int w, int C, int io_flags, void* stream);
/* xyz_l [T][V][h_l][w_l][4] from level-0 strided depth [T][V][hs][ws] (nearest-subsampled by
 * 2^level, model_utils.py:443-444), pixel grid (i+0.5)*stride*2^level-0.5 (:462-466), kinv/einv
 * indexed [v*T+t] (:467-473). */
int mvt_unproject(const float* depth_s, const float* kinv, const float* einv, float* xyz, int V, int T, int hs,
                  int ws, int stride, int level, void* stream);

/* ---------------------------------------------------------------------------------------------
 * kNN + correlation (reference: knn / PointcloudCorrBlock.corr_sample, mvtracker.py:26-90,
 * 800-846; feature init 1-NN, mvtracker.py:627-643).
 * --------------------------------------------------------------------------------------------- */
/* Exact brute-force kNN.  For every (query n, slot s): candidates are the P points of frame
 * frame_of_slot = min(frame0 + s*frame_step, T-1) of xyz [T][P][4]; query = coords[(n*S+s)*3..].
 * d2 = fma(dz,dz,fma(dy,dy,dx*dx)), neighbours ascending by (d2, index).  The candidate range
 * is cut into nseg segments scanned by different waves; keys [N][S][nseg][K] receive
 * (d2 bits << 32 | index) per segment, ascending (KEY_MAX = ~0 pads a segment that holds fewer than K
 * survivors of a seeded scan).  1 <= K <= 16, P >= K.
 * Optional pruning seed (exactness preserved): seed_idx [N][S][seed_k] int32, seed_k >= K distinct point
 * indices per (query, slot) -- e.g. the previous iteration's neighbours (seed_cw == 0), or the neighbours found
 * on the next coarser pyramid level (seed_cw, seed_ch = coarse per-view grid; seed_fw, seed_fh = this level's
 * per-view grid; coarse (v,y,x) maps to (v,2y,2x)).  The scan then starts from the bound
 * max_j d2(query, seed_j) >= (K-th nearest distance) instead of +inf.
 * Optional tile culling (exactness preserved): the scan walks the cloud in tiles of 64 points and skips a tile
 * whose bounding box (tile_box, from mvt_tile_aabb with the same grid_w/grid_h) lies farther than the current bound
 * from all queries of the wave; the box distance is evaluated with the scan's own arithmetic, every rounding step
 * of which is monotonic, so it never exceeds the d2 of a point inside the box.  grid_w = grid_h = 0: tile t = points
 * [64t, 64t+64); grid_w, grid_h > 0 (multiples of 8, P = views*grid_h*grid_w in raster order): 8x8 pixel patches.
 * The segments are runs of tiles; only the merged result (mvt_knn_merge) is layout independent. */
int mvt_tile_aabb(const float* xyz, long long P, int T, int grid_w, int grid_h,
                  float* box /* [T][ceil(P/64)][8] = lo.xyz, finite-point count, hi.xyz, 0 */,
                  void* stream);
/* Coarse level of the culling hierarchy: group_box [T][ceil(ntiles/64)][8] = the union of the boxes of 64 consecutive tiles
 * (same record layout).  Used by the single-segment searches below (clouds of 65..4096 tiles), which test the group boxes first
 * and skip whole runs of tiles; exactness is unaffected (a group box contains its tiles' boxes). */
int mvt_tile_group_aabb(const float* box, long long P, int T, float* group_box, void* stream);
int mvt_knn_scan(const float* xyz, long long P, const float* coords, int N, int S, int frame0, int frame_step,
                 int T, int K, int nseg, unsigned long long* keys, const int* seed_idx, int seed_k, int seed_cw,
                 int seed_ch, int seed_fw, int seed_fh, const float* tile_box, int grid_w, int grid_h, void* stream);
/* Merge the nseg per-segment key lists of mvt_knn_scan: idx_out [N][S][K] int32 = the K nearest neighbour
 * indices, ascending by (d2, index); indices are clamped to [0, P) (only NaN queries can be out of range). */
int mvt_knn_merge(const unsigned long long* keys, int N, int S, int K, int nseg, long long P, int* idx_out,
                  void* stream);
/* mvt_knn_scan (one segment) + mvt_knn_merge in ONE launch: every (track, slot) is searched by a single wave over the whole cloud
 * and its K neighbour indices go straight to idx_out [N][S][K] -- bit for bit the merged result of the two-launch form.  tile_box
 * required, group_box optional; seeds as in mvt_knn_scan (NULL: unseeded).  idx_out must not alias seed_idx when the seed comes
 * from another level (seed_cw > 0 reads other entries' indices only of the coarser tensor, never of idx_out). */
int mvt_knn_search(const float* xyz, long long P, const float* coords, int N, int S, int frame0, int frame_step, int T, int K,
                   const int* seed_idx, int seed_k, int seed_cw, int seed_ch, int seed_fw, int seed_fh, const float* tile_box,
                   const float* group_box, int grid_w, int grid_h, int* idx_out, void* stream);
/* Scan / merge of several pyramid levels in ONE launch each (grid.y = level).  After the first refinement iteration every
 * level is seeded by its own previous neighbours (seed_idx [N][S][seed_k], same-level indices), so the scans are
 * independent; one launch removes three launch / tail latencies per iteration.  Semantics per level = mvt_knn_scan /
 * mvt_knn_merge.  seed_k == 0: unseeded (every seed_idx NULL). */
typedef struct mvt_knn_level {
  const float* xyz;         /* [T][P][4] */
  long long P;
  unsigned long long* keys; /* [N][S][nseg][K] */
  const int* seed_idx;      /* [N][S][seed_k] or NULL */
  const float* tile_box;    /* from mvt_tile_aabb(grid_w, grid_h) or NULL */
  int nseg, grid_w, grid_h;
  int* idx_out;             /* [N][S][K] (merge) */
  const float* group_box;   /* from mvt_tile_group_aabb or NULL: coarse culling level of mvt_knn_search_levels */
} mvt_knn_level;
int mvt_knn_scan_levels(int levels, const mvt_knn_level* lv, const float* coords, int N, int S, int frame0, int frame_step,
                        int T, int K, int seed_k, void* stream);
int mvt_knn_merge_levels(int levels, const mvt_knn_level* lv, int N, int S, int K, void* stream);
/* Seeded scan + merge of all levels in ONE launch: every (track, slot) is searched by a single wave over the whole cloud
 * (tile boxes required; nseg and keys are ignored) and its K neighbour indices go straight to idx_out -- the result of
 * mvt_knn_scan_levels + mvt_knn_merge_levels, bit for bit (exact kNN, ties by index).  idx_out may alias seed_idx: a wave reads
 * the seeds of its own entries before it writes them.  seed_k >= K, or seed_k == 0 with every seed_idx NULL: unseeded searches
 * (the initial threshold is the farthest-corner bound of the nearest tile with >= K finite points). */
int mvt_knn_search_levels(int levels, const mvt_knn_level* lv, const float* coords, int N, int S, int frame0, int frame_step,
                          int T, int K, int seed_k, void* stream);
/* Gather-dot correlation for `levels` pyramid levels in ONE launch (grid.y = level).  Host arrays of per-level
 * DEVICE pointers: xyz[l] [T][P_l][4], fvec[l] [T][P_l][C] (C in {32,64,128,256}, groups == 1), idx[l]
 * [N][S][K] int32 from mvt_knn_merge, P[l].  For k < K:
 *   out[(n*S+s)*ldo + o_off + l*4K + 4k + {0,1,2,3}] = { <target, f_k>/sqrt(C), xyz_k - coord }
 * (mvtracker.py:827-842 with corr_add_neighbor_offset=True; the level-major concat of :374).  targets [N][S][C] fp32.
 * fvec_bf16: the feature rows are bf16 (256-B rows at C = 128; the dot accumulates in fp32). */
int mvt_corr_gather_dot(int levels, const float* const* xyz, const void* const* fvec, int fvec_bf16, const long long* P,
                        const int* const* idx, int C, const float* targets, const float* coords, int N, int S,
                        int frame0, int frame_step, int T, int K, float* out, int ldo, int o_off, void* stream);
/* The same operator with the reference's non-default correlation options (mvtracker.py:130-149, 832-846): `groups` grouped dots per
 * neighbour (each over C / groups channels, / sqrt(C / groups); a power of two dividing the lanes of a feature row), the neighbour
 * offset (add_offset) and / or the neighbour's coordinates (add_xyz) appended: OW = groups + 3 add_offset + 3 add_xyz values per
 * neighbour at out[(n*S+s)*ldo + o_off + (l*K + k)*OW ...]. */
int mvt_corr_gather_dot_opts(int levels, const float* const* xyz, const void* const* fvec, int fvec_bf16, const long long* P,
                             const int* const* idx, int C, const float* targets, const float* coords, int N, int S, int frame0,
                             int frame_step, int T, int K, int groups, int add_offset, int add_xyz, float* out, int ldo, int o_off,
                             void* stream);
/* 1-NN feature init: feat_out[n][C] (fp32) = fvec[frame][idx] with idx from keys [n][1][nseg][1]
 * (mvtracker.py:640-643); idx_out optional.  fvec_bf16: bf16 feature rows. */
int mvt_knn1_gather(const void* fvec, int fvec_bf16, long long P, int C, const unsigned long long* keys, int n, int nseg,
                    int frame, float* feat_out, int* idx_out, void* stream);

/* Secondary operator: bilinear-window correlation (CorrBlock.corr_sample,
 * spatracker/blocks.py:492-533 + bilinear_sampler :604-619).  fmap_l channels-last
 * [BS][h][w][C] for ONE level; targets [BS][N][C]; coords [BS][N][2] level-0 pixels;
 * out[(bs*N+n)*ldo + o_off + i*(2r+1)+j] for the (2r+1)^2 window around coords/2^level. */
int mvt_window_corr(const float* fmap, const float* targets, const float* coords, float* out, int BS, int N,
                    int C, int h, int w, int level, int radius, int ldo, int o_off, void* stream);
/* ... all pyramid levels of CorrBlock.corr_sample in ONE launch (the form the primary operator mvt_corr_gather_dot has), fp32 or
 * bf16 maps (fmap_bf16: under autocast the reference's CorrBlock holds bf16 fmaps and bf16 avg-pooled levels, blocks.py:423-449;
 * fp32 accumulation here).  fmaps[l] channels-last [BS][hs[l]][ws[l]][C]; out[(bs*N+n)*ldo + o_off + l*(2r+1)^2 + i*(2r+1)+j]. */
int mvt_window_corr_levels(int levels, const void* const* fmaps, int fmap_bf16, const int* hs, const int* ws, const float* targets,
                           const float* coords, float* out, int BS, int N, int C, int radius, int ldo, int o_off, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Token assembly and track update (mvtracker.py:324-349, 374-408; embeddings.py:35-50,
 * 88-106, 134-161).
 * --------------------------------------------------------------------------------------------- */
/* pos [N][D] = first D entries of the 3x(dim3/3) sin|cos embedding (dim_padded wide) of coords0 (row n at
 * coords[(n*S)*3], fp64 math, omega_j = 10000^(-j/(dim3/6))) (embeddings.py:35-50).  omega: NULL (computed in the kernel) or a
 * DEVICE table of dim_padded/6 doubles -- the reference's own numpy values (embeddings.py:95-97). */
int mvt_pos_embed(const float* coords, int N, int S, int D, int dim_padded, const double* omega, float* pos, void* stream);
/* x[(n*S+s)*ldx + ..] = cat[flow_embed(coords[n,s]-coords[n,0]) (3*E+3) | fcorr (Fc) |
 * ffeats (C) | mask | vis] + pos[n] + time[s]  (mvtracker.py:379-386). */
int mvt_token_assemble(const float* coords, const float* fcorr, int Fc, const float* ffeats, int C,
                       const float* mask_vis, const float* pos, const float* time_embed, int N, int S, int E,
                       float* x, int ldx, void* stream);
/* delta [N*S][ldd] -> coords += delta[:, 0:3]; dn [N*S][C] = GroupNorm(1,C)(delta[:, 3:3+C])
 * (eps 1e-5, affine gw/gb) (mvtracker.py:392-395, 398).  nan_flag (optional int*) is set to 1
 * if any updated coordinate is NaN (deferred form of the guard at :401-404). */
int mvt_delta_split(const float* delta, int ldd, const float* gw, const float* gb, float* coords, float* dn,
                    long long rows, int C, int* nan_flag, void* stream);
/* out[r] = <x[r][0:C], w> + b  (vis_predictor, mvtracker.py:180, 408). */
int mvt_rowdot(const float* x, int ldx, const float* w, const float* b, float* out, long long rows, int C,
               void* stream);

/* ---------------------------------------------------------------------------------------------
 * Updater transformer pieces (cotracker2/blocks.py:246-337, 455-494).
 * --------------------------------------------------------------------------------------------- */
/* y[r] = LayerNorm(x[r][0:C]) * w + b (w,b optional), biased variance, given eps (:284-287, 314-315). */
int mvt_layernorm(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, long long rows,
                  int C, float eps, void* stream);
/* softmax(q k^T / sqrt(dh)) v for `groups` independent groups, `heads` heads of width dh <= 64.
 * Row r of query item i in group g is  q + (g*q_gs + i*q_is) * ldq  (+ head*dh); likewise k, v, o.
 * One wave per (group, head, query); keys on lanes, online softmax in fp32
 * (FlashAttention.forward, blocks.py:258-271, attn_mask None). */
int mvt_attention(const float* q, int ldq, long long q_gs, long long q_is, const float* k, const float* v,
                  int ldkv, long long k_gs, long long k_is, float* o, int ldo, int groups, int nq, int nk,
                  int heads, int dh, void* stream);
/* Same contract on the bf16 matrix cores (flash-style, scores never leave registers; dh == 48): q/k/v are rounded to
 * bf16, accumulation and softmax in fp32. */
int mvt_attention_bf16(const void* q, int ldq, long long q_gs, long long q_is, const void* k, const void* v,
                       int ldkv, long long k_gs, long long k_is, void* o, int ldo, int groups, int nq, int nk,
                       int heads, int dh, int io_flags /* MVT_IO_IN_BF16: q, k, v are bf16 tensors; MVT_IO_OUT_BF16: o is */,
                       float* workspace /* NULL, or 4 * groups*heads*ceil(nq/64) * 4352 floats: lets a long-key attention
                                           with few (group, head) chunks cut its keys over 4 workgroups per chunk */,
                       void* stream);
/* io_flags bit: stop after the key-split partials (no merge launch, `o` is not written); the consumer combines them
 * (mvt_attn_block_fused_bf16, MVT_ATTN_PARTIALS).  Only valid when the key-split path is taken (nk >= 512, < 256 chunks). */
#define MVT_ATTN_PARTIALS_ONLY 8
#define MVT_ATTN_NSPLIT 4 /* key splits of that path */
/* x[(n*S+s)*ld + 0:C] = v[n][0:C] for all s  (virtual-token broadcast, blocks.py:458-459). */
int mvt_broadcast_rows(const float* v, float* x, int ld, int n, int S, int C, void* stream);

/* Track state of one sliding window (mvtracker.py:505-511, 645-655, 695): n tracks (sorted by query frame), the first p0 of
 * them carried over from the previous window (prev_coords [p0][S][3], prev_vis [p0][S] logits; window stride S/2).
 * qxyz [n][3], qt [n] int32 query frames, feat_init [n][C].  Writes coords [n][S][3], mask_vis [n][S][2] = (track mask,
 * initial visibility), ffeats [n][S][C].  Window slot s reads frame min(w + s, T - 1). */
int mvt_window_prepare(const float* qxyz, const int* qt, const float* feat_init, const float* prev_coords, const float* prev_vis,
                       int n, int p0, int S, int C, int w, int T, float* coords, float* mask_vis, float* ffeats, void* stream);
/* Results of the window into the clip outputs in the caller's query order (mvtracker.py:692-693, 710-711): for s < min(S, T - w)
 * traj[(w+s)][order[i]] = coords[i][s], vis_logit = vis[i][s], vis_prob = sigmoid(vis[i][s]).  traj [T][N][3], vis_* [T][N],
 * order [n] int64. */
int mvt_window_store(const float* coords, const float* vis, const long long* order, int n, int S, int w, int T, int N, float* traj,
                     float* vis_logit, float* vis_prob, void* stream);

/* Per-track evaluation metrics (mvtracker/evaluation/metrics.py:10-58, 61-171, 327-330; query_mode "first"): one row of
 * 11 + 2K floats per track -- movement, visible frames, occlusion accuracy (all / gt-occluded / gt-visible), average Jaccard,
 * average pts-within, MTE (lower median), ATE, FDE, survival, then pts-within[K] and Jaccard[K] (fractions, NaN where the
 * reference divides by zero).  gt_tracks / pred_tracks [T][N][D] (D = 2 or 3), gt_visible / pred_occluded [T][N] bytes,
 * query_frame [N] int32, thresholds: HOST array of K <= 8 distances.  T <= 1024. */
int mvt_track_metrics(const float* gt_tracks, const float* pred_tracks, const unsigned char* gt_visible,
                      const unsigned char* pred_occluded, const int* query_frame, int T, int N, int D, const float* thresholds, int K,
                      float survival_threshold, float* out, int ldo, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Composite entry points: one C call per stage of the hot path instead of one per kernel, so that a
 * non-Python caller can run the updater as ONE operation and the launch order lives in the library, not in host glue.
 *
 * mvt_updateformer_forward = EfficientUpdateFormer.forward (cotracker2/blocks.py:455-494) on the bf16 matrix cores for the
 * shipped geometry (hidden 256, 6 heads x 48, MLP 1024, 64 virtual tracks; configs/model/mvtracker.yaml): input transform,
 * virtual-token broadcast, depth x (time block, virtual<-point, virtual self, point<-virtual), flow head.
 *   x     [n*S][ldx] fp32 tokens (K = token_dim columns used), track-major (row = track * S + frame)
 *   delta [n*S][ldd] fp32, out_dim columns written
 * Weights: device pointers prepared once by the caller -- `frag` = fragment-major bf16 of mvt_pack_frag_bf16, `rows` = row-major
 * bf16 [N][ldw] (ldw = round_up(K, 64), zero padded) as mvt_gemm_bf16 takes them; biases / LayerNorm affines fp32.
 * flow0 / flow2 must be packed with N rounded up to a multiple of 4 (zero rows, zero bias) so that the pad columns of the hidden
 * activations -- read as K padding by the next layer -- are written as exact zeros by the GEMM itself.
 * `workspace`: mvt_updateformer_workspace_bytes(n, S) bytes, 256-B aligned, no state carried between calls.
 * --------------------------------------------------------------------------------------------- */
typedef struct mvt_lin_frag {
  const unsigned short* w; /* fragment-major bf16 of [N][K] */
  const float* b;          /* [N] */
  int N, K;
} mvt_lin_frag;
typedef struct mvt_lin_rows {
  const unsigned short* w; /* row-major bf16 [N][ldw] */
  const float* b;
  int N, K, ldw;
} mvt_lin_rows;
typedef struct mvt_updater_block {
  mvt_lin_frag qkv;   /* self-attention blocks (time, virtual self): fused [q|k|v] projection, N = 3 * heads * dim_head */
  mvt_lin_frag q, kv; /* cross-attention blocks: to_q on the block's own tokens, to_kv on the context tokens */
  const float* ctx_ln_w; /* cross blocks: norm_context affine (eps 1e-5, cotracker2/blocks.py:314-315) */
  const float* ctx_ln_b;
  mvt_lin_frag out, fc1, fc2;
} mvt_updater_block;
#define MVT_UPDATER_MAX_DEPTH 8
typedef struct mvt_updater_weights {
  int depth, hidden, heads, dim_head, n_virtual, S, token_dim, out_dim;
  int fuse_attention; /* bit 0: time attention, bit 1: point<-virtual, bit 2: virtual self attention run inside the block kernels
                         (mvt_attn_block_fused_bf16) instead of as separate launches; bit 3: unused (ignored); bit 4: the virtual<-point block
                         combines the key-split partials (MVT_ATTN_PARTIALS, no merge launch); bit 5 (with bits 1 and 2, >= 4096 point rows): the
                         virtual-self block's second launch runs inside the point<-virtual block (MVT_ATTN_FRAME_CTX).  Bits 4 and 5 only move
                         WHERE partial sums are combined: bit-identical with and without.  Bits 0-2 change the attention arithmetic (one
                         softmax pass over all key blocks, the time attention of a tile as one block-diagonal unit per head): results agree
                         with the separate launches to bf16 rounding (max ~7e-3 of the output scale), not bit for bit */
  const float* virtual_tokens; /* [n_virtual][hidden] */
  mvt_lin_rows input_transform, flow0, flow2, flow4;
  mvt_updater_block time_blk[MVT_UPDATER_MAX_DEPTH], v2p[MVT_UPDATER_MAX_DEPTH], vself[MVT_UPDATER_MAX_DEPTH], p2v[MVT_UPDATER_MAX_DEPTH];
  /* optional, for the fused track update of mvt_updateformer_forward (NULL w = not available): the flow head once more as
   * fragment-major bf16 (K of flow2 / flow4 padded to 144) and the feature updater of mvtracker.py:178-179, 393-399 */
  mvt_lin_frag flow0_frag, flow2_frag, flow4_frag, ffeats_updater;
  mvt_lin_frag input_frag; /* optional: input_transform once more as fragment-major bf16 with K padded to 592 (mvt_input_proj_bf16) */
  const float* ffeats_norm_w; /* GroupNorm(1, 128) affine */
  const float* ffeats_norm_b;
} mvt_updater_weights;
long long mvt_updateformer_workspace_bytes(int n, int S); /* host; -1 on bad arguments */
/* delta may be NULL when the track update is fused: with coords != NULL ([n*S][3]) and ffeats ([n*S][128]) the call also applies
 * coords += delta[:, 0:3], ffeats += gelu(Linear(GroupNorm(delta[:, 3:]))) (mvtracker.py:392-399) in the flow-head kernel
 * (mvt_update_head_bf16) and ORs 1 into *nan_flag (optional) when a coordinate became NaN (:401-404). */
int mvt_updateformer_forward(const mvt_updater_weights* w /* host struct of device pointers */, const float* x, int ldx, int n,
                             float* delta, int ldd, float* coords, float* ffeats, int* nan_flag, void* workspace,
                             long long workspace_bytes, void* stream);
/* mvt_encoder_forward = BasicEncoder.forward (spatracker/blocks.py:214-284) on n images as one call: x4 [n][H][W][4] normalised
 * RGB (mvt_rgb_*_to_nhwc4) -> out_rows [n][H/4][W/4][ldo] (latent_dim columns; bf16 when out_bf16 -- the frame store of bf16 mode).
 * bf16 mode with bf16 activations.  conv[i]: row-major bf16 weights [Cout][ldw] with rows (kh, kw, cin) as mvt_conv2d_bf16 takes them
 * (the 7x7 stem as [64][7][8][4]) + fp32 bias, in the order: 0 conv1; for layer l = 1..4: 1 + 5(l-1) + {0 .0.conv1, 1 .0.conv2,
 * 2 .0.downsample.0 (unused for l = 1), 3 .1.conv1, 4 .1.conv2}; 21 conv2; 22 conv3.  workspace: mvt_encoder_workspace_bytes bytes,
 * 256-B aligned, no state between calls. */
#define MVT_ENCODER_CONVS 23
typedef struct mvt_conv_weights {
  const unsigned short* w;
  const float* b;
} mvt_conv_weights;
typedef struct mvt_encoder_weights {
  int latent_dim;
  int short_workgroups; /* != 0: every convolution with MVT_IO_SHORT_WG (the call shares the GPU with other streams) */
  mvt_conv_weights conv[MVT_ENCODER_CONVS];
} mvt_encoder_weights;
long long mvt_encoder_workspace_bytes(int n, int H, int W, int latent_dim); /* host; -1 on bad arguments */
int mvt_encoder_forward(const mvt_encoder_weights* w, const float* x4, int n, int H, int W, void* out_rows, int ldo, int out_bf16,
                        void* workspace, long long workspace_bytes, void* stream);
/* ... the same with the stem reading the clip's planar frames itself: rgbs (V,T,3,H,W) in [0, 255], fp32 or (is_u8) uint8; the n images
 * img0 .. img0 + n - 1 in frame-major numbering (image t * V + v = view v of frame t); normalised on load exactly as mvt_rgb_*_to_nhwc4
 * does (2 (x / 255) - 1, mvtracker.py:455) -- bit-identical results without the [n][H][W][4] staging tensor and its launch. */
int mvt_encoder_forward_rgb(const mvt_encoder_weights* w, const void* rgbs, int is_u8, int V, int T, long long img0, int n, int H, int W,
                            void* out_rows, int ldo, int out_bf16, void* workspace, long long workspace_bytes, void* stream);

/* mvt_updateformer_forward with the 581-wide token rows ASSEMBLED inside its first kernel (mvt_token_input_proj_bf16: the
 * arithmetic of mvt_token_assemble) instead of read from a token matrix: one refinement iteration after the correlation is then
 * this single call.  coords [n][S][3], fcorr [n][S][Fc], ffeats [n][S][Cf], mask_vis [n][S][2], pos [n][D], time_embed [S][D],
 * D = 3E + 3 + Fc + Cf + 2 = token_dim.  Needs w->input_frag. */
typedef struct mvt_token_inputs {
  const float* coords;
  const float* fcorr;
  const float* ffeats;
  const float* mask_vis;
  const float* pos;
  const float* time_embed;
  int Fc, Cf, E;
} mvt_token_inputs;
int mvt_updateformer_forward_tokens(const mvt_updater_weights* w, const mvt_token_inputs* tokens, int n, float* delta, int ldd,
                                    float* coords, float* ffeats, int* nan_flag, void* workspace, long long workspace_bytes, void* stream);
int mvt_token_input_proj_bf16(const float* coords, const float* fcorr, int Fc, const float* ffeats, int Cf, const float* mask_vis,
                              const float* pos, const float* time_embed, int n_tracks, int S, int E, const unsigned short* win,
                              const float* bin, const float* virtual_tokens, float* x, int ldx, const mvt_block_next* next, int n_next,
                              long long M, int C, void* stream);

/* The head alone: flow head (256 -> 131 -> 131 -> 131, ReLU) on tok [rows][ldt] + track / feature update, see above.  w0 / w2 /
 * w4 / wu fragment-major bf16 of [131][256], [131][144], [131][144], [128][128] (mvt_pack_frag_bf16, zero padded). */
int mvt_update_head_bf16(const float* tok, int ldt, const unsigned short* w0, const float* b0, const unsigned short* w2,
                         const float* b2, const unsigned short* w4, const float* b4, const float* gn_w, const float* gn_b,
                         const unsigned short* wu, const float* bu, float* coords, float* ffeats, float* delta /* NULL or [rows][ldd] */,
                         int ldd, long long rows, int hidden, int out_dim, int* nan_flag, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MVTRACKER_HIP_H */
