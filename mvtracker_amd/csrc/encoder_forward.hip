// Composite entry point of the CNN encoder: BasicEncoder.forward (spatracker/blocks.py:214-284) on n images as ONE C call
// (SURVEY.md section 8b).  No kernels of its own: it sequences the library's bf16 kernels -- row-tile convolutions with the
// InstanceNorm statistics taken in their epilogues and InstanceNorm + ReLU of the producer applied while a consumer loads its
// patch, the per-tile statistics reduction, the residual InstanceNorm pass, the one-launch concat -- on the caller's stream over a
// caller-provided workspace.  bf16 mode, bf16 activations (the configuration bench.py measures).  56 launches per call.
#include "common.h"
#include <stdlib.h>

namespace {

inline long long align256(long long b) { return (b + 255) & ~255LL; }

struct Plan {
  long long stem, z0, y, s[4], d, cat, c2, part, st, total;  // byte offsets
};

// workspace: bf16 activations + statistics scratch.  stem / z0 / y: one 64-channel map at H/2 each (the largest activation);
// s[l]: the output of stage l, alive until the concat; d: the 1x1 downsample branch of a stage's first block.
inline Plan plan(long long n, int H, int W, int C) {
  const long long h2 = H / 2, w2 = W / 2, hs = H / 4, ws = W / 4;
  Plan P{};
  long long o = 0;
  auto take = [&](long long bytes) { const long long at = o; o += align256(bytes); return at; };
  const long long big = n * h2 * w2 * 64 * 2;
  P.stem = take(big);
  P.z0 = take(big);
  P.y = take(big);
  const int couts[4] = {64, 96, 128, 128};
  long long hh = h2, ww = w2;
  for (int l = 0; l < 4; ++l) {
    if (l) {
      hh = (hh - 1) / 2 + 1;
      ww = (ww - 1) / 2 + 1;
    }
    P.s[l] = take(n * hh * ww * couts[l] * 2);
  }
  P.d = take(n * hs * ws * 96 * 2);
  P.cat = take(n * hs * ws * 416 * 2);
  P.c2 = take(n * hs * ws * 2 * C * 2);
  // statistics: per conv n * slots * Cout * 2 floats, slots <= tiles of the H/2 map
  const long long slots_max = ((h2 + 7) / 8) * ((w2 + 31) / 32);
  P.part = take(n * slots_max * 256 * 2 * 4);
  P.st = take(8 * n * 256 * 2 * 4);                // up to 8 live (mean, rstd) tables
  P.total = o;
  return P;
}

}  // namespace

extern "C" long long mvt_encoder_workspace_bytes(int n, int H, int W, int C) {
  if (n <= 0 || H < 16 || W < 16 || C <= 0) return -1;
  return plan(n, H, W, C).total;
}

#define ENC_TRY(call)              \
  do {                             \
    const int rc_ = (call);        \
    if (rc_ != MVT_OK) return rc_; \
  } while (0)

int mvt_detail_conv_rows_slots(int Ho, int Wo, int tile_rows);
int mvt_detail_conv_rows_tile_rows(int ksize, int stride, int Ho);
int mvt_detail_stem7x7_rows_src(const float* in, const void* rgb, int rgb_u8, int rgb_V, int rgb_T, long long rgb_img0, const unsigned short* w,
                                int ldw, const float* bias, void* out, int n, int H, int W, int Cout, int ldo, int io_flags, float* out_partial,
                                hipStream_t stream);

// x4 != null: normalised [n][H][W][4] input; else the planar clip rgb (V,T,3,H,W), images img0 .. img0 + n - 1 frame-major
static int encoder_run(const mvt_encoder_weights* w, const float* x4, const void* rgb, int rgb_u8, int rgb_V, int rgb_T, long long img0,
                       int n, int H, int W, void* out_rows, int ldo, int out_bf16, void* workspace, long long workspace_bytes, void* stream) {
  MVT_REQUIRE(w && (x4 || rgb) && out_rows && workspace && n > 0 && H >= 16 && W >= 16 && H % 4 == 0 && W % 4 == 0);
  const int C = w->latent_dim;
  MVT_REQUIRE(C > 0 && C % 32 == 0 && ldo >= C && (out_bf16 == 0 || out_bf16 == 1));
  const Plan P = plan(n, H, W, C);
  MVT_REQUIRE(workspace_bytes >= P.total && ((uintptr_t)workspace % 256) == 0);
  for (int i = 0; i < MVT_ENCODER_CONVS; ++i) MVT_REQUIRE(w->conv[i].w && w->conv[i].b);
  char* base = (char*)workspace;
  float* part = (float*)(base + P.part);
  float* stb = (float*)(base + P.st);
  const long long st_stride = (long long)n * 256 * 2;
  int st_next = 0;
  auto new_st = [&]() { float* s = stb + (st_next % 8) * st_stride; ++st_next; return s; };
  const int BF = MVT_IO_IN_BF16 | MVT_IO_OUT_BF16;

  // one convolution + the (mean, rstd) of its output
  auto conv = [&](int ci, const void* in, int io_in, void* out, int hh, int ww, int cin, int cout, int k, int stride, int pad, int ld_out,
                  const float* in_stats, float** st_out, int io_out) -> int {
    const mvt_conv_weights& cw = w->conv[ci];
    const int ho = (hh + 2 * pad - k) / stride + 1, wo = (ww + 2 * pad - k) / stride + 1;
    float* pp = nullptr;
    int slots = 0;
    if (st_out) {
      slots = mvt_conv2d_stat_slots(hh, ww, cin, k, k, stride, pad, 0);
      MVT_REQUIRE(slots > 0);
      pp = part;
    }
    ENC_TRY(mvt_conv2d_bf16(in, cw.w, nullptr, cw.b, out, n, hh, ww, cin, cout, k, k, stride, pad, ld_out, MVT_ACT_NONE,
                            io_in | io_out | (w->short_workgroups ? MVT_IO_SHORT_WG : 0), in_stats, pp, stream));
    if (st_out) {
      *st_out = new_st();
      ENC_TRY(mvt_instnorm_finish_slots(pp, slots, *st_out, n, (long long)ho * wo, cout, stream));
    }
    return MVT_OK;
  };

  // stem: 7x7 / 2, its InstanceNorm + ReLU is applied by its two consumers (conv1 of layer1.0 and that block's skip)
  const int h2 = H / 2, w2 = W / 2;
  float* st_stem = nullptr;
  if (x4) {
    ENC_TRY(conv(0, x4, 0, base + P.stem, H, W, 4, 64, 7, 2, 3, 64, nullptr, &st_stem, MVT_IO_OUT_BF16));
  } else {
    // the stem reads the clip's planar frames itself (normalisation on load): no [n][H][W][4] staging tensor, one launch less
    const mvt_conv_weights& cw = w->conv[0];
    const int slots = mvt_detail_conv_rows_slots(h2, w2, mvt_detail_conv_rows_tile_rows(7, 2, h2));
    // (weight rows as mvt_conv2d_bf16 takes them: [64][round_up(7 * 32, 64)])
    ENC_TRY(mvt_detail_stem7x7_rows_src(nullptr, rgb, rgb_u8, rgb_V, rgb_T, img0, cw.w, (7 * 32 + 63) & ~63, cw.b, base + P.stem, n, H, W, 64, 64,
                                        MVT_IO_OUT_BF16, part, mvt_stream(stream)));
    st_stem = new_st();
    ENC_TRY(mvt_instnorm_finish_slots(part, slots, st_stem, n, (long long)h2 * w2, 64, stream));
  }

  // ResidualBlock (blocks.py:84-128): conv1 -> IN -> ReLU -> conv2 -> IN -> ReLU, + (downsampled, normalised) input, ReLU
  auto res_block = [&](int c1, int c2, int cd, const void* xin, const float* x_stats, int hh, int ww, int cin, int cout, int stride,
                       void* zout) -> int {
    const int ho = (hh + 2 - 3) / stride + 1, wo = (ww + 2 - 3) / stride + 1;
    float *s1 = nullptr, *s2 = nullptr, *sd = nullptr;
    static const bool no_fold = getenv("MVT_FOLD_DOWNSAMPLE") && atoi(getenv("MVT_FOLD_DOWNSAMPLE")) == 0;  // (A/B runs)
    const bool fold = cd >= 0 && stride == 2 && !x_stats && cout % 32 == 0 && !no_fold;  // conv1 + downsample[0] read the same x: one launch
    if (fold) {
      const int slots = mvt_conv2d_stat_slots(hh, ww, cin, 3, 3, 2, 1, 0);
      MVT_REQUIRE(slots > 0);
      float* part_d = part + (long long)n * slots * cout * 2;  // (the partials buffer is sized for the stem: room for both)
      ENC_TRY(mvt_conv3x3s2_down_bf16(xin, w->conv[c1].w, w->conv[c1].b, w->conv[cd].w, w->conv[cd].b, base + P.y, base + P.d, n, hh, ww, cin,
                                      cout, cout, part, part_d, stream));
      s1 = new_st();
      ENC_TRY(mvt_instnorm_finish_slots(part, slots, s1, n, (long long)ho * wo, cout, stream));
      sd = new_st();
      ENC_TRY(mvt_instnorm_finish_slots(part_d, slots, sd, n, (long long)ho * wo, cout, stream));
    } else {
      ENC_TRY(conv(c1, xin, MVT_IO_IN_BF16, base + P.y, hh, ww, cin, cout, 3, stride, 1, cout, x_stats, &s1, MVT_IO_OUT_BF16));
    }
    ENC_TRY(conv(c2, base + P.y, MVT_IO_IN_BF16, zout, ho, wo, cout, cout, 3, 1, 1, cout, s1, &s2, MVT_IO_OUT_BF16));
    const void* skip = xin;
    const float* skip_stats = x_stats;
    int flags = BF | (x_stats ? MVT_APPLY_SKIP_RELU : 0);
    if (cd >= 0) {
      if (!fold) ENC_TRY(conv(cd, xin, MVT_IO_IN_BF16, base + P.d, hh, ww, cin, cout, 1, stride, 0, cout, nullptr, &sd, MVT_IO_OUT_BF16));
      skip = base + P.d;
      skip_stats = sd;
      flags = BF;
    }
    return mvt_instnorm_apply(zout, s2, skip, skip_stats, zout, n, (long long)ho * wo, cout, flags, stream);
  };

  // conv indices: 0 stem; layer l (1..4): 1 + 5 (l - 1) + {0: .0.conv1, 1: .0.conv2, 2: .0.downsample, 3: .1.conv1, 4: .1.conv2};
  // 21: conv2 (3x3, 416 -> 2C); 22: conv3 (1x1, 2C -> C)
  const void* stage[4];
  int sh[4], sw[4], sc[4];
  const void* xin = base + P.stem;
  const float* xst = st_stem;
  int hh = h2, ww = w2, cin = 64;
  const int couts[4] = {64, 96, 128, 128}, strides[4] = {1, 2, 2, 2};
  for (int l = 0; l < 4; ++l) {
    const int b0 = 1 + 5 * l, cout = couts[l], stride = strides[l];
    ENC_TRY(res_block(b0, b0 + 1, l == 0 ? -1 : b0 + 2, xin, xst, hh, ww, cin, cout, stride, base + P.z0));
    hh = (hh + 2 - 3) / stride + 1;
    ww = (ww + 2 - 3) / stride + 1;
    ENC_TRY(res_block(b0 + 3, b0 + 4, -1, base + P.z0, nullptr, hh, ww, cout, cout, 1, base + P.s[l]));
    stage[l] = base + P.s[l]; sh[l] = hh; sw[l] = ww; sc[l] = cout;
    xin = base + P.s[l]; xst = nullptr; cin = cout;
  }
  const int hs = H / 4, ws = W / 4;
  ENC_TRY(mvt_concat_resize_bilinear_ac(4, stage, sh, sw, sc, base + P.cat, n, hs, ws, 416, BF, stream));
  float* st_c2 = nullptr;
  ENC_TRY(conv(21, base + P.cat, MVT_IO_IN_BF16, base + P.c2, hs, ws, 416, 2 * C, 3, 1, 1, 2 * C, nullptr, &st_c2, MVT_IO_OUT_BF16));
  ENC_TRY(conv(22, base + P.c2, MVT_IO_IN_BF16, out_rows, hs, ws, 2 * C, C, 1, 1, 0, ldo, st_c2, nullptr, out_bf16 ? MVT_IO_OUT_BF16 : 0));
  return MVT_OK;
}

extern "C" int mvt_encoder_forward(const mvt_encoder_weights* w, const float* x4, int n, int H, int W, void* out_rows, int ldo,
                                   int out_bf16, void* workspace, long long workspace_bytes, void* stream) {
  MVT_REQUIRE(x4);
  return encoder_run(w, x4, nullptr, 0, 0, 0, 0, n, H, W, out_rows, ldo, out_bf16, workspace, workspace_bytes, stream);
}

extern "C" int mvt_encoder_forward_rgb(const mvt_encoder_weights* w, const void* rgbs, int is_u8, int V, int T, long long img0, int n, int H,
                                       int W, void* out_rows, int ldo, int out_bf16, void* workspace, long long workspace_bytes,
                                       void* stream) {
  MVT_REQUIRE(rgbs && (is_u8 == 0 || is_u8 == 1) && V > 0 && T > 0 && img0 >= 0 && n > 0 && img0 + n <= (long long)V * T);
  return encoder_run(w, nullptr, rgbs, is_u8, V, T, img0, n, H, W, out_rows, ldo, out_bf16, workspace, workspace_bytes, stream);
}
