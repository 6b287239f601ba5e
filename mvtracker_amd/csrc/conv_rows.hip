// 3x3 / stride 1 / pad 1 convolution of the encoder on the bf16 matrix cores, second generation
// (reference: the ResidualBlock convs and BasicEncoder.conv2, spatracker/blocks.py:84-128, 246-282).
//
// The first halo kernel (gemm.hip) spent ~17 VALU + 8 SALU instructions per MFMA on branchy loaders and 64-bit
// epilogue addressing and ran at 14-24 % of the MFMA peak.  This one is laid out so that the inner loop is nothing but
// ds_read_b128 (immediate offsets) and MFMAs:
//   * a workgroup owns TR = 8 image rows x 32 columns of output pixels (256 px) and BN output channels; a 32-pixel MFMA
//     row block is one image row segment, so the A fragment of tap (kh, kw) is the patch row shifted by a constant;
//   * the input patch (10 x 34 pixels x 32 channels, bf16, 80-B pixel slots: conflict-free ds_read_b128) is staged once per
//     32-channel chunk -- InstanceNorm + ReLU of the producer applied on the way when in_stats is given -- and the
//     weights one filter row (3 taps) at a time; both come through registers (global loads of the next stage are in
//     flight during the MFMAs of the current one) with branch-free clamped addressing;
//   * wave tiles are TM x TN MFMA blocks (2x2, 2x3 or 4x2), i.e. 1 - 0.75 KiB of LDS reads per MFMA;
//   * the epilogue adds the bias, writes fp32 or bf16 and emits the per-channel (sum, sum of squares) of every 32-pixel
//     row segment for the fused InstanceNorm statistics.
#include <stdlib.h>

#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct RowsArgs {
  const float* in;   // [n][H][W][Cin] fp32 or bf16
  const unsigned short* w;  // bf16 [Cout][ldw]: row = (kh, kw, cin)
  const float* bias;
  float* out;        // [n][H][W][ldo] fp32 or bf16
  const float* in_stats;  // [n][Cin][2] or null
  float* out_part;        // [n][slots][Cout][2] or null
  int H, W, Cin, Cout, ldw, ldo, slots, in_bf16, out_bf16;
};

constexpr int TR = 8;            // output rows per workgroup
constexpr int TC = 32;           // output columns per workgroup
constexpr int PR = TR + 2, PC = TC + 2;
constexpr int NPIX = PR * PC;    // 340 patch pixels
constexpr int CK = 32;           // channels per chunk
constexpr int LDP = CK + 8;      // bf16 per pixel / weight-row slot (80 B)
constexpr int MAXC = 512;        // input channels the in-kernel normalisation supports

template <int TM, int TN, int WM, int WN, bool INB>
__global__ __launch_bounds__(256) void conv3x3_rows_bf16(RowsArgs p) {
  static_assert(WM * WN == 4 && WM * TM == TR, "four waves cover the 8 x 32 pixel tile");
  constexpr int BN = WN * TN * 32;
  constexpr int NPF = (NPIX * 8 + 255) / 256;   // fp32 input: 16-B pieces (4 channels) per thread per chunk
  constexpr int NPH = (NPIX * 4 + 255) / 256;   // bf16 input: 16-B pieces (8 channels)
  constexpr int NWF = (3 * BN * 4 + 255) / 256; // 16-B weight pieces per thread per stage (3 taps x BN rows x 64 B)
  __shared__ __attribute__((aligned(16))) unsigned short Ps[NPIX * LDP];
  __shared__ __attribute__((aligned(16))) unsigned short Ws[3 * BN * LDP];
  __shared__ __attribute__((aligned(16))) float Sst[2 * MAXC];  // (mean, rstd) of the input channels of this image

  const int t = threadIdx.x;
  const int tiles_x = (p.W + TC - 1) / TC, tiles_y = (p.H + TR - 1) / TR;
  const int tiles_n = (p.Cout + BN - 1) / BN;
  int b = blockIdx.x;
  const int tn = b % tiles_n; b /= tiles_n;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y;
  const long long img = b / tiles_y;
  const int y0 = ty * TR, x0 = tx * TC, n0 = tn * BN;
  const long long in_img = img * (long long)p.H * p.W * p.Cin;

  // ---- loader state (branch-free: out-of-image pixels read a clamped address and are zeroed by a select)
  constexpr int NPL = INB ? NPH : NPF;  // 16-B pieces per thread per chunk
  int pg[NPL];              // element offset of the piece inside the image (clamped), without the chunk offset
  bool pk[NPL];             // inside the image
  constexpr int ppp = INB ? 4 : 8;   // pieces per pixel
  constexpr int cpp = INB ? 8 : 4;   // channels per piece
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int f = t + 256 * i;
    const int pp = f / ppp, q = f - pp * ppp;
    const int py = pp / PC, px = pp - py * PC;
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    const bool inpatch = pp < NPIX;
    pk[i] = inpatch && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    const int cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);
    pg[i] = (cy * p.W + cx) * p.Cin + q * cpp;
  }
  int wg[NWF];
  bool wk[NWF];
#pragma unroll
  for (int i = 0; i < NWF; ++i) {
    const int u = t + 256 * i;
    const int row = u >> 2, part = u & 3;
    const int tap = row / BN, n = row - tap * BN;
    wk[i] = row < 3 * BN && n0 + n < p.Cout;
    wg[i] = min(n0 + n, p.Cout - 1) * p.ldw + tap * p.Cin + part * 8;
  }

  f32x4 rp[NPL];   // fp32 path: 4 channels; bf16 path: 8 packed channels (bit pattern)
  u32x4 rw[NWF];
  auto load_patch = [&](int c0) {
    if (INB) {
#pragma unroll
      for (int i = 0; i < NPL; ++i)
        rp[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned short*>(p.in) + in_img + pg[i] + c0);
    } else {
#pragma unroll
      for (int i = 0; i < NPL; ++i) rp[i] = *reinterpret_cast<const f32x4*>(p.in + in_img + pg[i] + c0);
    }
  };
  // InstanceNorm statistics: staged in LDS once (they would otherwise occupy 16 registers through the MFMA loop)
  if (p.in_stats) {
    for (int i = t; i < 2 * p.Cin; i += 256) Sst[i] = p.in_stats[img * 2 * p.Cin + i];
    __syncthreads();
  }
  auto norm4 = [&](f32x4 v, const f32x4& m, const f32x4& rs) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf((v[e] - m[e]) * rs[e], 0.f);
    return v;
  };
  auto store_patch = [&](int c0) {
    f32x4 sm0 = {0.f, 0.f, 0.f, 0.f}, sr0 = sm0, sm1 = sm0, sr1 = sm0;
    if (p.in_stats) {
      const float* sp = &Sst[(c0 + (t % ppp) * cpp) * 2];
      const f32x4 a = *reinterpret_cast<const f32x4*>(sp), b4 = *reinterpret_cast<const f32x4*>(sp + 4);
      sm0 = (f32x4){a[0], a[2], b4[0], b4[2]};
      sr0 = (f32x4){a[1], a[3], b4[1], b4[3]};
      if (INB) {
        const f32x4 c4 = *reinterpret_cast<const f32x4*>(sp + 8), d4 = *reinterpret_cast<const f32x4*>(sp + 12);
        sm1 = (f32x4){c4[0], c4[2], d4[0], d4[2]};
        sr1 = (f32x4){c4[1], c4[3], d4[1], d4[3]};
      }
    }
    if (INB) {
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int f = t + 256 * i, pp = f / ppp;
        if (pp >= NPIX) continue;
        u32x4 w = __builtin_bit_cast(u32x4, rp[i]);
        if (p.in_stats) {
          f32x4 lo = (f32x4){__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xFFFF0000u), __uint_as_float(w[1] << 16),
                             __uint_as_float(w[1] & 0xFFFF0000u)};
          f32x4 hi = (f32x4){__uint_as_float(w[2] << 16), __uint_as_float(w[2] & 0xFFFF0000u), __uint_as_float(w[3] << 16),
                             __uint_as_float(w[3] & 0xFFFF0000u)};
          const u32x2 a = __builtin_bit_cast(u32x2, __builtin_convertvector(norm4(lo, sm0, sr0), bf16x4));
          const u32x2 c = __builtin_bit_cast(u32x2, __builtin_convertvector(norm4(hi, sm1, sr1), bf16x4));
          w = (u32x4){a[0], a[1], c[0], c[1]};
        }
        if (!pk[i]) w = (u32x4){0u, 0u, 0u, 0u};  // zero padding applies after the normalisation
        *reinterpret_cast<u32x4*>(&Ps[pp * LDP + (f - pp * ppp) * cpp]) = w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int f = t + 256 * i, pp = f / ppp;
        if (pp >= NPIX) continue;
        f32x4 v = rp[i];
        if (p.in_stats) v = norm4(v, sm0, sr0);
        u32x2 w = __builtin_bit_cast(u32x2, __builtin_convertvector(v, bf16x4));
        if (!pk[i]) w = (u32x2){0u, 0u};
        *reinterpret_cast<u32x2*>(&Ps[pp * LDP + (f - pp * ppp) * cpp]) = w;
      }
    }
  };
  auto load_w = [&](int c0, int kh) {
    const int base = kh * 3 * p.Cin + c0;
#pragma unroll
    for (int i = 0; i < NWF; ++i) rw[i] = *reinterpret_cast<const u32x4*>(p.w + wg[i] + base);
  };
  auto store_w = [&]() {
#pragma unroll
    for (int i = 0; i < NWF; ++i) {
      const int u = t + 256 * i, row = u >> 2;
      if (row >= 3 * BN) continue;
      *reinterpret_cast<u32x4*>(&Ws[row * LDP + (u & 3) * 8]) = wk[i] ? rw[i] : (u32x4){0u, 0u, 0u, 0u};
    }
  };

  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  // lane bases; everything else in the inner loop is a compile-time offset
  const unsigned short* Pl = Ps + (wm * TM * PC + r) * LDP + h * 8;
  const unsigned short* Wl = Ws + (wn * TN * 32 + r) * LDP + h * 8;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nchunk = p.Cin / CK;
  load_patch(0);
  load_w(0, 0);
  for (int c = 0; c < nchunk; ++c) {
    store_patch(c * CK);  // (the barrier that ended the previous chunk's last stage made the patch free)
    const bool more = c + 1 < nchunk;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      store_w();
      __syncthreads();
      if (kh < 2) {
        load_w(c * CK, kh + 1);
      } else if (more) {
        load_w((c + 1) * CK, 0);
        load_patch((c + 1) * CK);
      }
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
        for (int ks = 0; ks < CK / 16; ++ks) {
          bf16x8 a[TM], bb[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i)
            a[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Pl + ((i + kh) * PC + kw) * LDP + ks * 16));
#pragma unroll
          for (int j = 0; j < TN; ++j)
            bb[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Wl + (kw * BN + j * 32) * LDP + ks * 16));
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
      }
      __syncthreads();
    }
  }

  // ---- epilogue: D[pixel][cout]: cout on the lanes (coalesced 128-B rows), pixels (e&3) + 8(e>>2) + 4h in the registers.
  // A row's address is a wave-uniform base (SGPRs) plus one 32-bit lane offset; the sixteen pixels of a register block
  // differ by compile-time multiples of ldo.
  const bool interior = x0 + TC <= p.W;
  const int slots_x = tiles_x;
  const int esz = p.out_bf16 ? 2 : 4;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + r;
    const bool nok = n < p.Cout;
    const float bv = (p.bias && nok) ? p.bias[n] : 0.f;
    const int lane_off = (4 * h * p.ldo + n) * esz;  // bytes
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int y = y0 + wm * TM + i;  // wave-uniform
      if (y >= p.H) continue;
      char* rowp = reinterpret_cast<char*>(p.out) + (((img * p.H + y) * (long long)p.W + x0) * p.ldo) * esz;  // wave-uniform
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int dx = (e & 3) + 8 * (e >> 2);
        if (nok && (interior || x0 + 4 * h + dx < p.W)) {
          const float v = acc[i][j][e] + bv;
          char* q = rowp + dx * p.ldo * esz + lane_off;
          if (p.out_bf16) *reinterpret_cast<unsigned short*>(q) = mvt_bf16_bits(v);
          else *reinterpret_cast<float*>(q) = v;
          s1 += v;
          s2 = fmaf(v, v, s2);
        }
      }
      if (p.out_part) {  // one writer per (row segment, channel): deterministic
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (h == 0 && nok) {
          float* pp = p.out_part + ((img * p.slots + (long long)y * slots_x + tx) * p.Cout + n) * 2;
          pp[0] = s1;
          pp[1] = s2;
        }
      }
    }
  }
}

}  // namespace

// slots of the fused statistics: one per 32-pixel row segment
__attribute__((visibility("hidden"))) int mvt_detail_conv_rows_slots(int H, int W) { return H * (int)mvt_cdiv(W, TC); }

__attribute__((visibility("hidden"))) int mvt_detail_conv3x3_rows(const void* in, const unsigned short* w, int ldw, const float* bias,
                                                                  void* out, int n, int H, int W, int Cin, int Cout, int ldo, int act,
                                                                  int io_flags, const float* in_stats, float* out_partial,
                                                                  hipStream_t stream) {
  MVT_REQUIRE((!in_stats || Cin <= MAXC) && Cin % CK == 0 && (long long)H * W * Cin < (1LL << 31) && (long long)Cout * ldw < (1LL << 31) && act == MVT_ACT_NONE);
  MVT_REQUIRE((long long)(TC + 8) * ldo * 4 < (1LL << 31));
  RowsArgs a{};
  a.in = (const float*)in; a.w = w; a.bias = bias; a.out = (float*)out; a.in_stats = in_stats; a.out_part = out_partial;
  a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ldw = ldw; a.ldo = ldo;
  a.slots = mvt_detail_conv_rows_slots(H, W);
  a.in_bf16 = io_flags & MVT_IO_IN_BF16 ? 1 : 0;
  a.out_bf16 = io_flags & MVT_IO_OUT_BF16 ? 1 : 0;
  const long long tiles = (long long)n * mvt_cdiv(H, TR) * mvt_cdiv(W, TC);
  MVT_REQUIRE(tiles * mvt_cdiv(Cout, 64) < (1LL << 31));
#define LAUNCH(TM_, TN_, WM_, WN_, BN_)                                                                                       \
  do {                                                                                                                       \
    const dim3 grid((unsigned)(tiles * mvt_cdiv(Cout, BN_)));                                                                \
    if (a.in_bf16) hipLaunchKernelGGL((conv3x3_rows_bf16<TM_, TN_, WM_, WN_, true>), grid, dim3(256), 0, stream, a);         \
    else hipLaunchKernelGGL((conv3x3_rows_bf16<TM_, TN_, WM_, WN_, false>), grid, dim3(256), 0, stream, a);                  \
  } while (0)
  static const int force = getenv("MVT_ROWS_CFG") ? atoi(getenv("MVT_ROWS_CFG")) : 0;  // tuning override
  if (force == 1) LAUNCH(2, 2, 4, 1, 64);
  else if (force == 2) LAUNCH(4, 2, 2, 2, 128);
  else if (force == 3) LAUNCH(2, 4, 4, 1, 128);
  else if (Cout % 64 != 0 && Cout % 96 == 0) LAUNCH(2, 3, 4, 1, 96);
  else LAUNCH(2, 2, 4, 1, 64);  // three workgroups per CU with bf16 tensors: measured 1.4x faster than the 128-channel tiles at one per CU
#undef LAUNCH
  return mvt_launch_status();
}
