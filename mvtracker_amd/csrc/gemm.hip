// fp32 GEMM / implicit-GEMM convolution on the CDNA4 matrix cores.
//
// One kernel body serves every nn.Linear of the updater and every nn.Conv2d of the encoder:
//   C[m][n] = R[m][n] + act( sum_k A(m,k) * Wt[n][k] + bias[n] )
// A(m,k) comes from a pluggable row loader: dense rows (mode 0), an NHWC im2col gather with
// Cin % 32 == 0 (mode 1: one 32-wide k-tile lies inside a single filter tap, so a tile row is 128
// contiguous bytes of the input) or the 7x7 stem with Cin padded to 4 (mode 2: one k-tile per
// filter row, 7 pixels x 4 channels contiguous).
//
// Matrix-core mapping: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 FLOP/clk/SIMD).  A wave
// owns TMx TN blocks of 32x32; lanes 0-31 feed k, lanes 32-63 feed k+4 of each 8-wide k-group so
// that one ds_read_b128 per operand block serves four consecutive MFMAs (the k permutation is
// the same for A and B, so the sum is unchanged).  LDS tiles are [rows][32+4] floats: the 144-byte
// row stride makes the 16-lane ds_read_b128 groups conflict-free and keeps ds_write_b128 aligned.
// Global loads for k-tile t+1 are issued into registers before the MFMAs of tile t.
#include "common.h"

namespace {

struct GemmArgs {
  const float* A;
  const float* W;
  const float* bias;
  const float* R;
  float* C;
  int M, N, K;
  int lda, ldw, ldc, ldr;
  int act;
  int mode;  // 0 dense, 1 conv (Cin % 32 == 0), 2 stem (Cin == 4)
  int H, Wd, Cin, Ho, Wo, KH, KW, stride, pad;
  int nk;  // number of k tiles (32 wide for the fp32 kernel, 64 wide for the bf16 kernels)
  const unsigned short* Whi;  // bf16 kernels: weights pre-split into bf16 hi (+ lo for the split-precision mode)
  const unsigned short* Wlo;
  // optional LayerNorm fused into the A loader of the dense bf16 kernels (K <= 1024): a = (a - mean) * rstd [* w + b]
  // InstanceNorm fusion (bf16 conv kernels): in_stats [n][Cin][2] (mean, rstd) -> the loader applies relu((x-mean)*rstd)
  // (3x3 halo kernel only); out_part [n][slots][N][2] receives per-32-row-block sums / sums of squares of the outputs
  const float* in_stats;
  float* out_part;
  int slots;
  int a_bf16, c_bf16;  // bf16 conv kernels: activations read / written as bf16 instead of fp32 (conv modes only)
  int ln;
  float ln_eps;
  const float* ln_w;
  const float* ln_b;
};

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;

template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_mfma_f32(GemmArgs p) {
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int AF = BM / 32;  // float4 per thread per k-tile (A)
  constexpr int BF = BN / 32;  // float4 per thread per k-tile (W)
  static_assert(WM * WN == 4, "four waves per workgroup");
  __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LDS_LD];
  float* As = lds;
  float* Bs = lds + BM * LDS_LD;

  const int t = threadIdx.x;
  const int c4 = t & 7;
  const int rbase = t >> 3;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tile_m = blockIdx.x / tiles_n;
  const int tile_n = blockIdx.x % tiles_n;
  const long long m0 = (long long)tile_m * BM;
  const int n0 = tile_n * BN;

  // ---- per-row loader state (rows are fixed over the k loop)
  const float* a_ptr[AF];
  unsigned a_hw[AF];  // conv: (ih0 + 32000) << 16 | (iw0 + 64); dense: 1 = row valid, 0 = not
#pragma unroll
  for (int i = 0; i < AF; ++i) {
    long long m = m0 + rbase + 32 * i;
    bool ok = m < p.M;
    if (p.mode == 0) {
      a_ptr[i] = p.A + (ok ? m : 0) * (long long)p.lda + c4 * 4;
      a_hw[i] = ok ? 1 : 0;
    } else {
      long long hw = (long long)p.Ho * p.Wo;
      long long img = ok ? m / hw : 0;
      int rem = ok ? (int)(m - img * hw) : 0;
      int oh = rem / p.Wo, ow = rem - oh * p.Wo;
      int ih0 = ok ? oh * p.stride - p.pad : -20000;
      int iw0 = ow * p.stride - p.pad;
      a_ptr[i] = p.A + img * (long long)p.H * p.Wd * p.Cin;
      a_hw[i] = ((unsigned)(ih0 + 32000) << 16) | (unsigned)(iw0 + 64);
    }
  }
  const float* w_ptr[BF];
  bool w_ok[BF];
#pragma unroll
  for (int i = 0; i < BF; ++i) {
    int n = n0 + rbase + 32 * i;
    w_ok[i] = n < p.N;
    w_ptr[i] = p.W + (long long)(w_ok[i] ? n : 0) * p.ldw + c4 * 4;
  }
  const int kp4 = (p.K + 3) & ~3;

  f32x4 ra[AF], rb[BF];
  auto load_tile = [&](int kt) {
    if (p.mode == 0) {
      const int k = kt * BK + c4 * 4;
      const bool kok = k < kp4;
#pragma unroll
      for (int i = 0; i < AF; ++i) {
        ra[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (kok && a_hw[i]) ra[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + kt * BK);
      }
    } else if (p.mode == 1) {
      const int k0 = kt * BK;
      const int tap = k0 / p.Cin;
      const int c0 = k0 - tap * p.Cin;
      const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
      for (int i = 0; i < AF; ++i) {
        int ih = (int)(a_hw[i] >> 16) - 32000 + kh;
        int iw = (int)(a_hw[i] & 0xFFFFu) - 64 + kw;
        ra[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.Wd)
          ra[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + ((long long)ih * p.Wd + iw) * p.Cin + c0 + c4 * 4);
      }
    } else {
      const int kh = kt;
#pragma unroll
      for (int i = 0; i < AF; ++i) {
        int ih = (int)(a_hw[i] >> 16) - 32000 + kh;
        int iw = (int)(a_hw[i] & 0xFFFFu) - 64 + c4;
        ra[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c4 < p.KW && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.Wd)
          ra[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + ((long long)ih * p.Wd + iw) * 4);
      }
    }
#pragma unroll
    for (int i = 0; i < BF; ++i) {
      rb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (w_ok[i]) rb[i] = *reinterpret_cast<const f32x4*>(w_ptr[i] + kt * BK);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < AF; ++i) *reinterpret_cast<f32x4*>(&As[(rbase + 32 * i) * LDS_LD + c4 * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < BF; ++i) *reinterpret_cast<f32x4*>(&Bs[(rbase + 32 * i) * LDS_LD + c4 * 4]) = rb[i];
  };

  const int wave = t >> 6, lane = t & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  load_tile(0);
  store_tile();
  __syncthreads();
  for (int kt = 0; kt < p.nk; ++kt) {
    const bool more = kt + 1 < p.nk;
    if (more) load_tile(kt + 1);
#pragma unroll
    for (int k8 = 0; k8 < BK / 8; ++k8) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = *reinterpret_cast<const f32x4*>(&As[((wm * TM + i) * 32 + r) * LDS_LD + k8 * 8 + h * 4]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[j] = *reinterpret_cast<const f32x4*>(&Bs[((wn * TN + j) * 32 + r) * LDS_LD + k8 * 8 + h * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (more) {
      store_tile();
      __syncthreads();
    }
  }

  // ---- epilogue: lane holds column n0 + .. + r, rows (reg&3) + 8*(reg>>2) + 4*h of each 32x32 block
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + r;
    if (n >= p.N) continue;
    const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        long long m = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m < p.M) {
          float v = mvt_act(acc[i][j][e] + bv, p.act);
          if (p.R) v += p.R[m * p.ldr + n];
          p.C[m * p.ldc + n] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// bf16 matrix-core path.  v_mfma_f32_32x32x16_bf16 runs 16x the fp32 MFMA rate.  Two modes:
//   SPLIT = false : operands rounded to bf16, fp32 accumulate (what torch autocast does to the
//                   reference's convs / linears).
//   SPLIT = true  : "bf16x3": every fp32 operand x is split as hi = bf16(x), lo = bf16(x - hi) and the
//                   product is accumulated as hi*hi + hi*lo + lo*hi in fp32 (the dropped lo*lo term is
//                   2^-16 relative): fp32-grade results at 3 bf16 MFMAs instead of 8 fp32 MFMAs per k=16.
// Activations are converted while they are staged into LDS; weights are pre-split once on the host side
// (mvt_split_bf16).  k-tile = 64; LDS rows are 64+8 bf16 (144 B: conflict-free ds_read_b128, 8-byte aligned
// ds_write_b64); lane (r, h) reads the 8 consecutive k of its half for each 16-wide MFMA step.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BKB = 64;
constexpr int LDB = BKB + 8;  // bf16 elements per LDS row

__device__ __forceinline__ void split4(const f32x4& v, uint2& hi, uint2& lo) {
  bf16x4 h = __builtin_convertvector(v, bf16x4);
  f32x4 hf = __builtin_convertvector(h, f32x4);
  bf16x4 l = __builtin_convertvector(v - hf, bf16x4);
  hi = __builtin_bit_cast(uint2, h);
  lo = __builtin_bit_cast(uint2, l);
}

template <int TM, int TN, int WM, int WN, bool SPLIT>
__global__ __launch_bounds__(256) void gemm_mfma_bf16(GemmArgs p) {
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int AF = BM / 16;  // float4 (A) / 4xbf16 (W) per thread per k-tile
  constexpr int BF = BN / 16;
  static_assert(WM * WN == 4, "four waves per workgroup");
  __shared__ __attribute__((aligned(16))) unsigned short lds[(BM + BN) * LDB * (SPLIT ? 2 : 1)];
  unsigned short* Ah = lds;
  unsigned short* Bh = Ah + BM * LDB;
  unsigned short* Al = Bh + BN * LDB;
  unsigned short* Bl = Al + (SPLIT ? BM * LDB : 0);

  const int t = threadIdx.x;
  const int c4 = t & 15;
  const int rbase = t >> 4;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tile_m = blockIdx.x / tiles_n;
  const int tile_n = blockIdx.x % tiles_n;
  const long long m0 = (long long)tile_m * BM;
  const int n0 = tile_n * BN;

  const float* a_ptr[AF];
  long long a_img[AF];  // conv modes: element offset of the row's image (the tensor may be bf16)
  unsigned a_hw[AF];
#pragma unroll
  for (int i = 0; i < AF; ++i) {
    long long m = m0 + rbase + 16 * i;
    bool ok = m < p.M;
    a_img[i] = 0;
    if (p.mode == 0) {
      a_ptr[i] = p.A + (ok ? m : 0) * (long long)p.lda + c4 * 4;
      a_hw[i] = ok ? 1 : 0;
    } else {
      long long hw = (long long)p.Ho * p.Wo;
      long long img = ok ? m / hw : 0;
      int rem = ok ? (int)(m - img * hw) : 0;
      int oh = rem / p.Wo, ow = rem - oh * p.Wo;
      int ih0 = ok ? oh * p.stride - p.pad : -20000;
      int iw0 = ow * p.stride - p.pad;
      a_img[i] = img * (long long)p.H * p.Wd * p.Cin;
      a_ptr[i] = p.A + a_img[i];
      a_hw[i] = ((unsigned)(ih0 + 32000) << 16) | (unsigned)(iw0 + 64);
    }
  }
  long long w_off[BF];
  bool w_ok[BF];
#pragma unroll
  for (int i = 0; i < BF; ++i) {
    int n = n0 + rbase + 16 * i;
    w_ok[i] = n < p.N;
    w_off[i] = (long long)(w_ok[i] ? n : 0) * p.ldw + c4 * 4;
  }
  const int kp4 = (p.K + 3) & ~3;

  // fused LayerNorm: per-row mean / rstd over the K columns.  A row of a k-tile is spread over the 16 threads that
  // share rbase (lane bits 0-3), so one xor-shuffle tree per row finishes the sums; the tile is read again (L2) below.
  float ln_mean[AF], ln_rstd[AF];
  if (p.ln) {
    float s1[AF], s2[AF];
#pragma unroll
    for (int i = 0; i < AF; ++i) s1[i] = s2[i] = 0.f;
    for (int kt = 0; kt < p.nk; ++kt) {
      const bool kok = kt * BKB + c4 * 4 < kp4;
#pragma unroll
      for (int i = 0; i < AF; ++i) {
        if (kok && a_hw[i]) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(a_ptr[i] + kt * BKB);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool in = kt * BKB + c4 * 4 + e < p.K;
            s1[i] += in ? v[e] : 0.f;
            s2[i] += in ? v[e] * v[e] : 0.f;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < AF; ++i) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        s1[i] += __shfl_xor(s1[i], o, 64);
        s2[i] += __shfl_xor(s2[i], o, 64);
      }
      ln_mean[i] = s1[i] / (float)p.K;
      const float var = fmaxf(s2[i] / (float)p.K - ln_mean[i] * ln_mean[i], 0.f);
      ln_rstd[i] = 1.0f / sqrtf(var + p.ln_eps);
    }
  }

  f32x4 ra[AF];
  uint2 rbh[BF], rbl[BF];
  auto load_tile = [&](int kt) {
    const int k = kt * BKB + c4 * 4;
    if (p.mode == 0) {
      const bool kok = k < kp4;
#pragma unroll
      for (int i = 0; i < AF; ++i) {
        ra[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (kok && a_hw[i]) ra[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + kt * BKB);
      }
      if (p.ln && kok) {
        f32x4 w4 = (f32x4){1.f, 1.f, 1.f, 1.f}, b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (p.ln_w) {
          w4 = *reinterpret_cast<const f32x4*>(p.ln_w + k);
          b4 = *reinterpret_cast<const f32x4*>(p.ln_b + k);
        }
#pragma unroll
        for (int i = 0; i < AF; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) ra[i][e] = (k + e < p.K && a_hw[i]) ? (ra[i][e] - ln_mean[i]) * ln_rstd[i] * w4[e] + b4[e] : 0.f;
      }
    } else if (p.mode == 1) {
      const int tap = k / p.Cin;
      const int c = k - tap * p.Cin;
      const int kh = tap / p.KW, kw = tap - kh * p.KW;
      const bool kok = k < p.K;
#pragma unroll
      for (int i = 0; i < AF; ++i) {
        int ih = (int)(a_hw[i] >> 16) - 32000 + kh;
        int iw = (int)(a_hw[i] & 0xFFFFu) - 64 + kw;
        ra[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (kok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.Wd)
          ra[i] = load_act4(p.A, a_img[i] + ((long long)ih * p.Wd + iw) * p.Cin + c, p.a_bf16);
      }
    } else {
      const int kh = k >> 5, kw = (k & 31) >> 2;
      const bool kok = kh < p.KH && kw < p.KW;
#pragma unroll
      for (int i = 0; i < AF; ++i) {
        int ih = (int)(a_hw[i] >> 16) - 32000 + kh;
        int iw = (int)(a_hw[i] & 0xFFFFu) - 64 + kw;
        ra[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (kok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.Wd)
          ra[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + ((long long)ih * p.Wd + iw) * 4);
      }
    }
#pragma unroll
    for (int i = 0; i < BF; ++i) {
      rbh[i] = make_uint2(0u, 0u);
      rbl[i] = make_uint2(0u, 0u);
      if (w_ok[i]) {
        rbh[i] = *reinterpret_cast<const uint2*>(p.Whi + w_off[i] + kt * BKB);
        if (SPLIT) rbl[i] = *reinterpret_cast<const uint2*>(p.Wlo + w_off[i] + kt * BKB);
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < AF; ++i) {
      uint2 hi, lo;
      split4(ra[i], hi, lo);
      const int o = (rbase + 16 * i) * LDB + c4 * 4;
      *reinterpret_cast<uint2*>(&Ah[o]) = hi;
      if (SPLIT) *reinterpret_cast<uint2*>(&Al[o]) = lo;
    }
#pragma unroll
    for (int i = 0; i < BF; ++i) {
      const int o = (rbase + 16 * i) * LDB + c4 * 4;
      *reinterpret_cast<uint2*>(&Bh[o]) = rbh[i];
      if (SPLIT) *reinterpret_cast<uint2*>(&Bl[o]) = rbl[i];
    }
  };

  const int wave = t >> 6, lane = t & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  load_tile(0);
  store_tile();
  __syncthreads();
  for (int kt = 0; kt < p.nk; ++kt) {
    const bool more = kt + 1 < p.nk;
    if (more) load_tile(kt + 1);
#pragma unroll
    for (int ks = 0; ks < BKB / 16; ++ks) {
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int o = ((wm * TM + i) * 32 + r) * LDB + ks * 16 + h * 8;
        ah[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&Ah[o]));
        if (SPLIT) al[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&Al[o]));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int o = ((wn * TN + j) * 32 + r) * LDB + ks * 16 + h * 8;
        bh[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&Bh[o]));
        if (SPLIT) bl[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&Bl[o]));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (SPLIT) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (more) {
      store_tile();
      __syncthreads();
    }
  }

#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + r;
    const bool nok = n < p.N;
    const float bv = (p.bias && nok) ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        long long m = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (nok && m < p.M) {
          float v = mvt_act(acc[i][j][e] + bv, p.act);
          if (p.R) v += p.R[m * p.ldr + n];
          store_act(p.C, m * p.ldc + n, v, p.c_bf16);
          s1 += v;
          s2 = fmaf(v, v, s2);
        }
      }
      if (p.out_part) {  // conv mode, Ho*Wo % 32 == 0: this 32-row block lies inside one image
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        const long long mrow = m0 + (wm * TM + i) * 32;
        const long long hw = (long long)p.Ho * p.Wo;
        if (h == 0 && nok && mrow < p.M) {
          const long long im = mrow / hw, slot = (mrow - im * hw) / 32;
          float* pp = p.out_part + ((im * p.slots + slot) * p.N + n) * 2;
          pp[0] = s1;
          pp[1] = s2;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with an LDS-resident input patch ("halo tile").
// The im2col loader above re-reads every input pixel 9 times (once per filter tap); at fp32 activations that
// makes the big encoder convs L2/HBM-bandwidth bound.  Here a workgroup owns an 8x16 block of output pixels:
// per 32-channel chunk it stages the 10x18 input patch ONCE (converted to bf16 on the way) and the nine taps
// read it back at shifted offsets, so global traffic per MAC drops ~6x and the kernel becomes MFMA-bound.
// Weights ([Cout][kh][kw][Cin] bf16, row stride ldw) are staged one filter row (3 taps) at a time.
// Patch / weight rows are 32+8 bf16 (80 B): conflict-free ds_read_b128, 16-B aligned.
constexpr int HT = 8, HW_ = 16;                 // output tile
constexpr int PH = HT + 2, PW = HW_ + 2;        // patch
constexpr int CK = 32, LDC = CK + 8;            // channel chunk, LDS row stride (bf16 elements)

template <int TM, int TN, int WM, int WN, bool SPLIT>
__global__ __launch_bounds__(256) void conv3x3_halo_bf16(GemmArgs p) {
  constexpr int BN = WN * TN * 32;
  static_assert(WM * TM == 4 && WM * WN == 4, "128 output pixels, four waves");
  constexpr int NP = PH * PW;                         // 180 patch pixels
  constexpr int NPF = (NP * 8 + 255) / 256;            // float4 per thread per chunk
  constexpr int NWF = (3 * BN * 4 + 255) / 256;        // 16-B weight pieces per thread per stage (3 taps)
  __shared__ __attribute__((aligned(16))) unsigned short lds[(NP + 3 * BN) * LDC * (SPLIT ? 2 : 1)];
  unsigned short* Ph = lds;
  unsigned short* Wh = Ph + NP * LDC;
  unsigned short* Pl = Wh + 3 * BN * LDC;
  unsigned short* Wl = Pl + (SPLIT ? NP * LDC : 0);

  const int t = threadIdx.x;
  const int tiles_x = (p.Wo + HW_ - 1) / HW_, tiles_y = (p.Ho + HT - 1) / HT;
  const int tiles_n = (p.N + BN - 1) / BN;
  int b = blockIdx.x;
  const int tn = b % tiles_n; b /= tiles_n;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y;
  const long long img = b / tiles_y;
  const int y0 = ty * HT, x0 = tx * HW_, n0 = tn * BN;
  const long long in_off = img * (long long)p.H * p.Wd * p.Cin;  // element offset (fp32 or bf16 input)

  // patch loader state: pixel / quad of each of this thread's float4
  long long poff[NPF];
  bool pok[NPF];
  int plds[NPF];
#pragma unroll
  for (int i = 0; i < NPF; ++i) {
    const int f = t + 256 * i;
    const int pp = f >> 3, q = f & 7;
    const int py = pp / PW, px = pp - py * PW;
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    pok[i] = pp < NP && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.Wd;
    poff[i] = ((long long)gy * p.Wd + gx) * p.Cin + q * 4;
    plds[i] = pp < NP ? pp * LDC + q * 4 : -1;
  }
  // weight loader state
  long long woff[NWF];
  int wlds[NWF];
#pragma unroll
  for (int i = 0; i < NWF; ++i) {
    const int u = t + 256 * i;
    const int row = u >> 2, part = u & 3;
    const int tl = row / BN, n = row - tl * BN;
    const bool ok = row < 3 * BN && n0 + n < p.N;
    woff[i] = ok ? (long long)(n0 + n) * p.ldw + (long long)tl * p.Cin + part * 8 : -1;
    wlds[i] = row < 3 * BN ? row * LDC + part * 8 : -1;
  }

  f32x4 rp[NPF];
  uint4 rwh[NWF], rwl[NWF];
  auto load_patch = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      rp[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (pok[i]) rp[i] = load_act4(p.A, in_off + poff[i] + c0, p.a_bf16);
    }
  };
  f32x4 st_m = (f32x4){0.f, 0.f, 0.f, 0.f}, st_r = (f32x4){1.f, 1.f, 1.f, 1.f};  // stats of this thread's channel quad
  auto load_stats = [&](int c0) {
    if (p.in_stats) {
      const float* sp = p.in_stats + (img * p.Cin + c0 + (t & 7) * 4) * 2;
      const f32x4 a = *reinterpret_cast<const f32x4*>(sp), b = *reinterpret_cast<const f32x4*>(sp + 4);
      st_m = (f32x4){a[0], a[2], b[0], b[2]};
      st_r = (f32x4){a[1], a[3], b[1], b[3]};
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      if (plds[i] >= 0) {
        if (p.in_stats && pok[i]) {  // zero padding applies AFTER the normalisation: out-of-image taps stay 0
#pragma unroll
          for (int e = 0; e < 4; ++e) rp[i][e] = fmaxf((rp[i][e] - st_m[e]) * st_r[e], 0.f);
        }
        uint2 hi, lo;
        split4(rp[i], hi, lo);
        *reinterpret_cast<uint2*>(&Ph[plds[i]]) = hi;
        if (SPLIT) *reinterpret_cast<uint2*>(&Pl[plds[i]]) = lo;
      }
    }
  };
  auto load_w = [&](int c0, int kh) {
    const long long base = (long long)kh * 3 * p.Cin + c0;
#pragma unroll
    for (int i = 0; i < NWF; ++i) {
      rwh[i] = make_uint4(0u, 0u, 0u, 0u);
      rwl[i] = make_uint4(0u, 0u, 0u, 0u);
      if (woff[i] >= 0) {
        rwh[i] = *reinterpret_cast<const uint4*>(p.Whi + woff[i] + base);
        if (SPLIT) rwl[i] = *reinterpret_cast<const uint4*>(p.Wlo + woff[i] + base);
      }
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int i = 0; i < NWF; ++i) {
      if (wlds[i] >= 0) {
        *reinterpret_cast<uint4*>(&Wh[wlds[i]]) = rwh[i];
        if (SPLIT) *reinterpret_cast<uint4*>(&Wl[wlds[i]]) = rwl[i];
      }
    }
  };

  const int wave = t >> 6, lane = t & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  int arow[TM];  // patch index of this lane's output pixel (tap 0,0) per 32-row block
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = (wm * TM + i) * 32 + r;
    arow[i] = (m / HW_) * PW + (m % HW_);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nchunk = p.Cin / CK;
  load_patch(0);
  load_stats(0);
  load_w(0, 0);
  for (int c = 0; c < nchunk; ++c) {
    store_patch();
    if (c + 1 < nchunk) {
      load_patch((c + 1) * CK);
      load_stats((c + 1) * CK);
    }
#pragma unroll 1
    for (int kh = 0; kh < 3; ++kh) {
      store_w();
      __syncthreads();
      if (kh < 2) load_w(c * CK, kh + 1);
      else if (c + 1 < nchunk) load_w((c + 1) * CK, 0);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
        for (int ks = 0; ks < CK / 16; ++ks) {
          bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int o = (arow[i] + kh * PW + kw) * LDC + ks * 16 + h * 8;
            ah[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&Ph[o]));
            if (SPLIT) al[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&Pl[o]));
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int o = (kw * BN + (wn * TN + j) * 32 + r) * LDC + ks * 16 + h * 8;
            bh[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&Wh[o]));
            if (SPLIT) bl[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&Wl[o]));
          }
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              if (SPLIT) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
              }
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            }
        }
      }
      __syncthreads();
    }
  }

  const long long out_off = img * (long long)p.Ho * p.Wo * p.ldc;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + r;
    const bool nok = n < p.N;
    const float bv = (p.bias && nok) ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int y = y0 + m / HW_, x = x0 + m % HW_;
        if (nok && y < p.Ho && x < p.Wo) {
          const float v = mvt_act(acc[i][j][e] + bv, p.act);
          store_act(p.C, out_off + ((long long)y * p.Wo + x) * p.ldc + n, v, p.c_bf16);
          s1 += v;
          s2 = fmaf(v, v, s2);
        }
      }
      if (p.out_part) {  // per-channel sums of this 32-pixel block (deterministic: one writer per slot)
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        const long long slot = ((long long)ty * tiles_x + tx) * 4 + wm * TM + i;
        if (h == 0 && nok) {
          float* pp = p.out_part + ((img * p.slots + slot) * p.N + n) * 2;
          pp[0] = s1;
          pp[1] = s2;
        }
      }
    }
  }
}

template <bool SPLIT>
int launch_conv3x3_halo(const GemmArgs& a, int nimg, hipStream_t s) {
  const long long tiles = (long long)nimg * mvt_cdiv(a.Ho, HT) * mvt_cdiv(a.Wo, HW_);
  if (a.N % 128 != 0 && a.N % 96 == 0) {
    hipLaunchKernelGGL((conv3x3_halo_bf16<1, 3, 4, 1, SPLIT>), dim3((unsigned)(tiles * mvt_cdiv(a.N, 96))), dim3(256), 0, s, a);
  } else if (a.N <= 64) {
    hipLaunchKernelGGL((conv3x3_halo_bf16<1, 2, 4, 1, SPLIT>), dim3((unsigned)(tiles * mvt_cdiv(a.N, 64))), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((conv3x3_halo_bf16<2, 2, 2, 2, SPLIT>), dim3((unsigned)(tiles * mvt_cdiv(a.N, 128))), dim3(256), 0, s, a);
  }
  return mvt_launch_status();
}

inline int pick_tile(const GemmArgs& a);

template <bool SPLIT>
int launch_gemm_bf16(const GemmArgs& a, hipStream_t s) {
  auto blocks = [&](int bm, int bn) { return (unsigned)(mvt_cdiv(a.M, bm) * mvt_cdiv(a.N, bn)); };
  switch (pick_tile(a)) {
    case 1: hipLaunchKernelGGL((gemm_mfma_bf16<1, 3, 4, 1, SPLIT>), dim3(blocks(128, 96)), dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL((gemm_mfma_bf16<2, 2, 4, 1, SPLIT>), dim3(blocks(256, 64)), dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL((gemm_mfma_bf16<1, 2, 2, 2, SPLIT>), dim3(blocks(64, 128)), dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL((gemm_mfma_bf16<1, 1, 2, 2, SPLIT>), dim3(blocks(64, 64)), dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((gemm_mfma_bf16<2, 2, 2, 2, SPLIT>), dim3(blocks(128, 128)), dim3(256), 0, s, a); break;
  }
  return mvt_launch_status();
}

__global__ void split_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ hi, unsigned short* __restrict__ lo,
                                  long long n4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    uint2 h, l;
    split4(*reinterpret_cast<const f32x4*>(src + i * 4), h, l);
    *reinterpret_cast<uint2*>(hi + i * 4) = h;
    if (lo) *reinterpret_cast<uint2*>(lo + i * 4) = l;
  }
}

// Tile choice: the largest tile that still gives every CU at least two workgroups (latency hiding comes from
// co-resident blocks); narrow N picks the matching BN so no MFMA columns are wasted.
inline int pick_tile(const GemmArgs& a) {
  auto nb = [&](int bm, int bn) { return mvt_cdiv(a.M, bm) * mvt_cdiv(a.N, bn); };
  if (a.N % 128 != 0 && a.N % 96 == 0) return 1;          // 128 x 96
  if (a.N <= 64) return nb(256, 64) >= 512 ? 2 : 4;      // 256 x 64, else 64 x 64
  if (nb(128, 128) >= 512) return 0;                      // 128 x 128
  if (nb(64, 128) >= 512) return 3;                       // 64 x 128
  return 4;                                               // 64 x 64
}

int launch_gemm(const GemmArgs& a, hipStream_t s) {
  auto blocks = [&](int bm, int bn) { return (unsigned)(mvt_cdiv(a.M, bm) * mvt_cdiv(a.N, bn)); };
  switch (pick_tile(a)) {
    case 1: hipLaunchKernelGGL((gemm_mfma_f32<1, 3, 4, 1>), dim3(blocks(128, 96)), dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL((gemm_mfma_f32<2, 2, 4, 1>), dim3(blocks(256, 64)), dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL((gemm_mfma_f32<1, 2, 2, 2>), dim3(blocks(64, 128)), dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL((gemm_mfma_f32<1, 1, 2, 2>), dim3(blocks(64, 64)), dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((gemm_mfma_f32<2, 2, 2, 2>), dim3(blocks(128, 128)), dim3(256), 0, s, a); break;
  }
  return mvt_launch_status();
}

}  // namespace

extern "C" int mvt_gemm(const float* A, int lda, const float* Wt, int ldw, const float* bias, const float* R, int ldr,
                        float* C, int ldc, int M, int N, int K, int act, void* stream) {
  MVT_REQUIRE(A && Wt && C && M > 0 && N > 0 && K > 0);
  MVT_REQUIRE(lda % 4 == 0 && lda >= ((K + 3) & ~3));
  MVT_REQUIRE(ldw % 32 == 0 && ldw >= ((K + 31) & ~31) && ldc >= N && (!R || ldr >= N));
  MVT_REQUIRE(act >= 0 && act <= 3);
  MVT_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)Wt % 16 == 0));
  GemmArgs a{};
  a.A = A; a.W = Wt; a.bias = bias; a.R = R; a.C = C;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr; a.act = act;
  a.mode = 0;
  a.nk = (K + BK - 1) / BK;
  return launch_gemm(a, mvt_stream(stream));
}

extern "C" int mvt_conv2d(const float* in, const float* wt, const float* bias, float* out, int n, int H, int W, int Cin,
                          int Cout, int KH, int KW, int stride, int pad, int ldo, int act, void* stream) {
  MVT_REQUIRE(in && wt && out && n > 0 && H > 0 && W > 0 && Cout > 0);
  MVT_REQUIRE(KH >= 1 && KH <= 7 && KW >= 1 && KW <= 7 && stride >= 1 && stride <= 2 && pad >= 0 && pad <= 3);
  MVT_REQUIRE(H < 16384 && W < 16384 && ldo >= Cout && act >= 0 && act <= 3);
  MVT_REQUIRE((Cin % 32 == 0) || (Cin == 4));
  MVT_REQUIRE(((uintptr_t)in % 16 == 0) && ((uintptr_t)wt % 16 == 0));
  const int Ho = (H + 2 * pad - KH) / stride + 1;
  const int Wo = (W + 2 * pad - KW) / stride + 1;
  MVT_REQUIRE(Ho > 0 && Wo > 0);
  const long long M = (long long)n * Ho * Wo;
  MVT_REQUIRE(M < (1LL << 31));
  GemmArgs a{};
  a.A = in; a.W = wt; a.bias = bias; a.R = nullptr; a.C = out;
  a.M = (int)M; a.N = Cout; a.ldc = ldo; a.ldr = 0; a.act = act;
  a.H = H; a.Wd = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
  if (Cin == 4) {
    a.mode = 2; a.nk = KH; a.K = KH * 32;
  } else {
    a.mode = 1; a.K = KH * KW * Cin; a.nk = a.K / BK;
  }
  a.ldw = (a.K + 63) & ~63;  // weights are [Cout][round_up(K, 64)], zero padded
  a.lda = 0;
  return launch_gemm(a, mvt_stream(stream));
}

extern "C" int mvt_split_bf16(const float* src, unsigned short* hi, unsigned short* lo, long long n, void* stream) {
  MVT_REQUIRE(src && hi && n > 0 && n % 4 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)hi % 8 == 0));
  long long n4 = n / 4;
  long long g = mvt_cdiv(n4, 256);
  hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, mvt_stream(stream), src, hi, lo, n4);
  return mvt_launch_status();
}

extern "C" int mvt_gemm_bf16(const float* A, int lda, const unsigned short* Whi, const unsigned short* Wlo, int ldw,
                             const float* bias, const float* R, int ldr, void* C, int ldc, int M, int N, int K, int act,
                             int io_flags, void* stream) {
  MVT_REQUIRE(A && Whi && C && M > 0 && N > 0 && K > 0);
  MVT_REQUIRE(lda % 4 == 0 && lda >= ((K + 3) & ~3));
  MVT_REQUIRE(ldw % 64 == 0 && ldw >= ((K + 63) & ~63) && ldc >= N && (!R || ldr >= N));
  MVT_REQUIRE(act >= 0 && act <= 3 && (io_flags & ~MVT_IO_OUT_BF16) == 0);
  MVT_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)Whi % 8 == 0) && ((uintptr_t)Wlo % 8 == 0));
  GemmArgs a{};
  a.A = A; a.Whi = Whi; a.Wlo = Wlo; a.bias = bias; a.R = R; a.C = (float*)C;
  a.c_bf16 = io_flags & MVT_IO_OUT_BF16 ? 1 : 0;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr; a.act = act;
  a.mode = 0;
  a.nk = (K + BKB - 1) / BKB;
  return Wlo ? launch_gemm_bf16<true>(a, mvt_stream(stream)) : launch_gemm_bf16<false>(a, mvt_stream(stream));
}

int mvt_detail_conv_rows_slots(int Ho, int Wo, int tile_rows);
int mvt_detail_conv_rows_tile_rows(int ksize, int stride, int Ho);
int mvt_detail_stem7x7_rows(const float* in, const unsigned short* w, int ldw, const float* bias, void* out, int n, int H, int W, int Cout,
                            int ldo, int io_flags, float* out_partial, hipStream_t stream);
int mvt_detail_conv_rows(const void* in, const unsigned short* w, int ldw, const float* bias, void* out, int n, int H, int W, int Cin,
                         int Cout, int ksize, int stride, int ldo, int io_flags, const float* in_stats, float* out_partial,
                         hipStream_t stream);

extern "C" int mvt_conv2d_stat_slots(int H, int W, int Cin, int KH, int KW, int stride, int pad, int split) {
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return 0;
  if (!split && Cin % 32 == 0 && KH == KW && ((KH == 3 && pad == 1) || (KH == 1 && pad == 0)) && stride <= 2)
    return mvt_detail_conv_rows_slots(Ho, Wo, mvt_detail_conv_rows_tile_rows(KH, stride, Ho));  // row-tile kernels: one slot per tile
  if (KH == 3 && KW == 3 && stride == 1 && pad == 1 && Cin % CK == 0) return (int)(mvt_cdiv(Ho, HT) * mvt_cdiv(Wo, HW_) * 4);
  if (!split && Cin == 4 && KH == 7 && KW == 7 && stride == 2 && pad == 3)
    return mvt_detail_conv_rows_slots(Ho, Wo, mvt_detail_conv_rows_tile_rows(7, 2, Ho));  // stem kernel
  return ((long long)Ho * Wo) % 256 == 0 ? Ho * Wo / 32 : 0;  // im2col tiles are <= 256 rows: they must not straddle images
}

int mvt_detail_conv3x3s2_down(const void* in, const unsigned short* w3, int ldw3, const float* b3, const unsigned short* wd, int ldwd,
                              const float* bd, void* out3, void* outd, int n, int H, int W, int Cin, int Cout, int ldo, float* part3,
                              float* partd, hipStream_t stream);
extern "C" int mvt_conv3x3s2_down_bf16(const void* in, const unsigned short* w3, const float* b3, const unsigned short* wd, const float* bd,
                                       void* out3, void* outd, int n, int H, int W, int Cin, int Cout, int ldo, float* part3,
                                       float* partd, void* stream) {
  MVT_REQUIRE(in && w3 && wd && out3 && outd && n > 0 && H > 1 && W > 1 && H < 16384 && W < 16384 && Cin > 0 && Cout > 0);
  MVT_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0 && (long long)n * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) < (1LL << 31));
  // (weight rows as mvt_conv2d_bf16 takes them: [Cout][round_up(K, 64)], zero padded)
  return mvt_detail_conv3x3s2_down(in, w3, (9 * Cin + BKB - 1) / BKB * BKB, b3, wd, (Cin + BKB - 1) / BKB * BKB, bd, out3, outd, n, H, W, Cin,
                                   Cout, ldo, part3, partd, mvt_stream(stream));
}

extern "C" int mvt_conv2d_bf16(const void* in, const unsigned short* wt_hi, const unsigned short* wt_lo, const float* bias,
                               void* out, int n, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int ldo,
                               int act, int io_flags, const float* in_stats, float* out_partial, void* stream) {
  MVT_REQUIRE(in && wt_hi && out && n > 0 && H > 0 && W > 0 && Cout > 0);
  MVT_REQUIRE(KH >= 1 && KH <= 7 && KW >= 1 && KW <= 7 && stride >= 1 && stride <= 2 && pad >= 0 && pad <= 3);
  MVT_REQUIRE(H < 16384 && W < 16384 && ldo >= Cout && act >= 0 && act <= 3);
  MVT_REQUIRE((Cin % 32 == 0) || (Cin == 4));
  MVT_REQUIRE(((uintptr_t)in % 16 == 0) && ((uintptr_t)wt_hi % 8 == 0) && ((uintptr_t)wt_lo % 8 == 0));
  const int Ho = (H + 2 * pad - KH) / stride + 1;
  const int Wo = (W + 2 * pad - KW) / stride + 1;
  MVT_REQUIRE(Ho > 0 && Wo > 0);
  const long long M = (long long)n * Ho * Wo;
  MVT_REQUIRE(M < (1LL << 31));
  GemmArgs a{};
  a.A = (const float*)in; a.Whi = wt_hi; a.Wlo = wt_lo; a.bias = bias; a.R = nullptr; a.C = (float*)out;
  a.M = (int)M; a.N = Cout; a.ldc = ldo; a.ldr = 0; a.act = act;
  a.H = H; a.Wd = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
  if (Cin == 4) {
    a.mode = 2; a.K = KH * 32;
  } else {
    a.mode = 1; a.K = KH * KW * Cin;
  }
  a.nk = (a.K + BKB - 1) / BKB;
  a.ldw = a.nk * BKB;  // weights are [Cout][round_up(K, 64)], zero padded
  a.lda = 0;
  const bool halo = KH == 3 && KW == 3 && stride == 1 && pad == 1 && Cin % CK == 0;
  // bf16 mode: every 3x3 (pad 1) and 1x1 (pad 0) convolution with Cin % 32 == 0 runs on the row-tile kernels (conv_rows.hip)
  const bool rows = !wt_lo && act == MVT_ACT_NONE && Cin % 32 == 0 && KH == KW && ((KH == 3 && pad == 1) || (KH == 1 && pad == 0));
  MVT_REQUIRE(!in_stats || ((halo || rows) && ((uintptr_t)in_stats % 16 == 0)));  // normalise-on-load: halo / row-tile kernels only
  a.in_stats = in_stats;
  a.out_part = out_partial;
  a.a_bf16 = io_flags & MVT_IO_IN_BF16 ? 1 : 0;
  a.c_bf16 = io_flags & MVT_IO_OUT_BF16 ? 1 : 0;
  MVT_REQUIRE((io_flags & ~(MVT_IO_IN_BF16 | MVT_IO_OUT_BF16 | MVT_IO_SHORT_WG)) == 0);
  MVT_REQUIRE(!(io_flags & (MVT_IO_IN_BF16 | MVT_IO_OUT_BF16)) || !wt_lo);  // bf16 tensors: bf16 mode only
  MVT_REQUIRE(!a.a_bf16 || (Cin % 32 == 0 && (uintptr_t)in % 8 == 0));                          // (the stem reads fp32 RGB)
  a.slots = mvt_conv2d_stat_slots(H, W, Cin, KH, KW, stride, pad, wt_lo != nullptr);
  MVT_REQUIRE(!out_partial || (a.slots > 0 && (wt_lo || act == MVT_ACT_NONE)));
  if (!wt_lo && Cin == 4 && KH == 7 && KW == 7 && stride == 2 && pad == 3 && Cout <= 64 && act == MVT_ACT_NONE)
    return mvt_detail_stem7x7_rows((const float*)in, wt_hi, a.ldw, bias, out, n, H, W, Cout, ldo, io_flags, out_partial, mvt_stream(stream));
  if (rows)
    return mvt_detail_conv_rows(in, wt_hi, a.ldw, bias, out, n, H, W, Cin, Cout, KH, stride, ldo, io_flags, in_stats, out_partial,
                                mvt_stream(stream));
  if (KH == 3 && KW == 3 && stride == 1 && pad == 1 && Cin % CK == 0 && (long long)n * mvt_cdiv(Ho, HT) * mvt_cdiv(Wo, HW_) * 4 < (1LL << 31))
    return wt_lo ? launch_conv3x3_halo<true>(a, n, mvt_stream(stream)) : launch_conv3x3_halo<false>(a, n, mvt_stream(stream));
  return wt_lo ? launch_gemm_bf16<true>(a, mvt_stream(stream)) : launch_gemm_bf16<false>(a, mvt_stream(stream));
}

extern "C" int mvt_ln_gemm_bf16(const float* A, int lda, const float* ln_w, const float* ln_b, float ln_eps,
                                const unsigned short* Whi, const unsigned short* Wlo, int ldw, const float* bias, const float* R,
                                int ldr, float* C, int ldc, int M, int N, int K, int act, void* stream) {
  MVT_REQUIRE(A && Whi && C && M > 0 && N > 0 && K > 0 && K % 4 == 0 && K <= 1024);
  MVT_REQUIRE((ln_w == nullptr) == (ln_b == nullptr) && ((uintptr_t)ln_w % 16 == 0) && ((uintptr_t)ln_b % 16 == 0));
  MVT_REQUIRE(lda % 4 == 0 && lda >= K);
  MVT_REQUIRE(ldw % 64 == 0 && ldw >= ((K + 63) & ~63) && ldc >= N && (!R || ldr >= N));
  MVT_REQUIRE(act >= 0 && act <= 3);
  MVT_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)Whi % 8 == 0) && ((uintptr_t)Wlo % 8 == 0));
  GemmArgs a{};
  a.A = A; a.Whi = Whi; a.Wlo = Wlo; a.bias = bias; a.R = R; a.C = C;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr; a.act = act;
  a.mode = 0;
  a.nk = (K + BKB - 1) / BKB;
  a.ln = 1; a.ln_eps = ln_eps; a.ln_w = ln_w; a.ln_b = ln_b;
  return Wlo ? launch_gemm_bf16<true>(a, mvt_stream(stream)) : launch_gemm_bf16<false>(a, mvt_stream(stream));
}
