// Fused transformer MLP for the updater (cotracker2/blocks.py:297-300, 334-337):
//     x += fc2( gelu_tanh( fc1( LayerNorm(x) ) ) )
// in ONE kernel on the bf16 matrix cores.  The unfused path costs a LayerNorm pass plus two GEMMs that write
// and re-read the [M][4C] hidden activations through HBM; here the hidden activations never leave registers.
//
// A workgroup owns 128 token rows and runs 8 waves: wave (mb, jh) handles the 32-token block mb and the
// half jh of every 64-wide hidden chunk, so each SIMD hosts two waves whose GEMM / GELU / LDS phases overlap.
// Everything is computed transposed so that the hidden tile feeds the second GEMM straight from the accumulator
// (no LDS round trip):
//     H^T (j x m) = W1c (j x C) . X^T          A = W1 rows (LDS), B = X rows (LDS, LayerNorm applied, bf16)
//     out^T (n x m) += W2c (n x j) . H^T       A = W2 rows (LDS), B = gelu(H^T) taken from the accumulator
// A 32x32 MFMA accumulator holds its column (token m) on the lane and its rows (hidden unit j) in the 16
// registers; registers 8s..8s+7 are exactly the B fragment of k-step s with the k order
// j = 16s + 8(e>>2) + 4h + (e&3), so the W2 fragment is fetched in that order (two 8-byte LDS reads).
// Weights stream through LDS in chunks of 64 hidden units (W1: 64 x C, W2: C x 64), prefetched one chunk ahead
// in registers; the two partial outputs of a token block are summed through LDS at the end.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128;  // token rows per workgroup
constexpr int HC = 64;   // hidden units per chunk
constexpr int NT = 512;  // threads per workgroup

struct MlpArgs {
  float* x;
  int ldx;
  const unsigned short* w1;  // [H][ldw1] bf16, ldw1 >= C
  const float* b1;
  const unsigned short* w2;  // [C][ldw2] bf16, ldw2 >= H
  const float* b2;
  int ldw1, ldw2;
  long long M;
  int H;
  float eps;
};

template <int C>
__global__ __launch_bounds__(NT) void mlp_fused_bf16(MlpArgs p) {
  constexpr int LDX = C + 8;    // bf16 elements per X / W1 LDS row
  constexpr int LDW2 = HC + 8;  // bf16 elements per W2 LDS row
  constexpr int NB = C / 32;    // output channel blocks
  constexpr int W1F = HC * (C / 8) / NT;  // 16-B pieces per thread per chunk
  constexpr int W2F = C * (HC / 8) / NT;
  constexpr int LDS_ELEMS = BM * LDX + HC * LDX + C * LDW2;
  static_assert(LDS_ELEMS * 2 >= 4 * NB * 16 * 64 * 4, "the pair-reduction image overlays the staging buffers");
  __shared__ __attribute__((aligned(16))) unsigned short lds[LDS_ELEMS];
  __shared__ float b1s[4 * C];  // fc1 bias (H <= 4C): no global load may sit inside the chunk loop (vmcnt is in-order)
  unsigned short* Xs = lds;
  unsigned short* W1s = Xs + BM * LDX;
  unsigned short* W2s = W1s + HC * LDX;

  const int t = threadIdx.x;
  const long long m0 = (long long)blockIdx.x * BM;

  for (int i = t; i < p.H; i += NT) b1s[i] = p.b1[i];
  // ---- LayerNorm of the 128 x C tile -> bf16 in LDS (four threads per row)
  {
    const int row = t >> 2, part = t & 3;
    const long long m = m0 + row;
    const bool ok = m < p.M;
    const float* xr = p.x + (ok ? m : 0) * (long long)p.ldx + part * (C / 4);
    f32x4 v[C / 16];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < C / 16; ++i) {
      v[i] = *reinterpret_cast<const f32x4*>(xr + i * 4);
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    const float mean = s / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < C / 16; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[i][e] - mean;
        ss = fmaf(d, d, ss);
      }
    ss += __shfl_xor(ss, 1, 64);
    ss += __shfl_xor(ss, 2, 64);
    const float rstd = 1.0f / sqrtf(ss / (float)C + p.eps);
#pragma unroll
    for (int i = 0; i < C / 16; ++i) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = ok ? (v[i][e] - mean) * rstd : 0.f;
      bf16x4 b = __builtin_convertvector(o, bf16x4);
      *reinterpret_cast<uint2*>(&Xs[row * LDX + part * (C / 4) + i * 4]) = __builtin_bit_cast(uint2, b);
    }
  }

  // ---- weight chunk loaders (16-B pieces); fixed per-thread offsets, advanced by the chunk index
  u32x4 r1[W1F], r2[W2F];
  const unsigned short* g1[W1F];
  const unsigned short* g2[W2F];
  int l1[W1F], l2[W2F];
#pragma unroll
  for (int i = 0; i < W1F; ++i) {
    const int u = t + NT * i;
    const int row = u / (C / 8), part = u - row * (C / 8);
    g1[i] = p.w1 + (long long)row * p.ldw1 + part * 8;
    l1[i] = row * LDX + part * 8;
  }
#pragma unroll
  for (int i = 0; i < W2F; ++i) {
    const int u = t + NT * i;
    const int row = u / (HC / 8), part = u - row * (HC / 8);
    g2[i] = p.w2 + (long long)row * p.ldw2 + part * 8;
    l2[i] = row * LDW2 + part * 8;
  }
#define LOAD_W(hc_)                                                                                     \
  {                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < W1F; ++i) r1[i] = *reinterpret_cast<const u32x4*>(g1[i] + (long long)(hc_) * HC * p.ldw1); \
    _Pragma("unroll") for (int i = 0; i < W2F; ++i) r2[i] = *reinterpret_cast<const u32x4*>(g2[i] + (hc_) * HC);                    \
  }
#define STORE_W()                                                                                       \
  {                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < W1F; ++i) *reinterpret_cast<u32x4*>(&W1s[l1[i]]) = r1[i];     \
    _Pragma("unroll") for (int i = 0; i < W2F; ++i) *reinterpret_cast<u32x4*>(&W2s[l2[i]]) = r2[i];     \
  }

  const int wave = t >> 6, lane = t & 63;
  const int mb = wave & 3, jh = wave >> 2;  // token block, hidden half
  const int r = lane & 31, h = lane >> 5;
  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[nb][e] = 0.f;

  const unsigned short* xrow = &Xs[(mb * 32 + r) * LDX + h * 8];
  const unsigned short* wrow = &W1s[(jh * 32 + r) * LDX + h * 8];
  const unsigned short* w2row = &W2s[r * LDW2 + jh * 32 + 4 * h];

  const int nchunk = p.H / HC;
  LOAD_W(0);
  for (int hc = 0; hc < nchunk; ++hc) {
    STORE_W();
    __syncthreads();
    if (hc + 1 < nchunk) LOAD_W(hc + 1);
    // GEMM 1: H^T block (32 hidden units x 32 tokens); next k-step's fragments are read before the current MFMA
    f32x16 hacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) hacc[e] = 0.f;
    {
      uint4 xc = *reinterpret_cast<const uint4*>(xrow);
      uint4 wc = *reinterpret_cast<const uint4*>(wrow);
#pragma unroll
      for (int ks = 0; ks < C / 16; ++ks) {
        uint4 xn = xc, wn = wc;
        if (ks + 1 < C / 16) {
          xn = *reinterpret_cast<const uint4*>(xrow + (ks + 1) * 16);
          wn = *reinterpret_cast<const uint4*>(wrow + (ks + 1) * 16);
        }
        hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wc), __builtin_bit_cast(bf16x8, xc), hacc, 0, 0, 0);
        xc = xn;
        wc = wn;
      }
    }
    // bias + GELU(tanh) on the accumulator -> the two B fragments of GEMM 2
    bf16x8 hb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f32x4 lo4, hi4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int e0 = 8 * s + e, e1 = 8 * s + 4 + e;
        const int j0 = hc * HC + jh * 32 + (e0 & 3) + 8 * (e0 >> 2) + 4 * h;
        const int j1 = hc * HC + jh * 32 + (e1 & 3) + 8 * (e1 >> 2) + 4 * h;
        lo4[e] = mvt_gelu_tanh(hacc[e0] + b1s[j0]);
        hi4[e] = mvt_gelu_tanh(hacc[e1] + b1s[j1]);
      }
      const bf16x4 bl = __builtin_convertvector(lo4, bf16x4), bh = __builtin_convertvector(hi4, bf16x4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hb[s][e] = bl[e];
        hb[s][4 + e] = bh[e];
      }
    }
    // GEMM 2 straight from registers; W2 fragment in the accumulator's k order: j = 16s + 4h + (0..3) and + 8
    {
      uint2 a0[2], a1[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        a0[s] = *reinterpret_cast<const uint2*>(w2row + 16 * s);
        a1[s] = *reinterpret_cast<const uint2*>(w2row + 16 * s + 8);
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        uint2 n0[2] = {a0[0], a0[1]}, n1[2] = {a1[0], a1[1]};
        if (nb + 1 < NB) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            n0[s] = *reinterpret_cast<const uint2*>(w2row + (nb + 1) * 32 * LDW2 + 16 * s);
            n1[s] = *reinterpret_cast<const uint2*>(w2row + (nb + 1) * 32 * LDW2 + 16 * s + 8);
          }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const bf16x8 wa = __builtin_bit_cast(bf16x8, make_uint4(a0[s].x, a0[s].y, a1[s].x, a1[s].y));
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, hb[s], acc[nb], 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          a0[s] = n0[s];
          a1[s] = n1[s];
        }
      }
    }
    __syncthreads();
  }

  // ---- sum the two hidden halves of every token block through LDS ([mb][nb*16+e][lane] floats)
  float* red = reinterpret_cast<float*>(lds) + mb * (NB * 16 * 64);
  if (jh == 1) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) red[(nb * 16 + e) * 64 + lane] = acc[nb][e];
  }
  __syncthreads();
  if (jh == 1) return;
  // ---- epilogue: out^T rows = channels (registers), column = token (lane): x[m][n..n+3] += acc + b2
  const long long m = m0 + mb * 32 + r;
  if (m < p.M) {
    float* xr = p.x + m * (long long)p.ldx;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nb * 32 + 8 * g + 4 * h;
        f32x4 v = *reinterpret_cast<const f32x4*>(xr + n);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.b2 + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (acc[nb][4 * g + e] + red[(nb * 16 + 4 * g + e) * 64 + lane]) + b[e];
        *reinterpret_cast<f32x4*>(xr + n) = v;
      }
    }
  }
}
#undef LOAD_W
#undef STORE_W

}  // namespace

extern "C" int mvt_mlp_fused_bf16(float* x, int ldx, const unsigned short* w1, int ldw1, const float* b1,
                                  const unsigned short* w2, int ldw2, const float* b2, long long M, int C, int H, float eps,
                                  void* stream) {
  MVT_REQUIRE(x && w1 && b1 && w2 && b2 && M > 0 && H > 0 && H % HC == 0 && H <= 4 * C);
  MVT_REQUIRE(ldx % 4 == 0 && ldx >= C && ldw1 % 8 == 0 && ldw1 >= C && ldw2 % 8 == 0 && ldw2 >= H);
  MVT_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)w1 % 16 == 0) && ((uintptr_t)w2 % 16 == 0) && ((uintptr_t)b2 % 16 == 0));
  MlpArgs a{x, ldx, w1, b1, w2, b2, ldw1, ldw2, M, H, eps};
  const unsigned blocks = (unsigned)mvt_cdiv(M, BM);
  switch (C) {
    case 256: hipLaunchKernelGGL((mlp_fused_bf16<256>), dim3(blocks), dim3(NT), 0, mvt_stream(stream), a); break;
    default: return MVT_ERR_ARG;
  }
  return mvt_launch_status();
}
