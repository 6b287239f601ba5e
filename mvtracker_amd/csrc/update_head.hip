// Fused "head" of a refinement iteration on the bf16 matrix cores: the updater's flow head (cotracker2/blocks.py:489,
// Linear 256 -> 131 -> 131 -> 131 with ReLU) and the track / feature update that consumes it (mvtracker.py:392-399):
//     delta  = W4 . relu(W2 . relu(W0 . tok + b0) + b2) + b4                       [rows][131]
//     coords += delta[:, 0:3]                                    (NaN guard: flag set when a coordinate is NaN, :401-404)
//     ffeats += gelu_erf(Wu . GroupNorm1(delta[:, 3:131]) + bu)                    GroupNorm(1, 128), eps 1e-5, affine
// Unfused this is three GEMM launches, delta_split and one more GEMM whose 12288 x 131 intermediates round-trip through HBM
// and whose five launch boundaries sit on the critical path of every iteration.  One workgroup owns 64 token rows; every
// activation stays in LDS (bf16 operands, fp32 delta), the four weight matrices (168 KB of bf16, L2 resident) are streamed
// as fragment-major MFMA A operands exactly like the block kernel does (out^T = W . act^T).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int NT = 512, BM = 64, C = 256, OUT = 131, NB = 5 /* 32-column blocks of the 131 outputs */, CF = 128;
constexpr int LDX = C + 8, LDH = NB * 32 + 8, LDD = 132, LDN = CF + 8;
constexpr int K1 = 9;   // k-steps of the 131-wide layers (131 -> 144)
constexpr int FS = 512; // elements per (32-row block, k-step) weight fragment

struct HeadArgs {
  const float* tok; int ldt;
  const unsigned short *w0, *w2, *w4, *wu;  // fragment-major bf16: [160][256], [160][144], [160][144], [128][128]
  const float *b0, *b2, *b4, *bu, *gw, *gb;
  float* coords;   // [rows][3], updated in place
  float* ffeats;   // [rows][128], updated in place
  float* delta;    // optional [rows][ldd] copy of delta (tracing / tests)
  int ldd;
  long long rows;
  int* nan_flag;
};

__device__ __forceinline__ bf16x8 ldg_frag(const unsigned short* p) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p)); }
__device__ __forceinline__ bf16x8 lds_frag(const unsigned short* p) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p)); }

// acc[mb] += W[n0 + r][k] * Act[mb*32 + r'][k], k in [0, 16*KS): W fragments from global (four k-steps ahead), Act from LDS
template <int KS>
__device__ __forceinline__ void gemm_wt(f32x16 (&acc)[2], const unsigned short* wrow, const unsigned short* arow, int lda) {
  constexpr int PF = 4;
  bf16x8 wq[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) wq[i] = ldg_frag(wrow + (i < KS ? i : 0) * FS);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const bf16x8 wa = wq[ks % PF];
    if (ks + PF < KS) wq[ks % PF] = ldg_frag(wrow + (ks + PF) * FS);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bf16x8 xb = lds_frag(arow + i * 32 * lda + ks * 16);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, xb, acc[i], 0, 0, 0);
    }
  }
}

__global__ __launch_bounds__(NT) void update_head_kernel(HeadArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned short Xs[BM * LDX];   // tokens (bf16); later the fp32 delta tile
  __shared__ __attribute__((aligned(16))) unsigned short Ha[BM * LDH], Hb[BM * LDH];
  __shared__ __attribute__((aligned(16))) unsigned short Dn[BM * LDN];
  __shared__ float bs[3 * NB * 32 + 3 * CF];  // b0 | b2 | b4 (padded to 160) | bu | gw | gb
  static_assert(BM * LDD * 4 <= BM * LDX * 2, "the fp32 delta tile fits the token tile");
  float* Dl = reinterpret_cast<float*>(Xs);
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, r = lane & 31, h = lane >> 5;
  const long long m0 = (long long)blockIdx.x * BM;

  for (int i = t; i < NB * 32; i += NT) {
    bs[i] = i < OUT ? p.b0[i] : 0.f;
    bs[NB * 32 + i] = i < OUT ? p.b2[i] : 0.f;
    bs[2 * NB * 32 + i] = i < OUT ? p.b4[i] : 0.f;
  }
  for (int i = t; i < CF; i += NT) {
    bs[3 * NB * 32 + i] = p.bu[i];
    bs[3 * NB * 32 + CF + i] = p.gw[i];
    bs[3 * NB * 32 + 2 * CF + i] = p.gb[i];
  }
  // token tile fp32 -> bf16
  for (int f = t; f < BM * (C / 4); f += NT) {
    const int row = f / (C / 4), c = (f - row * (C / 4)) * 4;
    const long long m = m0 + row;
    u32x2 w = (u32x2){0u, 0u};
    if (m < p.rows) w = __builtin_bit_cast(u32x2, __builtin_convertvector(*reinterpret_cast<const f32x4*>(p.tok + m * (long long)p.ldt + c), bf16x4));
    *reinterpret_cast<u32x2*>(&Xs[row * LDX + c]) = w;
  }
  __syncthreads();

  // one 131-wide layer: waves 0..4 own the five 32-column blocks; out -> bf16 tile (ReLU) or the fp32 delta tile
  auto layer = [&](auto ks_tag, const unsigned short* wfrag, const unsigned short* act, int lda, const float* bias, unsigned short* dst_bf,
                   float* dst_f32) {
    constexpr int KS = decltype(ks_tag)::value;
    if (wave < NB) {
      f32x16 acc[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
      gemm_wt<KS>(acc, wfrag + ((long long)wave * KS * 64 + lane) * 8, act + r * lda + 8 * h, lda);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = wave * 32 + 8 * g + 4 * h;
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = acc[i][4 * g + e] + bias[n + e];
          if (dst_bf) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
            *reinterpret_cast<u32x2*>(&dst_bf[(i * 32 + r) * LDH + n]) = __builtin_bit_cast(u32x2, __builtin_convertvector(o, bf16x4));
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < LDD) dst_f32[(i * 32 + r) * LDD + n + e] = o[e];
          }
        }
    }
  };
  layer(std::integral_constant<int, C / 16>{}, p.w0, Xs, LDX, bs, Ha, nullptr);
  __syncthreads();
  layer(std::integral_constant<int, K1>{}, p.w2, Ha, LDH, bs + NB * 32, Hb, nullptr);
  __syncthreads();  // (also: every wave is done reading Xs -- layer 0 -- before the delta tile overwrites it)
  layer(std::integral_constant<int, K1>{}, p.w4, Hb, LDH, bs + 2 * NB * 32, nullptr, Dl);
  __syncthreads();

  // per token: coords += delta[0:3]; GroupNorm(1, 128) of delta[3:131] (two-pass, biased variance, eps 1e-5) -> bf16 operand tile.
  // Eight lanes per token, 16 channels each.
  {
    const int row = t >> 3, sub = t & 7;
    const long long m = m0 + row;
    const float* d = Dl + row * LDD;
    if (sub < 3 && m < p.rows) {
      const float nc = p.coords[m * 3 + sub] + d[sub];
      p.coords[m * 3 + sub] = nc;
      if (p.nan_flag && nc != nc) atomicOr(p.nan_flag, 1);
    }
    if (p.delta && m < p.rows)
      for (int c = sub; c < OUT; c += 8) p.delta[m * (long long)p.ldd + c] = d[c];
    float v[16];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      v[i] = d[3 + sub + 8 * i];
      s += v[i];
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)CF;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float dd = v[i] - mean;
      ss += dd * dd;
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float rstd = 1.0f / sqrtf(ss / (float)CF + 1e-5f);
    const float* gw = bs + 3 * NB * 32 + CF;
    const float* gb = gw + CF;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = sub + 8 * i;
      Dn[row * LDN + c] = mvt_bf16_bits((v[i] - mean) * rstd * gw[c] + gb[c]);
    }
  }
  __syncthreads();

  // ffeats += gelu_erf(Wu . dn + bu): waves 0..3 own the four 32-channel blocks
  if (wave < CF / 32) {
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    gemm_wt<CF / 16>(acc, p.wu + ((long long)wave * (CF / 16) * 64 + lane) * 8, Dn + r * LDN + 8 * h, LDN);
    const float* bu = bs + 3 * NB * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long long m = m0 + i * 32 + r;
      if (m >= p.rows) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = wave * 32 + 8 * g + 4 * h;
        float* fp = p.ffeats + m * CF + n;
        f32x4 o = *reinterpret_cast<const f32x4*>(fp);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += mvt_gelu_erf(acc[i][4 * g + e] + bu[n + e]);
        *reinterpret_cast<f32x4*>(fp) = o;
      }
    }
  }
}

}  // namespace

extern "C" int mvt_update_head_bf16(const float* tok, int ldt, const unsigned short* w0, const float* b0, const unsigned short* w2,
                                    const float* b2, const unsigned short* w4, const float* b4, const float* gn_w, const float* gn_b,
                                    const unsigned short* wu, const float* bu, float* coords, float* ffeats, float* delta, int ldd,
                                    long long rows, int hidden, int out_dim, int* nan_flag, void* stream) {
  MVT_REQUIRE(tok && w0 && b0 && w2 && b2 && w4 && b4 && gn_w && gn_b && wu && bu && coords && ffeats && rows > 0);
  MVT_REQUIRE(hidden == C && out_dim == OUT && ldt % 4 == 0 && ldt >= C && (!delta || ldd >= OUT));
  MVT_REQUIRE(((uintptr_t)tok % 16 == 0) && ((uintptr_t)w0 % 16 == 0) && ((uintptr_t)w2 % 16 == 0) && ((uintptr_t)w4 % 16 == 0) &&
              ((uintptr_t)wu % 16 == 0) && ((uintptr_t)ffeats % 16 == 0));
  HeadArgs a{};
  a.tok = tok; a.ldt = ldt; a.w0 = w0; a.w2 = w2; a.w4 = w4; a.wu = wu; a.b0 = b0; a.b2 = b2; a.b4 = b4; a.bu = bu; a.gw = gn_w; a.gb = gn_b;
  a.coords = coords; a.ffeats = ffeats; a.delta = delta; a.ldd = ldd; a.rows = rows; a.nan_flag = nan_flag;
  hipLaunchKernelGGL(update_head_kernel, dim3((unsigned)mvt_cdiv(rows, BM)), dim3(NT), 0, mvt_stream(stream), a);
  return mvt_launch_status();
}
