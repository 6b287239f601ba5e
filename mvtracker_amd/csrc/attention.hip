// Small-sequence attention for the updater (time attention over 12 frames, virtual<->point cross
// attention over N tracks, virtual self attention over 64 tokens; cotracker2/blocks.py:258-271).
//
// One wave per (group, head, query).  Keys live on lanes: every lane scores its own key row
// against the wave-uniform query (q in scalar registers), keeps a private online-softmax state
// (m, l, acc[dh]) over its strided key subset, and the 64 partial states are merged once at the
// end with shuffle reductions.  Strided row addressing (group stride / item stride in rows) lets
// the same kernel walk the track-major token buffer along time or along tracks with no permute.
#include "common.h"

namespace {

template <int DH>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ q, int ldq, long long q_gs, long long q_is,
                                                        const float* __restrict__ k, const float* __restrict__ v, int ldkv, long long k_gs,
                                                        long long k_is, float* __restrict__ o, int ldo, int groups, int nq, int nk,
                                                        int heads) {
  const int lane = threadIdx.x & 63;
  const long long task = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long ntask = (long long)groups * heads * nq;
  if (task >= ntask) return;
  const int qi = (int)(task % nq);
  const int hd = (int)((task / nq) % heads);
  const long long g = task / ((long long)nq * heads);

  const long long qrow = g * q_gs + (long long)qi * q_is;
  const float* qp = q + qrow * ldq + hd * DH;
  float qv[DH];
#pragma unroll
  for (int d = 0; d < DH; d += 4) {
    f32x4 t = *reinterpret_cast<const f32x4*>(qp + d);
#pragma unroll
    for (int e = 0; e < 4; ++e) qv[d + e] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(t[e])));
  }
  const float scale = 1.0f / sqrtf((float)DH);

  float m = -INFINITY, l = 0.f;
  float acc[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) acc[d] = 0.f;

  for (int j = lane; j < nk; j += 64) {
    const long long krow = g * k_gs + (long long)j * k_is;
    const float* kp = k + krow * ldkv + hd * DH;
    const float* vp = v + krow * ldkv + hd * DH;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
      f32x4 t = *reinterpret_cast<const f32x4*>(kp + d);
      s = fmaf(qv[d], t[0], s);
      s = fmaf(qv[d + 1], t[1], s);
      s = fmaf(qv[d + 2], t[2], s);
      s = fmaf(qv[d + 3], t[3], s);
    }
    s *= scale;
    const float mn = fmaxf(m, s);
    const float corr = expf(m - mn);  // exp(-inf) = 0 on the first key
    const float p = expf(s - mn);
    l = l * corr + p;
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
      f32x4 t = *reinterpret_cast<const f32x4*>(vp + d);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[d + e] = fmaf(p, t[e], acc[d + e] * corr);
    }
    m = mn;
  }
  // merge the 64 lane states
  const float M = wave_max(m);
  const float w = (m == -INFINITY) ? 0.f : expf(m - M);
  const float L = wave_sum(l * w);
  const float inv = 1.0f / L;
  float* op = o + qrow * ldo + hd * DH;
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    const float r = wave_sum(acc[d] * w) * inv;
    if (lane == (d & 63)) op[d] = r;
  }
}

}  // namespace

extern "C" int mvt_attention(const float* q, int ldq, long long q_gs, long long q_is, const float* k, const float* v, int ldkv,
                             long long k_gs, long long k_is, float* o, int ldo, int groups, int nq, int nk, int heads, int dh,
                             void* stream) {
  MVT_REQUIRE(q && k && v && o && groups > 0 && nq > 0 && nk > 0 && heads > 0);
  MVT_REQUIRE(ldq % 4 == 0 && ldkv % 4 == 0 && ldq >= heads * dh && ldkv >= heads * dh && ldo >= heads * dh);
  MVT_REQUIRE(((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0));
  const long long ntask = (long long)groups * heads * nq;
  const unsigned blocks = (unsigned)mvt_cdiv(ntask, 4);
#define LAUNCH(DH)                                                                                                                \
  hipLaunchKernelGGL((attention_kernel<DH>), dim3(blocks), dim3(256), 0, mvt_stream(stream), q, ldq, q_gs, q_is, k, v, ldkv, k_gs, k_is, \
                     o, ldo, groups, nq, nk, heads)
  switch (dh) {
    case 32: LAUNCH(32); break;
    case 48: LAUNCH(48); break;
    case 64: LAUNCH(64); break;
    default: return MVT_ERR_ARG;
  }
#undef LAUNCH
  return mvt_launch_status();
}
