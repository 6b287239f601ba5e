// Small-sequence attention for the updater (time attention over 12 frames, virtual<->point cross
// attention over N tracks, virtual self attention over 64 tokens; cotracker2/blocks.py:258-271).
//
// Two mappings, both walking the track-major token buffer through (group stride, item stride) row
// addressing so that "along time" and "along tracks" need no permute:
//   * nk <= 64  (time, virtual-self, point<-virtual): one LANE per (group, head, query).  The lane
//     keeps q, the online-softmax state and the dh-wide accumulator in registers and streams the
//     keys; lanes of a wave that share (group, head) read the same K/V rows (broadcast loads), and
//     nothing is ever reduced across lanes.
//   * nk > 64   (virtual<-point): one WAVE per (group, head, query), keys on lanes, each lane a private
//     online-softmax state over its strided key subset; the 64 partial states are merged once through
//     LDS (conflict-free [lane][dh+1] image, lane d sums column d).
#include "common.h"

namespace {

template <int DH>
__global__ __launch_bounds__(256) void attention_qlane_kernel(const float* __restrict__ q, int ldq, long long q_gs, long long q_is,
                                                              const float* __restrict__ k, const float* __restrict__ v, int ldkv,
                                                              long long k_gs, long long k_is, float* __restrict__ o, int ldo,
                                                              int groups, int nq, int nk, int heads) {
  const long long task = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long ntask = (long long)groups * heads * nq;
  if (task >= ntask) return;
  const int qi = (int)(task % nq);
  const int hd = (int)((task / nq) % heads);
  const long long g = task / ((long long)nq * heads);
  const long long qrow = g * q_gs + (long long)qi * q_is;
  const float* qp = q + qrow * ldq + hd * DH;
  const float scale = 1.0f / sqrtf((float)DH);
  float qv[DH], acc[DH];
#pragma unroll
  for (int d = 0; d < DH; d += 4) {
    f32x4 t = *reinterpret_cast<const f32x4*>(qp + d);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      qv[d + e] = t[e];
      acc[d + e] = 0.f;
    }
  }
  float m = -INFINITY, l = 0.f;
  const float* kp = k + (g * k_gs) * ldkv + hd * DH;
  const float* vp = v + (g * k_gs) * ldkv + hd * DH;
  const long long kstep = k_is * ldkv;
#pragma unroll 1
  for (int j = 0; j < nk; ++j, kp += kstep, vp += kstep) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
      f32x4 t = *reinterpret_cast<const f32x4*>(kp + d);
      s0 = fmaf(qv[d], t[0], s0);
      s1 = fmaf(qv[d + 1], t[1], s1);
      s2 = fmaf(qv[d + 2], t[2], s2);
      s3 = fmaf(qv[d + 3], t[3], s3);
    }
    const float s = ((s0 + s1) + (s2 + s3)) * scale;
    const float mn = fmaxf(m, s);
    const float corr = __expf(m - mn);
    const float p = __expf(s - mn);
    l = l * corr + p;
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
      f32x4 t = *reinterpret_cast<const f32x4*>(vp + d);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[d + e] = fmaf(p, t[e], acc[d + e] * corr);
    }
    m = mn;
  }
  const float inv = 1.0f / l;
  float* op = o + qrow * ldo + hd * DH;
#pragma unroll
  for (int d = 0; d < DH; d += 4) {
    f32x4 t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = acc[d + e] * inv;
    *reinterpret_cast<f32x4*>(op + d) = t;
  }
}

template <int DH>
__global__ __launch_bounds__(256) void attention_klane_kernel(const float* __restrict__ q, int ldq, long long q_gs, long long q_is,
                                                              const float* __restrict__ k, const float* __restrict__ v, int ldkv,
                                                              long long k_gs, long long k_is, float* __restrict__ o, int ldo,
                                                              int groups, int nq, int nk, int heads) {
  __shared__ float red[4][64 * (DH + 1)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long task = (long long)blockIdx.x * 4 + wave;
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
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
      f32x4 t = *reinterpret_cast<const f32x4*>(kp + d);
      s0 = fmaf(qv[d], t[0], s0);
      s1 = fmaf(qv[d + 1], t[1], s1);
      s2 = fmaf(qv[d + 2], t[2], s2);
      s3 = fmaf(qv[d + 3], t[3], s3);
    }
    const float s = ((s0 + s1) + (s2 + s3)) * scale;
    const float mn = fmaxf(m, s);
    const float corr = __expf(m - mn);  // exp(-inf) = 0 on the first key
    const float p = __expf(s - mn);
    l = l * corr + p;
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
      f32x4 t = *reinterpret_cast<const f32x4*>(vp + d);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[d + e] = fmaf(p, t[e], acc[d + e] * corr);
    }
    m = mn;
  }
  // merge the 64 lane states through LDS
  const float M = wave_max(m);
  const float w = (m == -INFINITY) ? 0.f : __expf(m - M);
  const float L = wave_sum(l * w);
  float* r = red[wave];
#pragma unroll
  for (int d = 0; d < DH; ++d) r[lane * (DH + 1) + d] = acc[d] * w;
  __builtin_amdgcn_wave_barrier();
  if (lane < DH) {
    float sum = 0.f;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) sum += r[i * (DH + 1) + lane];
    o[qrow * ldo + hd * DH + lane] = sum / L;
  }
}

// Queries on lanes, keys wave-uniform: a wave owns 64 consecutive queries of one (group, head); every key / value
// row is the same for all lanes.  The wave stages CH keys at a time into its private LDS slice (coalesced 16-B
// loads, the next chunk already in flight in registers while the current one is consumed) and reads them back as
// broadcast ds_read_b128, so the FMAs never wait on global latency.  KS > 1 splits the keys of one query chunk over
// KS waves whose online-softmax states are merged through LDS (virtual<-point attention: 64 queries x 1024 keys).
template <int DH, int KS>
__global__ __launch_bounds__(KS == 1 ? 256 : KS * 64) void attention_ukeys_kernel(
    const float* __restrict__ q, int ldq, long long q_gs, long long q_is, const float* __restrict__ k, const float* __restrict__ v,
    int ldkv, long long k_gs, long long k_is, float* __restrict__ o, int ldo, int groups, int nq, int nk, int heads) {
  constexpr int NW = KS == 1 ? 4 : KS;         // waves per block
  constexpr int CH = 16;                       // keys per staged chunk
  constexpr int Q4 = DH / 4;                   // float4 per row
  constexpr int NLD = CH * Q4 / 64;            // float4 per lane per chunk (K and V each)
  constexpr int STAGE = NW * 2 * CH * DH;      // floats
  constexpr int MERGE = KS > 1 ? (KS - 1) * 64 * (DH + 2) : 0;
  __shared__ __attribute__((aligned(16))) float lds[STAGE > MERGE ? STAGE : MERGE];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* sk = lds + wave * 2 * CH * DH;
  float* sv = sk + CH * DH;
  const int chunks = (nq + 63) / 64;
  const long long nchunk = (long long)groups * heads * chunks;
  const long long chunk_id = KS == 1 ? (long long)blockIdx.x * 4 + wave : (long long)blockIdx.x;
  const bool chunk_ok = chunk_id < nchunk;  // KS == 1 only: trailing waves of the last block
  const long long cid = chunk_ok ? chunk_id : 0;
  const int qc = (int)(cid % chunks);
  const int hd = (int)((cid / chunks) % heads);
  const long long g = cid / ((long long)chunks * heads);
  const int qi = qc * 64 + lane;
  const bool active = chunk_ok && qi < nq;
  const long long qrow = g * q_gs + (long long)(active ? qi : 0) * q_is;
  const float* qp = q + qrow * ldq + hd * DH;
  const float scale = 1.0f / sqrtf((float)DH);
  float qv[DH], acc[DH];
#pragma unroll
  for (int d = 0; d < DH; d += 4) {
    f32x4 t = *reinterpret_cast<const f32x4*>(qp + d);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      qv[d + e] = t[e] * scale;
      acc[d + e] = 0.f;
    }
  }
  float m = -INFINITY, l = 0.f;
  const int per = (nk + KS - 1) / KS;
  const int j0 = KS == 1 ? 0 : wave * per;
  const int j1 = KS == 1 ? nk : (j0 + per < nk ? j0 + per : nk);
  const long long kstep = k_is * ldkv;
  const float* kb = k + (g * k_gs) * ldkv + hd * DH;
  const float* vb = v + (g * k_gs) * ldkv + hd * DH;

  f32x4 rk[NLD], rv[NLD];
  auto fetch = [&](int jc) {  // keys jc .. jc+CH-1 of this wave's range -> registers
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int f = i * 64 + lane;
      const int key = f / Q4, c4 = f - key * Q4;
      rk[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      rv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (jc + key < j1) {
        rk[i] = *reinterpret_cast<const f32x4*>(kb + (long long)(jc + key) * kstep + c4 * 4);
        rv[i] = *reinterpret_cast<const f32x4*>(vb + (long long)(jc + key) * kstep + c4 * 4);
      }
    }
  };
  if (j0 < j1) fetch(j0);
#pragma unroll 1
  for (int jc = j0; jc < j1; jc += CH) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      *reinterpret_cast<f32x4*>(sk + (i * 64 + lane) * 4) = rk[i];
      *reinterpret_cast<f32x4*>(sv + (i * 64 + lane) * 4) = rv[i];
    }
    __builtin_amdgcn_wave_barrier();
    if (jc + CH < j1) fetch(jc + CH);  // next chunk in flight while this one is consumed
    const int nkeys = j1 - jc < CH ? j1 - jc : CH;
#pragma unroll 1
    for (int kk = 0; kk < nkeys; ++kk) {
      const f32x4* kr = reinterpret_cast<const f32x4*>(sk + kk * DH);
      const f32x4* vr = reinterpret_cast<const f32x4*>(sv + kk * DH);
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int d = 0; d < Q4; ++d) {
        const f32x4 t = kr[d];
        s0 = fmaf(qv[4 * d], t[0], s0);
        s1 = fmaf(qv[4 * d + 1], t[1], s1);
        s2 = fmaf(qv[4 * d + 2], t[2], s2);
        s3 = fmaf(qv[4 * d + 3], t[3], s3);
      }
      const float sc = (s0 + s1) + (s2 + s3);
      if (__ballot(sc > m)) {  // some lane has a new running maximum: rescale (rare after the first keys)
        const float mn = fmaxf(m, sc);
        const float corr = __expf(m - mn);
        l *= corr;
#pragma unroll
        for (int d = 0; d < DH; ++d) acc[d] *= corr;
        m = mn;
      }
      const float p = __expf(sc - m);
      l += p;
#pragma unroll
      for (int d = 0; d < Q4; ++d) {
        const f32x4 t = vr[d];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * d + e] = fmaf(p, t[e], acc[4 * d + e]);
      }
    }
  }
  if (KS > 1) {
    __syncthreads();  // every wave is done with its staging slice: the merge image may overlay it
    if (wave > 0) {
      float* r = lds + ((wave - 1) * 64 + lane) * (DH + 2);
      r[0] = m;
      r[1] = l;
#pragma unroll
      for (int d = 0; d < DH; ++d) r[2 + d] = acc[d];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll 1
    for (int w = 0; w < KS - 1; ++w) {
      const float* r = lds + (w * 64 + lane) * (DH + 2);
      const float mw = r[0];
      const float mn = fmaxf(m, mw);
      const float ca = (m == -INFINITY) ? 0.f : __expf(m - mn);
      const float cb = (mw == -INFINITY) ? 0.f : __expf(mw - mn);
      l = l * ca + r[1] * cb;
#pragma unroll
      for (int d = 0; d < DH; ++d) acc[d] = acc[d] * ca + r[2 + d] * cb;
      m = mn;
    }
  }
  if (active) {
    const float inv = 1.0f / l;
    float* op = o + qrow * ldo + hd * DH;
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
      f32x4 t;
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = acc[d + e] * inv;
      *reinterpret_cast<f32x4*>(op + d) = t;
    }
  }
}

}  // namespace

extern "C" int mvt_attention(const float* q, int ldq, long long q_gs, long long q_is, const float* k, const float* v, int ldkv,
                             long long k_gs, long long k_is, float* o, int ldo, int groups, int nq, int nk, int heads, int dh,
                             void* stream) {
  MVT_REQUIRE(q && k && v && o && groups > 0 && nq > 0 && nk > 0 && heads > 0);
  MVT_REQUIRE(ldq % 4 == 0 && ldkv % 4 == 0 && ldo % 4 == 0 && ldq >= heads * dh && ldkv >= heads * dh && ldo >= heads * dh);
  MVT_REQUIRE(((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) && ((uintptr_t)o % 16 == 0));
  const long long ntask = (long long)groups * heads * nq;
#define LAUNCH(KERN, DH, BLOCKS)                                                                                                 \
  hipLaunchKernelGGL((KERN<DH>), dim3((unsigned)(BLOCKS)), dim3(256), 0, mvt_stream(stream), q, ldq, q_gs, q_is, k, v, ldkv, k_gs, k_is, \
                     o, ldo, groups, nq, nk, heads)
  if (nq >= 64) {  // queries on lanes, keys wave-uniform
    const long long nchunk = (long long)groups * heads * ((nq + 63) / 64);
#define LAUNCH_U(DH, KS, BLOCKS, THREADS)                                                                                        \
  hipLaunchKernelGGL((attention_ukeys_kernel<DH, KS>), dim3((unsigned)(BLOCKS)), dim3(THREADS), 0, mvt_stream(stream), q, ldq, q_gs, \
                     q_is, k, v, ldkv, k_gs, k_is, o, ldo, groups, nq, nk, heads)
    if (nk > 64) {
      switch (dh) {
        case 32: LAUNCH_U(32, 8, nchunk, 512); break;
        case 48: LAUNCH_U(48, 8, nchunk, 512); break;
        case 64: LAUNCH_U(64, 8, nchunk, 512); break;
        default: return MVT_ERR_ARG;
      }
    } else {
      const long long blocks = mvt_cdiv(nchunk, 4);
      switch (dh) {
        case 32: LAUNCH_U(32, 1, blocks, 256); break;
        case 48: LAUNCH_U(48, 1, blocks, 256); break;
        case 64: LAUNCH_U(64, 1, blocks, 256); break;
        default: return MVT_ERR_ARG;
      }
    }
#undef LAUNCH_U
  } else if (nk <= 64) {
    const long long blocks = mvt_cdiv(ntask, 256);
    switch (dh) {
      case 32: LAUNCH(attention_qlane_kernel, 32, blocks); break;
      case 48: LAUNCH(attention_qlane_kernel, 48, blocks); break;
      case 64: LAUNCH(attention_qlane_kernel, 64, blocks); break;
      default: return MVT_ERR_ARG;
    }
  } else {
    const long long blocks = mvt_cdiv(ntask, 4);
    switch (dh) {
      case 32: LAUNCH(attention_klane_kernel, 32, blocks); break;
      case 48: LAUNCH(attention_klane_kernel, 48, blocks); break;
      case 64: LAUNCH(attention_klane_kernel, 64, blocks); break;
      default: return MVT_ERR_ARG;
    }
  }
#undef LAUNCH
  return mvt_launch_status();
}
