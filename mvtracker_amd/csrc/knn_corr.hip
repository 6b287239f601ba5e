// Exact kNN over the fused feature point cloud and the gather-dot correlation that consumes it
// (reference: knn / PointcloudCorrBlock.corr_sample, mvtracker.py:26-90, 800-846).
//
// knn_scan   - VALU-bound brute force.  A wave owns Q queries (wave-uniform registers) and
//              streams its candidate segment 64 points per step (one float4 per lane, 1 KiB
//              coalesced).  A candidate survives for query i when d2 <= thr_i (the current K-th
//              best); survivors are compacted into a per-query LDS list (ballot + mbcnt) and the
//              list is reduced to its K smallest (d2,index) keys whenever it could overflow, which
//              tightens thr_i.  After a warm-up almost every step is 7 VALU + 1 compare per query.
// corr_gather_dot - HBM-bound.  One wave per (track, frame): merge the per-segment key lists,
//              gather the K neighbour rows (C floats each, two rows per 1-KiB wave load, all loads
//              in flight together), dot them with the track feature by half-wave shuffle
//              reductions, append the neighbour offsets.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr unsigned long long KEY_MAX = ~0ULL;
constexpr int CAP = 128;  // per-query LDS list capacity (keys)

__device__ __forceinline__ float uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// Reduce the first `cnt` keys of `list` to the K smallest, sorted ascending into list[0..K-1].
// Returns the K-th smallest key (KEY_MAX if cnt < K).  One wave; cnt <= CAP.
__device__ __forceinline__ unsigned long long select_k(unsigned long long* list, int cnt, int K, int lane) {
  unsigned long long e0 = lane < cnt ? list[lane] : KEY_MAX;
  unsigned long long e1 = 64 + lane < cnt ? list[64 + lane] : KEY_MAX;
  unsigned long long mine = KEY_MAX, last = KEY_MAX;
  for (int r = 0; r < K; ++r) {
    unsigned long long m = wave_min_u64(e0 < e1 ? e0 : e1);
    if (e0 == m) e0 = KEY_MAX;  // indices are unique, so at most one live entry matches (KEY_MAX stays KEY_MAX)
    else if (e1 == m) e1 = KEY_MAX;
    if (lane == r) mine = m;
    last = m;
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < K) list[lane] = mine;
  __builtin_amdgcn_wave_barrier();
  return last;
}

// K-th smallest of the d2 bit patterns held two per lane (b0, b1; 0xFFFFFFFF = empty): radix select,
// one ballot + scalar popcount per bit, no cross-lane data movement.  d2 >= 0, so bit 31 is clear.
__device__ __forceinline__ unsigned kth_smallest_bits(unsigned b0, unsigned b1, int K) {
  unsigned prefix = 0;
  int need = K;
  bool c0 = b0 != 0xFFFFFFFFu, c1 = b1 != 0xFFFFFFFFu;  // still matching the prefix
#pragma unroll 1
  for (int bit = 30; bit >= 0; --bit) {
    const unsigned m = 1u << bit;
    const bool z0 = c0 && !(b0 & m), z1 = c1 && !(b1 & m);
    const int zeros = __popcll(__ballot(z0)) + __popcll(__ballot(z1));
    if (zeros >= need) {  // the K-th smallest has this bit clear
      c0 = z0;
      c1 = z1;
    } else {
      need -= zeros;
      prefix |= m;
      c0 = c0 && (b0 & m);
      c1 = c1 && (b1 & m);
    }
  }
  return prefix;
}

// Shrink list[0..cnt) (cnt <= CAP) to the entries with d2 <= (K-th smallest d2); returns the new count
// and that K-th smallest d2 through thr_bits.  Needs cnt >= K.
__device__ __forceinline__ int tighten(unsigned long long* list, int cnt, int K, int lane, unsigned long long lt_mask,
                                       unsigned* thr_bits) {
  const unsigned long long e0 = lane < cnt ? list[lane] : KEY_MAX;
  const unsigned long long e1 = 64 + lane < cnt ? list[64 + lane] : KEY_MAX;
  const unsigned b0 = (unsigned)(e0 >> 32), b1 = (unsigned)(e1 >> 32);
  const unsigned t = kth_smallest_bits(b0, b1, K);
  const bool k0 = b0 <= t, k1 = b1 <= t && b1 != 0xFFFFFFFFu;
  const unsigned long long m0 = __ballot(k0 && b0 != 0xFFFFFFFFu), m1 = __ballot(k1);
  const int n0 = __popcll(m0);
  __builtin_amdgcn_wave_barrier();
  if (k0 && b0 != 0xFFFFFFFFu) list[__popcll(m0 & lt_mask)] = e0;
  if (k1) list[n0 + __popcll(m1 & lt_mask)] = e1;
  __builtin_amdgcn_wave_barrier();
  *thr_bits = t;
  return n0 + __popcll(m1);
}

// Sort list[0..cnt) (cnt <= 64) ascending by the full (d2, index) key and return, in lane r < K, the r-th key.
__device__ __forceinline__ unsigned long long rank_sort(const unsigned long long* list, int cnt, int K, int lane) {
  const unsigned long long e = lane < cnt ? list[lane] : KEY_MAX;
  const unsigned lo = (unsigned)e, hi = (unsigned)(e >> 32);
  int rank = 0;
#pragma unroll 1
  for (int j = 0; j < cnt; ++j) {
    const unsigned long long ej = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, j) << 32) |
                                  (unsigned)__builtin_amdgcn_readlane((int)lo, j);
    rank += ej < e ? 1 : 0;
  }
  // lane r receives the key whose rank is r: ranks of the cnt live lanes are a permutation of 0..cnt-1 (keys are unique), so the
  // forward permute (every lane pushes its key to lane `rank`) has one writer per destination below cnt; the idle lanes (rank =
  // cnt, key KEY_MAX) all push to lane cnt, harmlessly.  Two ds_permute instead of a 64-iteration readlane loop.
  const unsigned plo = (unsigned)__builtin_amdgcn_ds_permute(rank * 4, (int)lo), phi = (unsigned)__builtin_amdgcn_ds_permute(rank * 4, (int)hi);
  return lane < cnt ? (((unsigned long long)phi << 32) | plo) : KEY_MAX;
}

// Candidate tiles.  The scan visits the cloud 64 points (one wave load) at a time:
//   linear tiles (grid_w == 0): tile t = points [64 t, 64 t + 64);
//   patch tiles  (grid_w  > 0): the cloud is V images of grid_h x grid_w points in raster order (both multiples of 8) and
//                               tile t is an 8x8 pixel patch - eight 128-byte row pieces - whose bounding box is compact.
// (32-bit arithmetic: P < 2^31 is required by every caller, and a 64-bit division per visited tile cost more than the tile's compares)
__device__ __forceinline__ int tile_point(int tile, int lane, int grid_w, int grid_h) {
  if (grid_w == 0) return tile * 64 + lane;
  const unsigned tpr = (unsigned)grid_w >> 3, tpv = tpr * ((unsigned)grid_h >> 3);
  const unsigned v = (unsigned)tile / tpv;
  const unsigned r = (unsigned)tile - v * tpv;
  const unsigned ty = r / tpr, tx = r - ty * tpr;
  return (int)((v * grid_h + ty * 8 + (lane >> 3)) * grid_w + tx * 8 + (lane & 7));
}

// box[frame][tile] = {lo.xyz, number of finite points, hi.xyz, 0}; NaN points are ignored (fminf / fmaxf drop them), an all-NaN tile gets lo = +inf, hi = -inf
__global__ __launch_bounds__(256) void tile_aabb_kernel(const float* __restrict__ xyz, long long P, long long ntiles, long long total,
                                                        int grid_w, int grid_h, float* __restrict__ box) {
  const int lane = threadIdx.x & 63;
  const long long id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);  // frame * ntiles + tile
  if (id >= total) return;
  const long long frame = id / ntiles, tile = id - frame * ntiles;
  const long long c = tile_point((int)tile, lane, grid_w, grid_h);
  const float inf = __int_as_float(0x7f800000);
  float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
  bool finite = false;
  if (c < P) {
    const f32x4 p = *reinterpret_cast<const f32x4*>(xyz + (frame * P + c) * 4);
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      if (p[e] == p[e]) lo[e] = hi[e] = p[e];
    }
    finite = fabsf(p[0]) < inf && fabsf(p[1]) < inf && fabsf(p[2]) < inf;
  }
  const int nfinite = __popcll(__ballot(finite));
#pragma unroll
  for (int e = 0; e < 3; ++e) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo[e] = fminf(lo[e], __shfl_xor(lo[e], o, 64));
      hi[e] = fmaxf(hi[e], __shfl_xor(hi[e], o, 64));
    }
  }
  if (lane == 0) {
    float* b = box + id * 8;
    *reinterpret_cast<f32x4*>(b) = (f32x4){lo[0], lo[1], lo[2], (float)nfinite};  // .w: points with finite coordinates
    *reinterpret_cast<f32x4*>(b + 4) = (f32x4){hi[0], hi[1], hi[2], 0.f};
  }
}

// gbox[frame][g] = the union of the boxes of tiles [64 g, 64 g + 64) (same record layout; .w = their finite points): the coarse
// level of the culling hierarchy -- a single-segment search tests these first and skips whole runs of 64 tiles.
__global__ __launch_bounds__(256) void tile_group_aabb_kernel(const float* __restrict__ box, long long ntiles, long long ngroups,
                                                              long long total, float* __restrict__ gbox) {
  const int lane = threadIdx.x & 63;
  const long long id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);  // frame * ngroups + group
  if (id >= total) return;
  const long long frame = id / ngroups, g = id - frame * ngroups;
  const long long t = g * 64 + lane;
  const float inf = __int_as_float(0x7f800000);
  float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf}, cnt = 0.f;
  if (t < ntiles) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(box + (frame * ntiles + t) * 8), b = *reinterpret_cast<const f32x4*>(box + (frame * ntiles + t) * 8 + 4);
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      lo[e] = a[e];
      hi[e] = b[e];
    }
    cnt = a[3];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      lo[e] = fminf(lo[e], __shfl_xor(lo[e], o, 64));
      hi[e] = fmaxf(hi[e], __shfl_xor(hi[e], o, 64));
    }
    cnt += __shfl_xor(cnt, o, 64);
  }
  if (lane == 0) {
    float* b = gbox + id * 8;
    *reinterpret_cast<f32x4*>(b) = (f32x4){lo[0], lo[1], lo[2], cnt};
    *reinterpret_cast<f32x4*>(b + 4) = (f32x4){hi[0], hi[1], hi[2], 0.f};
  }
}

template <int Q>
__device__ __forceinline__ void knn_scan_body(unsigned long long* lds, const float* __restrict__ xyz, long long P,
                                              const float* __restrict__ coords, int N, int S, int frame0, int frame_step, int T, int K,
                                              int nseg, unsigned long long* __restrict__ keys, int qgroups,
                                              const int* __restrict__ seed_idx, int seed_k, int seed_cw, int seed_ch, int seed_fw,
                                              int seed_fh, const float* __restrict__ box, int grid_w, int grid_h,
                                              int* __restrict__ idx_direct = nullptr, const float* __restrict__ gbox = nullptr,
                                              bool xcd_frames = false) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // task id -> (segment, query group, slot); segment fastest so that heavy frames spread over CUs  (32-bit: the launchers bound it)
  // xcd_frames (the seeded all-level search, every frame equally heavy): tasks are slot-major, so a contiguous run of blocks is a
  // run of whole frames -- XCD c (the hardware deals consecutive block ids round-robin to the 8 XCDs) takes the run [c, c+1) *
  // gridDim.x / 8 and keeps ITS 1.5 frames' point clouds and tile boxes (1.3 MB per frame over the four levels) in its own 4 MB
  // L2, instead of all 8 XCDs cycling all S frames (16 MB at S = 12) through theirs.
  unsigned bxr = blockIdx.x;
#ifndef MVT_KNN_NO_XCD  // (A/B builds)
  if (xcd_frames && gridDim.x % 8 == 0) bxr = (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
#endif
  const unsigned task = bxr * 4u + (unsigned)wave;
  const unsigned ntask = (unsigned)qgroups * (unsigned)S * (unsigned)nseg;
  if (task >= ntask) return;
  const unsigned tq = task / (unsigned)nseg;
  const int seg = (int)(task - tq * (unsigned)nseg);
  const int s = (int)(tq / (unsigned)qgroups);
  const int qg = (int)(tq - (unsigned)s * (unsigned)qgroups);
  int frame = frame0 + s * frame_step;
  frame = frame < T - 1 ? frame : T - 1;
  const float* cand = xyz + (long long)frame * P * 4;
  const int ntiles = (int)((P + 63) >> 6);
  const int tper = (ntiles + nseg - 1) / nseg;
  const int t0 = seg * tper;
  const int t1 = t0 + tper < ntiles ? t0 + tper : ntiles;
  const float* fbox = box ? box + (long long)frame * ntiles * 8 : nullptr;

  unsigned long long* list = lds + (long long)wave * Q * CAP;
  float qx[Q], qy[Q], qz[Q], thr[Q];
  int cnt[Q];
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    const int n = qg * Q + i;
    const float* c = coords + ((long long)(n < N ? n : N - 1) * S + s) * 3;
    qx[i] = uniform_f(c[0]);
    qy[i] = uniform_f(c[1]);
    qz[i] = uniform_f(c[2]);
    thr[i] = __int_as_float(0x7f800000);  // +inf
    cnt[i] = 0;
    if (seed_idx) {
      // Any seed_k >= K distinct points of this frame bound the K-th nearest distance from above: start with
      // thr = max d2 over the seeds (the previous iteration's neighbours, or the coarser level's neighbours mapped
      // to this level's grid), so that almost no candidate survives the compare.  Exactness is unaffected.
      float d2s = 0.f;
      if (lane < seed_k) {
        int si = seed_idx[((long long)(n < N ? n : N - 1) * S + s) * seed_k + lane];
        if (seed_cw > 0) {  // coarse (v, y, x) -> fine (v, 2y, 2x)
          const int pv = seed_cw * seed_ch;
          const int v = si / pv, rem = si - v * pv;
          const int y = rem / seed_cw, x = rem - y * seed_cw;
          si = (v * seed_fh + 2 * y) * seed_fw + 2 * x;
        }
        si = si < 0 ? 0 : (si < P ? si : (int)(P - 1));
        const f32x4 sp = *reinterpret_cast<const f32x4*>(cand + (long long)si * 4);
        const float dx = sp[0] - qx[i], dy = sp[1] - qy[i], dz = sp[2] - qz[i];
        d2s = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
        if (!(d2s == d2s)) d2s = __int_as_float(0x7f800000);  // NaN coordinates: no bound
      }
      thr[i] = uniform_f(wave_max(d2s));
    }
  }
  if (fbox && !seed_idx) {
    // Unseeded scan: a tile with >= K finite points puts the K-th nearest neighbour within the FARTHEST corner of its box
    // (per-axis max gap, same monotonic arithmetic as the scan, so the bound is >= the d2 of every point of the tile).
    // The minimum over this wave's tiles is an exact initial threshold; without it the first 64 tiles are scanned in full.
    float ub[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) ub[i] = __int_as_float(0x7f800000);
    for (int tb = t0; tb < t1; tb += 64) {
      const int mt = tb + lane;
      if (mt < t1) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(fbox + (long long)mt * 8), hi = *reinterpret_cast<const f32x4*>(fbox + (long long)mt * 8 + 4);
        if (lo[3] >= (float)K) {
#pragma unroll
          for (int i = 0; i < Q; ++i) {
            const float dx = fmaxf(fabsf(lo[0] - qx[i]), fabsf(hi[0] - qx[i]));
            const float dy = fmaxf(fabsf(lo[1] - qy[i]), fabsf(hi[1] - qy[i]));
            const float dz = fmaxf(fabsf(lo[2] - qz[i]), fabsf(hi[2] - qz[i]));
            const float u = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
            ub[i] = u < ub[i] ? u : ub[i];  // (a NaN bound -- NaN query -- never replaces +inf)
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      float u = ub[i];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) u = fminf(u, __shfl_xor(u, o, 64));
      thr[i] = uniform_f(u);
    }
  }
  const unsigned long long lt_mask = (1ULL << lane) - 1ULL;
  const float qnan = __int_as_float(0x7fc00000);  // lanes past the cloud end carry NaN points: every compare fails

  auto load_tile = [&](int tile, int& c) {
    c = tile_point(tile, lane, grid_w, grid_h);
    f32x4 p = (f32x4){qnan, qnan, qnan, 0.f};
    if (c < P) p = *reinterpret_cast<const f32x4*>(cand + (long long)c * 4);
    return p;
  };

  // Coarse level (single segment, <= 64 groups of 64 tiles): one group box per lane, same bound arithmetic -- a group box contains
  // its tiles' boxes, so its bound is <= theirs and a culled group cannot hold a survivor either.
  unsigned long long gmask = ~0ULL;
  const bool use_groups = gbox && fbox && nseg == 1 && ntiles > 64 && ntiles <= 64 * 64;
  if (use_groups) {
    const int ng = (ntiles + 63) >> 6;
    bool near = false;
    if (lane < ng) {
      const float* gb = gbox + ((long long)frame * ng + lane) * 8;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(gb), hi = *reinterpret_cast<const f32x4*>(gb + 4);
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        const float dx = fmaxf(fmaxf(lo[0] - qx[i], qx[i] - hi[0]), 0.f);
        const float dy = fmaxf(fmaxf(lo[1] - qy[i], qy[i] - hi[1]), 0.f);
        const float dz = fmaxf(fmaxf(lo[2] - qz[i], qz[i] - hi[2]), 0.f);
        const float lb = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
        near = near || !(lb > thr[i]);
      }
    }
    gmask = __ballot(near);
  }

  for (int tb = t0; tb < t1; tb += 64) {
    if (use_groups && !((gmask >> (tb >> 6)) & 1ULL)) continue;  // (t0 = 0 and tb < 4096 whenever the mask is in use: the shift stays below 64)
    // which of the next 64 tiles can hold a point within thr of any of the Q queries?  One tile per lane.  The bound
    // uses the distance arithmetic of the scan itself on the per-axis gaps to the box, and every rounding step is
    // monotonic, so bound <= d2 of every point in the box in floating point: a culled tile cannot hold a survivor.
    const int mt = tb + lane;
    bool visit = mt < t1;
    if (fbox && visit) {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(fbox + (long long)mt * 8), hi = *reinterpret_cast<const f32x4*>(fbox + (long long)mt * 8 + 4);
      bool near = false;
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        const float dx = fmaxf(fmaxf(lo[0] - qx[i], qx[i] - hi[0]), 0.f);
        const float dy = fmaxf(fmaxf(lo[1] - qy[i], qy[i] - hi[1]), 0.f);
        const float dz = fmaxf(fmaxf(lo[2] - qz[i], qz[i] - hi[2]), 0.f);
        const float lb = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
        near = near || !(lb > thr[i]);  // NaN bound (NaN query): never cull
      }
      visit = near;
    }
    unsigned long long todo = __ballot(visit);
    if (!todo) continue;
    int c, cn = 0;
    f32x4 p = load_tile(tb + __builtin_ctzll(todo), c);
    while (todo) {
      todo &= todo - 1;
      f32x4 pn = p;
      if (todo) pn = load_tile(tb + __builtin_ctzll(todo), cn);  // prefetch the next surviving tile
      float d2[Q];
      unsigned long long m[Q];
      unsigned long long any = 0;
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        const float dx = p[0] - qx[i], dy = p[1] - qy[i], dz = p[2] - qz[i];
        d2[i] = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
        m[i] = __ballot(d2[i] <= thr[i]);
        any |= m[i];
      }
      if (any) {  // rare after the warm-up: some query has a survivor in this step
#pragma unroll
        for (int i = 0; i < Q; ++i) {
          if (m[i]) {
            unsigned long long* l = list + i * CAP;
            if ((m[i] >> lane) & 1ULL) l[cnt[i] + __popcll(m[i] & lt_mask)] = ((unsigned long long)__float_as_uint(d2[i]) << 32) | (unsigned)c;
            cnt[i] += __popcll(m[i]);
            if (cnt[i] > CAP - 64) {
              unsigned tbits;
              int nc = tighten(l, cnt[i], K, lane, lt_mask, &tbits);
              if (nc > CAP - 64) {  // > 48 exact distance ties at the threshold: resolve them by index now
                select_k(l, nc, K, lane);
                nc = K;
              }
              cnt[i] = nc;
              thr[i] = __uint_as_float(tbits);
            }
          }
        }
      }
      p = pn;
      c = cn;
    }
  }
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    const int n = qg * Q + i;
    unsigned long long* l = list + i * CAP;
    __builtin_amdgcn_wave_barrier();
    int nc = cnt[i];
    if (nc > 64) {  // (up to 64 entries the rank sort below picks the K smallest directly: cheaper than the radix select)
      unsigned tbits;
      nc = tighten(l, nc, K, lane, lt_mask, &tbits);
    }
    unsigned long long v;
    if (nc <= 64) {
      v = rank_sort(l, nc, K, lane);
    } else {
      select_k(l, nc, K, lane);
      v = lane < K ? l[lane] : KEY_MAX;
    }
    if (n < N && lane < K) {
      if (idx_direct) {  // single segment: the sorted list IS the result (mvt_knn_merge's index extraction and clamp)
        unsigned id = (unsigned)v;
        if ((long long)id >= P) id = (unsigned)(P - 1);
        idx_direct[((long long)n * S + s) * K + lane] = (int)id;
      } else {
        keys[(((long long)n * S + s) * nseg + seg) * K + lane] = v;
      }
    }
  }
}

template <int Q>
__global__ __launch_bounds__(256) void knn_scan_kernel(const float* __restrict__ xyz, long long P, const float* __restrict__ coords,
                                                       int N, int S, int frame0, int frame_step, int T, int K, int nseg,
                                                       unsigned long long* __restrict__ keys, int qgroups,
                                                       const int* __restrict__ seed_idx, int seed_k, int seed_cw, int seed_ch,
                                                       int seed_fw, int seed_fh, const float* __restrict__ box, int grid_w, int grid_h,
                                                       int* __restrict__ idx_direct, const float* __restrict__ gbox) {
  __shared__ unsigned long long lds[4 * Q * CAP];
  knn_scan_body<Q>(lds, xyz, P, coords, N, S, frame0, frame_step, T, K, nseg, keys, qgroups, seed_idx, seed_k, seed_cw, seed_ch, seed_fw,
                   seed_fh, box, grid_w, grid_h, idx_direct, gbox);
}

// All pyramid levels of one refinement iteration in ONE launch (grid.y = level): after the first iteration every level is
// seeded by its own previous neighbours, so the four scans are independent and each alone is launch / tail latency.
struct KnnLevels {
  mvt_knn_level lv[8];
};

template <int Q>
__global__ __launch_bounds__(256) void knn_scan_levels_kernel(KnnLevels a, const float* __restrict__ coords, int N, int S, int frame0,
                                                              int frame_step, int T, int K, int qgroups, int seed_k) {
  __shared__ unsigned long long lds[4 * Q * CAP];
  const mvt_knn_level L = a.lv[blockIdx.y];
  knn_scan_body<Q>(lds, L.xyz, L.P, coords, N, S, frame0, frame_step, T, K, L.nseg, L.keys, qgroups, L.seed_idx, seed_k, 0, 0, 0, 0,
                   L.tile_box, L.grid_w, L.grid_h);
}

// Seeded search of all levels, single segment, neighbour indices written directly (no key lists, no merge launch).  A wave reads
// the seeds of its own (track, slot) entries before it writes them, so idx_out may alias seed_idx.
template <int Q>
__global__ __launch_bounds__(256) void knn_search_levels_kernel(KnnLevels a, const float* __restrict__ coords, int N, int S, int frame0,
                                                                int frame_step, int T, int K, int qgroups, int seed_k) {
  __shared__ unsigned long long lds[4 * Q * CAP];
  const mvt_knn_level L = a.lv[blockIdx.y];
  knn_scan_body<Q>(lds, L.xyz, L.P, coords, N, S, frame0, frame_step, T, K, 1, nullptr, qgroups, L.seed_idx, seed_k, 0, 0, 0, 0, L.tile_box,
                   L.grid_w, L.grid_h, L.idx_out, L.group_box, true);
}

// Merge the nseg per-segment lists of one (track, slot) into its K nearest neighbour indices.  Every lane holds
// one key; its rank among the E = nseg*K keys is counted with wave-uniform readlane broadcasts (keys are unique).
__device__ __forceinline__ void knn_merge_body(const unsigned long long* __restrict__ keys, long long rows, int K, int nseg, long long P,
                                               int* __restrict__ idx_out) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int E = nseg * K;
  const unsigned long long key = lane < E ? keys[row * E + lane] : KEY_MAX;
  int rank = lane;
  if (nseg > 1) {
    const unsigned lo = (unsigned)key, hi = (unsigned)(key >> 32);
    rank = 0;
#pragma unroll 1
    for (int j = 0; j < E; ++j) {
      const unsigned long long kj = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, j) << 32) |
                                    (unsigned)__builtin_amdgcn_readlane((int)lo, j);
      rank += (kj < key || (kj == key && j < lane)) ? 1 : 0;  // KEY_MAX padding ties are broken by lane
    }
  }
  if (lane < E && rank < K) {
    unsigned idx = (unsigned)key;
    if ((long long)idx >= P) idx = (unsigned)(P - 1);  // only reachable with NaN coordinates; stay in bounds
    idx_out[row * K + rank] = (int)idx;
  }
}

__global__ __launch_bounds__(256) void knn_merge_kernel(const unsigned long long* __restrict__ keys, long long rows, int K, int nseg,
                                                        long long P, int* __restrict__ idx_out) {
  knn_merge_body(keys, rows, K, nseg, P, idx_out);
}

__global__ __launch_bounds__(256) void knn_merge_levels_kernel(KnnLevels a, long long rows, int K) {
  const mvt_knn_level L = a.lv[blockIdx.y];
  knn_merge_body(L.keys, rows, K, L.nseg, L.P, L.idx_out);
}

struct CorrLevels {
  const float* xyz[8];
  const float* fvec[8];
  const int* idx[8];
  long long P[8];
};

// One wave per (track, frame) unit and level: gather the K neighbour rows, dot with the target, neighbour offsets.
// LPR lanes cover one feature row with 16 bytes each: C = 4 * LPR fp32 values, or (BF) C = 8 * LPR bf16 values -- the bf16
// frame store of bf16 mode, 256-B rows at C = 128: four rows per 1-KiB wave load, fp32 dot accumulation.
template <int LPR, int BF>
__global__ __launch_bounds__(256) void corr_gather_dot_kernel(CorrLevels lv, const float* __restrict__ targets,
                                                              const float* __restrict__ coords, int N, int S, int frame0,
                                                              int frame_step, int T, int K, float* __restrict__ out, int ldo, int o_off) {
  constexpr int EPL = BF ? 8 : 4;  // elements per lane
  constexpr int C = EPL * LPR;
  constexpr int RPL = 64 / LPR;  // rows per wave load
  const int lane = threadIdx.x & 63;
  const int level = blockIdx.y;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);  // n * S + s
  if (row >= (long long)N * S) return;
  const int s = (int)(row % S);
  int frame = frame0 + s * frame_step;
  frame = frame < T - 1 ? frame : T - 1;
  const long long P = lv.P[level];
  const float* __restrict__ xyz = lv.xyz[level];
  const float* __restrict__ fvec = lv.fvec[level];

  unsigned idx = (unsigned)lv.idx[level][row * K + (lane < K ? lane : 0)];
  if ((long long)idx >= P) idx = (unsigned)(P - 1);

  const int sub = lane / LPR, cq = lane % LPR;
  f32x4 tg[EPL / 4];
#pragma unroll
  for (int e = 0; e < EPL / 4; ++e) tg[e] = *reinterpret_cast<const f32x4*>(targets + row * C + cq * EPL + 4 * e);
  const long long fbase = (long long)frame * P * C + cq * EPL;  // element offset into the level's rows
  constexpr int MAXJ = (16 + RPL - 1) / RPL;
  f32x4 f[MAXJ][EPL / 4];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int k = j * RPL + sub;
    const unsigned ik = __shfl(idx, k < K ? k : 0, 64);
    if (BF) {
      const uint4 w = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(fvec) + fbase + (long long)ik * C);
      f[j][0] = (f32x4){__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xFFFF0000u), __uint_as_float(w.y << 16),
                        __uint_as_float(w.y & 0xFFFF0000u)};
      f[j][EPL / 4 - 1] = (f32x4){__uint_as_float(w.z << 16), __uint_as_float(w.z & 0xFFFF0000u), __uint_as_float(w.w << 16),
                                  __uint_as_float(w.w & 0xFFFF0000u)};
    } else {
      f[j][0] = *reinterpret_cast<const f32x4*>(fvec + fbase + (long long)ik * C);
    }
  }
  // neighbour offsets (lane k < K)
  const f32x4 nx = *reinterpret_cast<const f32x4*>(xyz + ((long long)frame * P + idx) * 4);
  const float* cw = coords + row * 3;
  const float ox = nx[0] - cw[0], oy = nx[1] - cw[1], oz = nx[2] - cw[2];

  const float scale = sqrtf((float)C);
  float mine = 0.f;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    float d = tg[0][0] * f[j][0][0];
    d = fmaf(tg[0][1], f[j][0][1], d);
    d = fmaf(tg[0][2], f[j][0][2], d);
    d = fmaf(tg[0][3], f[j][0][3], d);
    if (BF) {
#pragma unroll
      for (int e = 0; e < 4; ++e) d = fmaf(tg[EPL / 4 - 1][e], f[j][EPL / 4 - 1][e], d);
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    // lane k takes the value of its (j, sub) = (k / RPL, k % RPL)
    const float v = __shfl(d, (lane % RPL) * LPR, 64);
    if (lane / RPL == j) mine = v;
  }
  if (lane < K) {
    float* o = out + row * ldo + o_off + level * K * 4 + lane * 4;
    if ((((uintptr_t)out | (unsigned)(ldo * 4) | (unsigned)(o_off * 4)) & 15) == 0) {  // (uniform: the tracker's layout is aligned)
      *reinterpret_cast<f32x4*>(o) = (f32x4){mine / scale, ox, oy, oz};
    } else {
      o[0] = mine / scale;
      o[1] = ox;
      o[2] = oy;
      o[3] = oz;
    }
  }
}

// The same operator with the reference's non-default options (mvtracker.py:832-846): ``groups`` grouped dots per neighbour (each over
// C / groups channels, / sqrt(C / groups)), the neighbour offset and / or the neighbour's own coordinates appended -- OW = groups +
// 3 * add_offset + 3 * add_xyz outputs per neighbour.  Same loads as the shipped-layout kernel above; the group sums stop the shuffle
// reduction at the group's lanes and the group leaders store their value themselves (4-byte stores: a generality path, not the
// benchmarked one).
template <int LPR, int BF>
__global__ __launch_bounds__(256) void corr_gather_dot_opts_kernel(CorrLevels lv, const float* __restrict__ targets,
                                                                   const float* __restrict__ coords, int N, int S, int frame0,
                                                                   int frame_step, int T, int K, float* __restrict__ out, int ldo, int o_off,
                                                                   int G, int add_offset, int add_xyz) {
  constexpr int EPL = BF ? 8 : 4;
  constexpr int C = EPL * LPR;
  constexpr int RPL = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int level = blockIdx.y;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);  // n * S + s
  if (row >= (long long)N * S) return;
  const int s = (int)(row % S);
  int frame = frame0 + s * frame_step;
  frame = frame < T - 1 ? frame : T - 1;
  const long long P = lv.P[level];
  const float* __restrict__ xyz = lv.xyz[level];
  const float* __restrict__ fvec = lv.fvec[level];
  unsigned idx = (unsigned)lv.idx[level][row * K + (lane < K ? lane : 0)];
  if ((long long)idx >= P) idx = (unsigned)(P - 1);
  const int sub = lane / LPR, cq = lane % LPR;
  f32x4 tg[EPL / 4];
#pragma unroll
  for (int e = 0; e < EPL / 4; ++e) tg[e] = *reinterpret_cast<const f32x4*>(targets + row * C + cq * EPL + 4 * e);
  const long long fbase = (long long)frame * P * C + cq * EPL;
  constexpr int MAXJ = (16 + RPL - 1) / RPL;
  const int OW = G + 3 * add_offset + 3 * add_xyz;
  const int lpg = LPR / G;  // lanes per group
  const float scale = sqrtf((float)(C / G));
  float* orow = out + row * ldo + o_off + (long long)level * K * OW;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int k = j * RPL + sub;
    const unsigned ik = __shfl(idx, k < K ? k : 0, 64);
    f32x4 f0, f1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (BF) {
      const uint4 w = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(fvec) + fbase + (long long)ik * C);
      f0 = (f32x4){__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xFFFF0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xFFFF0000u)};
      f1 = (f32x4){__uint_as_float(w.z << 16), __uint_as_float(w.z & 0xFFFF0000u), __uint_as_float(w.w << 16), __uint_as_float(w.w & 0xFFFF0000u)};
    } else {
      f0 = *reinterpret_cast<const f32x4*>(fvec + fbase + (long long)ik * C);
    }
    float d = tg[0][0] * f0[0];
    d = fmaf(tg[0][1], f0[1], d);
    d = fmaf(tg[0][2], f0[2], d);
    d = fmaf(tg[0][3], f0[3], d);
    if (BF) {
#pragma unroll
      for (int e = 0; e < 4; ++e) d = fmaf(tg[EPL / 4 - 1][e], f1[e], d);
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) {
      const float other = __shfl_xor(d, o, 64);
      if (o < lpg) d += other;  // (wave-uniform: the reduction stops at the group's lanes)
    }
    if (k < K && cq % lpg == 0) orow[k * OW + cq / lpg] = d / scale;
  }
  if (lane < K && (add_offset || add_xyz)) {
    const f32x4 nx = *reinterpret_cast<const f32x4*>(xyz + ((long long)frame * P + idx) * 4);
    const float* cw = coords + row * 3;
    float* o = orow + lane * OW + G;
    if (add_offset) {
      o[0] = nx[0] - cw[0];
      o[1] = nx[1] - cw[1];
      o[2] = nx[2] - cw[2];
      o += 3;
    }
    if (add_xyz) {
      o[0] = nx[0];
      o[1] = nx[1];
      o[2] = nx[2];
    }
  }
}

template <int BF>
__global__ __launch_bounds__(256) void knn1_gather_kernel(const float* __restrict__ fvec, long long P, int C,
                                                          const unsigned long long* __restrict__ keys, int n, int nseg, int frame,
                                                          float* __restrict__ feat_out, int* __restrict__ idx_out) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= n) return;
  unsigned long long key = lane < nseg ? keys[(long long)q * nseg + lane] : KEY_MAX;
  key = wave_min_u64(key);
  unsigned idx = (unsigned)key;
  if ((long long)idx >= P) idx = (unsigned)(P - 1);
  if (idx_out && lane == 0) idx_out[q] = (int)idx;
  const long long src = ((long long)frame * P + idx) * C;
  for (int c = lane * 4; c < C; c += 256) *reinterpret_cast<f32x4*>(feat_out + (long long)q * C + c) = load_act4(fvec, src + c, BF);
}

// ------------------------------------------------------------------------------------------------
// Secondary operator: (2r+1)^2 bilinear window correlation, one wave per (frame, track).
// All window samples share one fractional offset, so the wave first dots the (2r+2)^2 texel
// patch with the target (two texel rows per 1-KiB load, half-wave shuffle reduction, results
// staged in LDS) and then blends four neighbouring dots per output.
template <int LPR>
__global__ __launch_bounds__(256) void window_corr_kernel(const float* __restrict__ fmap, const float* __restrict__ targets,
                                                          const float* __restrict__ coords, float* __restrict__ out, int BS, int N,
                                                          int h, int w, int level, int radius, int ldo, int o_off) {
  constexpr int C = 4 * LPR;
  constexpr int RPL = 64 / LPR;
  __shared__ float dots[4][16 * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long row = (long long)blockIdx.x * 4 + wave;  // bs * N + n
  if (row >= (long long)BS * N) return;
  const long long bs = row / N;
  const int side = 2 * radius + 2;
  const float inv = 1.0f / (float)(1 << level);
  const float cx = coords[row * 2] * inv, cy = coords[row * 2 + 1] * inv;
  const float fx0 = floorf(cx), fy0 = floorf(cy);
  const int x0 = (int)fx0 - radius, y0 = (int)fy0 - radius;
  const float ax = cx - fx0, ay = cy - fy0;
  const int sub = lane / LPR, cq = lane % LPR;
  const f32x4 tg = *reinterpret_cast<const f32x4*>(targets + row * C + cq * 4);
  const float* fb = fmap + bs * (long long)h * w * C + cq * 4;
  const int ntex = side * side;
  for (int t0 = 0; t0 < ntex; t0 += RPL) {
    const int tt = t0 + sub;
    const int ty = tt / side, tx = tt - ty * side;  // patch[ty][tx] = texel (x0 + tx, y0 + ty)
    const int gx = x0 + tx, gy = y0 + ty;
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (tt < ntex && (unsigned)gx < (unsigned)w && (unsigned)gy < (unsigned)h)
      v = *reinterpret_cast<const f32x4*>(fb + ((long long)gy * w + gx) * C);
    float d = tg[0] * v[0];
    d = fmaf(tg[1], v[1], d);
    d = fmaf(tg[2], v[2], d);
    d = fmaf(tg[3], v[3], d);
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    if (cq == 0 && tt < ntex) dots[wave][ty * 16 + tx] = d;
  }
  __builtin_amdgcn_wave_barrier();
  const int win = 2 * radius + 1;
  const float scale = sqrtf((float)C);
  for (int o = lane; o < win * win; o += 64) {
    const int a = o / win, b = o - a * win;  // sample at (cx + a - r, cy + b - r): first index runs along x
    const float d00 = dots[wave][b * 16 + a], d01 = dots[wave][b * 16 + a + 1];
    const float d10 = dots[wave][(b + 1) * 16 + a], d11 = dots[wave][(b + 1) * 16 + a + 1];
    const float v = (1.f - ay) * ((1.f - ax) * d00 + ax * d01) + ay * ((1.f - ax) * d10 + ax * d11);
    out[row * ldo + o_off + o] = v / scale;
  }
}


// The same operator for ALL pyramid levels in one launch (grid.y = level) and for bf16 maps (BF: the feature maps of bf16 mode --
// under autocast the reference's CorrBlock holds bf16 fmaps, blocks.py:423-449 -- 8 channels per 16-byte lane piece, so a 1-KiB
// wave load covers 64 / LPR texels).  U wave loads are in flight before the first dot (a load -> dot -> shuffle -> LDS loop with a
// run-time trip count pays one memory round trip per iteration).  fp32 accumulation; dots are blended exactly as above.
struct WinLevels {
  const void* fmap[8];
  int h[8], w[8];
};

template <int LPR, int BF>
__global__ __launch_bounds__(256) void window_corr_levels_kernel(WinLevels lv, const float* __restrict__ targets,
                                                                 const float* __restrict__ coords, float* __restrict__ out, int BS, int N,
                                                                 int radius, int ldo, int o_off) {
  constexpr int EPL = BF ? 8 : 4;
  constexpr int C = EPL * LPR;
  constexpr int RPL = 64 / LPR;
  constexpr int U = 4;
  __shared__ float dots[4][16 * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int level = blockIdx.y;
  // XCD-aware unit order (cdna_hip_programming.md T1): consecutive workgroup ids go round-robin to the 8 XCDs, each with its own
  // 4 MiB L2.  All windows of one frame read the same map (4 MiB at level 0 in bf16), so every XCD is given WHOLE frames -- frame f
  // to XCD f % 8 -- and only the S % 8 leftover frames are shared, each by 8 / (S % 8) XCDs.  In plain row order every XCD touched
  // every frame's map and the memory side saw 5x the unique bytes (profiles/r03_window_corr.json).
  long long wg = blockIdx.x;
  {
    const long long per_frame = N / 4, total = (long long)BS * per_frame;
    const int rem = BS % 8;
    if (N % 4 == 0 && gridDim.x == total && total % 8 == 0 && (rem * per_frame) % 8 == 0) {
      const long long xcd = wg % 8, j = wg / 8, whole = (long long)(BS / 8) * per_frame;
      if (j < whole) {
        wg = ((j / per_frame) * 8 + xcd) * per_frame + j % per_frame;
      } else {
        const long long u = xcd * (rem * per_frame / 8) + (j - whole);
        wg = ((long long)(BS - rem) + u / per_frame) * per_frame + u % per_frame;
      }
    }
  }
  const long long row = wg * 4 + wave;  // bs * N + n
  if (row >= (long long)BS * N) return;
  const long long bs = row / N;
  const int h = lv.h[level], w = lv.w[level];
  const int side = 2 * radius + 2;
  const float inv = 1.0f / (float)(1 << level);
  const float cx = coords[row * 2] * inv, cy = coords[row * 2 + 1] * inv;
  const float fx0 = floorf(cx), fy0 = floorf(cy);
  const int x0 = (int)fx0 - radius, y0 = (int)fy0 - radius;
  const float ax = cx - fx0, ay = cy - fy0;
  const int sub = lane / LPR, cq = lane % LPR;
  f32x4 tg[EPL / 4];
#pragma unroll
  for (int e = 0; e < EPL / 4; ++e) tg[e] = *reinterpret_cast<const f32x4*>(targets + row * C + cq * EPL + 4 * e);
  const long long img = bs * (long long)h * w * C + cq * EPL;  // element offset of this lane's channel piece in texel (0, 0)
  const int ntex = side * side;
  for (int t0 = 0; t0 < ntex; t0 += RPL * U) {
    f32x4 v[U][EPL / 4];
    int slot[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int tt = t0 + u * RPL + sub;
      const int ty = tt / side, tx = tt - ty * side;  // patch[ty][tx] = texel (x0 + tx, y0 + ty)
      const int gx = x0 + tx, gy = y0 + ty;
      slot[u] = tt < ntex ? ty * 16 + tx : -1;
#pragma unroll
      for (int e = 0; e < EPL / 4; ++e) v[u][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (tt < ntex && (unsigned)gx < (unsigned)w && (unsigned)gy < (unsigned)h) {
        const long long off = img + ((long long)gy * w + gx) * C;
        if (BF) {
          const uint4 q = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(lv.fmap[level]) + off);
          v[u][0] = (f32x4){__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xFFFF0000u), __uint_as_float(q.y << 16),
                            __uint_as_float(q.y & 0xFFFF0000u)};
          v[u][EPL / 4 - 1] = (f32x4){__uint_as_float(q.z << 16), __uint_as_float(q.z & 0xFFFF0000u), __uint_as_float(q.w << 16),
                                      __uint_as_float(q.w & 0xFFFF0000u)};
        } else {
          v[u][0] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(lv.fmap[level]) + off);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float d = tg[0][0] * v[u][0][0];
      d = fmaf(tg[0][1], v[u][0][1], d);
      d = fmaf(tg[0][2], v[u][0][2], d);
      d = fmaf(tg[0][3], v[u][0][3], d);
      if (BF) {
#pragma unroll
        for (int e = 0; e < 4; ++e) d = fmaf(tg[EPL / 4 - 1][e], v[u][EPL / 4 - 1][e], d);
      }
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
      if (cq == 0 && slot[u] >= 0) dots[wave][slot[u]] = d;
    }
  }
  __builtin_amdgcn_wave_barrier();
  const int win = 2 * radius + 1;
  const float scale = sqrtf((float)C);
  for (int o = lane; o < win * win; o += 64) {
    const int a = o / win, b = o - a * win;  // sample at (cx + a - r, cy + b - r): first index runs along x
    const float d00 = dots[wave][b * 16 + a], d01 = dots[wave][b * 16 + a + 1];
    const float d10 = dots[wave][(b + 1) * 16 + a], d11 = dots[wave][(b + 1) * 16 + a + 1];
    const float val = (1.f - ay) * ((1.f - ax) * d00 + ax * d01) + ay * ((1.f - ax) * d10 + ax * d11);
    out[row * ldo + o_off + level * win * win + o] = val / scale;
  }
}

}  // namespace

extern "C" int mvt_tile_aabb(const float* xyz, long long P, int T, int grid_w, int grid_h, float* box, void* stream) {
  MVT_REQUIRE(xyz && box && P > 0 && T > 0 && P < (1LL << 31));
  MVT_REQUIRE((grid_w == 0 && grid_h == 0) || (grid_w > 0 && grid_h > 0 && grid_w % 8 == 0 && grid_h % 8 == 0 && P % ((long long)grid_w * grid_h) == 0));
  const long long ntiles = (P + 63) / 64, total = ntiles * T;
  hipLaunchKernelGGL(tile_aabb_kernel, dim3((unsigned)mvt_cdiv(total, 4)), dim3(256), 0, mvt_stream(stream), xyz, P, ntiles, total, grid_w,
                     grid_h, box);
  return mvt_launch_status();
}

extern "C" int mvt_tile_group_aabb(const float* box, long long P, int T, float* group_box, void* stream) {
  MVT_REQUIRE(box && group_box && P > 0 && T > 0 && P < (1LL << 31));
  const long long ntiles = (P + 63) / 64, ngroups = (ntiles + 63) / 64, total = ngroups * T;
  hipLaunchKernelGGL(tile_group_aabb_kernel, dim3((unsigned)mvt_cdiv(total, 4)), dim3(256), 0, mvt_stream(stream), box, ntiles, ngroups, total,
                     group_box);
  return mvt_launch_status();
}

static int knn_scan_launch(const float* xyz, long long P, const float* coords, int N, int S, int frame0, int frame_step, int T, int K, int nseg,
                           unsigned long long* keys, const int* seed_idx, int seed_k, int seed_cw, int seed_ch, int seed_fw, int seed_fh,
                           const float* tile_box, int grid_w, int grid_h, int* idx_direct, const float* group_box, void* stream);

extern "C" int mvt_knn_scan(const float* xyz, long long P, const float* coords, int N, int S, int frame0, int frame_step, int T,
                            int K, int nseg, unsigned long long* keys, const int* seed_idx, int seed_k, int seed_cw, int seed_ch,
                            int seed_fw, int seed_fh, const float* tile_box, int grid_w, int grid_h, void* stream) {
  MVT_REQUIRE(keys);
  return knn_scan_launch(xyz, P, coords, N, S, frame0, frame_step, T, K, nseg, keys, seed_idx, seed_k, seed_cw, seed_ch, seed_fw, seed_fh,
                         tile_box, grid_w, grid_h, nullptr, nullptr, stream);
}

extern "C" int mvt_knn_search(const float* xyz, long long P, const float* coords, int N, int S, int frame0, int frame_step, int T, int K,
                              const int* seed_idx, int seed_k, int seed_cw, int seed_ch, int seed_fw, int seed_fh, const float* tile_box,
                              const float* group_box, int grid_w, int grid_h, int* idx_out, void* stream) {
  MVT_REQUIRE(idx_out && tile_box);
  return knn_scan_launch(xyz, P, coords, N, S, frame0, frame_step, T, K, 1, nullptr, seed_idx, seed_k, seed_cw, seed_ch, seed_fw, seed_fh,
                         tile_box, grid_w, grid_h, idx_out, group_box, stream);
}

static int knn_scan_launch(const float* xyz, long long P, const float* coords, int N, int S, int frame0, int frame_step, int T, int K, int nseg,
                           unsigned long long* keys, const int* seed_idx, int seed_k, int seed_cw, int seed_ch, int seed_fw, int seed_fh,
                           const float* tile_box, int grid_w, int grid_h, int* idx_direct, const float* group_box, void* stream) {
  MVT_REQUIRE(!seed_idx || (seed_k >= K && seed_k <= 64 && seed_cw >= 0));
  MVT_REQUIRE(!seed_idx || seed_cw == 0 || (seed_ch > 0 && seed_fw >= 2 * seed_cw && seed_fh >= 2 * seed_ch));
  MVT_REQUIRE(xyz && coords && (keys || idx_direct) && N > 0 && S > 0 && T > 0 && frame0 >= 0 && frame0 < T && frame_step >= 0);
  MVT_REQUIRE(K >= 1 && K <= 16 && nseg >= 1 && nseg * K <= 64 && P < (1LL << 31) && P >= K);
  MVT_REQUIRE((grid_w == 0 && grid_h == 0) || (grid_w > 0 && grid_h > 0 && grid_w % 8 == 0 && grid_h % 8 == 0 && P % ((long long)grid_w * grid_h) == 0));
  const long long ntiles = (P + 63) / 64, tper = (ntiles + nseg - 1) / nseg;
  MVT_REQUIRE(tper * (nseg - 1) < ntiles);  // no empty segment (a short one pads its key list with KEY_MAX)
  // queries per wave: 8 amortise the point loads of a brute-force scan; with boxes the scan is short and the per-query
  // selection work dominates, which spreads better over many small waves
  static const int q_env = getenv("MVT_KNN_Q") ? atoi(getenv("MVT_KNN_Q")) : 0;
  const int Q = q_env ? q_env : (tile_box ? 2 : 8);
  const int qgroups = (N + Q - 1) / Q;
  const long long ntask = (long long)qgroups * S * nseg;
  MVT_REQUIRE(ntask < (1LL << 31));
#define LAUNCH(QQ)                                                                                                                   \
  hipLaunchKernelGGL((knn_scan_kernel<QQ>), dim3((unsigned)mvt_cdiv(ntask, 4)), dim3(256), 0, mvt_stream(stream), xyz, P, coords, N, S, \
                     frame0, frame_step, T, K, nseg, keys, qgroups, seed_idx, seed_k, seed_cw, seed_ch, seed_fw, seed_fh, tile_box,    \
                     grid_w, grid_h, idx_direct, group_box)
  switch (Q) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 4: LAUNCH(4); break;
    case 8: LAUNCH(8); break;
    default: return MVT_ERR_ARG;
  }
#undef LAUNCH
  return mvt_launch_status();
}

static int knn_check_level(const mvt_knn_level& L, int K) {
  MVT_REQUIRE(L.xyz && L.keys && L.P >= K && L.P < (1LL << 31) && L.nseg >= 1 && L.nseg * K <= 64);
  MVT_REQUIRE((L.grid_w == 0 && L.grid_h == 0) ||
              (L.grid_w > 0 && L.grid_h > 0 && L.grid_w % 8 == 0 && L.grid_h % 8 == 0 && L.P % ((long long)L.grid_w * L.grid_h) == 0));
  const long long ntiles = (L.P + 63) / 64, tper = (ntiles + L.nseg - 1) / L.nseg;
  MVT_REQUIRE(tper * (L.nseg - 1) < ntiles);
  return MVT_OK;
}

extern "C" int mvt_knn_scan_levels(int levels, const mvt_knn_level* lv, const float* coords, int N, int S, int frame0, int frame_step,
                                   int T, int K, int seed_k, void* stream) {
  MVT_REQUIRE(levels >= 1 && levels <= 8 && lv && coords && N > 0 && S > 0 && T > 0 && frame0 >= 0 && frame0 < T && frame_step >= 0);
  MVT_REQUIRE(K >= 1 && K <= 16 && (seed_k == 0 || (seed_k >= K && seed_k <= 64)));
  KnnLevels a{};
  int max_nseg = 1;
  bool boxes = true;
  for (int l = 0; l < levels; ++l) {
    if (int rc = knn_check_level(lv[l], K)) return rc;
    MVT_REQUIRE((lv[l].seed_idx != nullptr) == (seed_k > 0));
    a.lv[l] = lv[l];
    max_nseg = lv[l].nseg > max_nseg ? lv[l].nseg : max_nseg;
    boxes = boxes && lv[l].tile_box;
  }
  static const int q_env = getenv("MVT_KNN_Q") ? atoi(getenv("MVT_KNN_Q")) : 0;
  const int Q = q_env ? q_env : (boxes ? 2 : 8);
  const int qgroups = (N + Q - 1) / Q;
  MVT_REQUIRE((long long)qgroups * S * max_nseg < (1LL << 31));
  const dim3 grid((unsigned)mvt_cdiv((long long)qgroups * S * max_nseg, 4), (unsigned)levels);
#define LAUNCH(QQ)                                                                                                              \
  hipLaunchKernelGGL((knn_scan_levels_kernel<QQ>), grid, dim3(256), 0, mvt_stream(stream), a, coords, N, S, frame0, frame_step, T, K, \
                     qgroups, seed_k)
  switch (Q) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 4: LAUNCH(4); break;
    case 8: LAUNCH(8); break;
    default: return MVT_ERR_ARG;
  }
#undef LAUNCH
  return mvt_launch_status();
}

extern "C" int mvt_knn_merge_levels(int levels, const mvt_knn_level* lv, int N, int S, int K, void* stream) {
  MVT_REQUIRE(levels >= 1 && levels <= 8 && lv && N > 0 && S > 0 && K >= 1 && K <= 16);
  KnnLevels a{};
  for (int l = 0; l < levels; ++l) {
    if (int rc = knn_check_level(lv[l], K)) return rc;
    MVT_REQUIRE(lv[l].idx_out);
    a.lv[l] = lv[l];
  }
  const long long rows = (long long)N * S;
  hipLaunchKernelGGL(knn_merge_levels_kernel, dim3((unsigned)mvt_cdiv(rows, 4), (unsigned)levels), dim3(256), 0, mvt_stream(stream), a,
                     rows, K);
  return mvt_launch_status();
}

extern "C" int mvt_knn_search_levels(int levels, const mvt_knn_level* lv, const float* coords, int N, int S, int frame0, int frame_step,
                                     int T, int K, int seed_k, void* stream) {
  MVT_REQUIRE(levels >= 1 && levels <= 8 && lv && coords && N > 0 && S > 0 && T > 0 && frame0 >= 0 && frame0 < T && frame_step >= 0);
  MVT_REQUIRE(K >= 1 && K <= 16 && (seed_k == 0 || (seed_k >= K && seed_k <= 64)));
  KnnLevels a{};
  for (int l = 0; l < levels; ++l) {
    const mvt_knn_level& L = lv[l];
    MVT_REQUIRE(L.xyz && L.idx_out && L.tile_box && L.P >= K && L.P < (1LL << 31) && (L.seed_idx != nullptr) == (seed_k > 0));
    MVT_REQUIRE((L.grid_w == 0 && L.grid_h == 0) ||
                (L.grid_w > 0 && L.grid_h > 0 && L.grid_w % 8 == 0 && L.grid_h % 8 == 0 && L.P % ((long long)L.grid_w * L.grid_h) == 0));
    a.lv[l] = L;
  }
  static const int q_env = getenv("MVT_KNN_Q") ? atoi(getenv("MVT_KNN_Q")) : 0;
  const int Q = q_env ? q_env : 2;
  const int qgroups = (N + Q - 1) / Q;
  MVT_REQUIRE((long long)qgroups * S < (1LL << 31));
  const dim3 grid((unsigned)mvt_cdiv((long long)qgroups * S, 4), (unsigned)levels);
#define LAUNCH(QQ)                                                                                                                \
  hipLaunchKernelGGL((knn_search_levels_kernel<QQ>), grid, dim3(256), 0, mvt_stream(stream), a, coords, N, S, frame0, frame_step, T, K, \
                     qgroups, seed_k)
  switch (Q) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 4: LAUNCH(4); break;
    case 8: LAUNCH(8); break;
    default: return MVT_ERR_ARG;
  }
#undef LAUNCH
  return mvt_launch_status();
}

extern "C" int mvt_knn_merge(const unsigned long long* keys, int N, int S, int K, int nseg, long long P, int* idx_out, void* stream) {
  MVT_REQUIRE(keys && idx_out && N > 0 && S > 0 && K >= 1 && K <= 16 && nseg >= 1 && nseg * K <= 64 && P >= K);
  const long long rows = (long long)N * S;
  hipLaunchKernelGGL(knn_merge_kernel, dim3((unsigned)mvt_cdiv(rows, 4)), dim3(256), 0, mvt_stream(stream), keys, rows, K, nseg, P,
                     idx_out);
  return mvt_launch_status();
}

extern "C" int mvt_corr_gather_dot(int levels, const float* const* xyz, const void* const* fvec, int fvec_bf16, const long long* P,
                                   const int* const* idx, int C, const float* targets, const float* coords, int N, int S,
                                   int frame0, int frame_step, int T, int K, float* out, int ldo, int o_off, void* stream) {
  MVT_REQUIRE(levels >= 1 && levels <= 8 && xyz && fvec && P && idx && targets && coords && out);
  MVT_REQUIRE((fvec_bf16 == 0 || fvec_bf16 == 1));
  MVT_REQUIRE(N > 0 && S > 0 && T > 0 && frame0 >= 0 && frame0 < T && frame_step >= 0 && K >= 1 && K <= 16);
  MVT_REQUIRE(o_off >= 0 && ldo >= o_off + levels * 4 * K);
  CorrLevels lv{};
  for (int l = 0; l < levels; ++l) {
    MVT_REQUIRE(xyz[l] && fvec[l] && idx[l] && P[l] >= K && P[l] < (1LL << 31));
    lv.xyz[l] = xyz[l];
    lv.fvec[l] = (const float*)fvec[l];
    lv.idx[l] = idx[l];
    lv.P[l] = P[l];
  }
  const dim3 grid((unsigned)mvt_cdiv((long long)N * S, 4), (unsigned)levels);
#define LAUNCH(LPR, BF)                                                                                                              \
  hipLaunchKernelGGL((corr_gather_dot_kernel<LPR, BF>), grid, dim3(256), 0, mvt_stream(stream), lv, (const float*)targets, coords, N, S, \
                     frame0, frame_step, T, K, out, ldo, o_off)
  if (fvec_bf16) {
    switch (C) {
      case 32: LAUNCH(4, 1); break;
      case 64: LAUNCH(8, 1); break;
      case 128: LAUNCH(16, 1); break;
      case 256: LAUNCH(32, 1); break;
      default: return MVT_ERR_ARG;
    }
  } else {
    switch (C) {
      case 32: LAUNCH(8, 0); break;
      case 64: LAUNCH(16, 0); break;
      case 128: LAUNCH(32, 0); break;
      case 256: LAUNCH(64, 0); break;
      default: return MVT_ERR_ARG;
    }
  }
#undef LAUNCH
  return mvt_launch_status();
}

extern "C" int mvt_knn1_gather(const void* fvec, int fvec_bf16, long long P, int C, const unsigned long long* keys, int n, int nseg,
                               int frame, float* feat_out, int* idx_out, void* stream) {
  MVT_REQUIRE(fvec && keys && feat_out && n > 0 && nseg >= 1 && nseg <= 64 && C > 0 && C % 4 == 0 && P > 0 && frame >= 0);
  MVT_REQUIRE(fvec_bf16 == 0 || fvec_bf16 == 1);
  if (fvec_bf16)
    hipLaunchKernelGGL(knn1_gather_kernel<1>, dim3((unsigned)mvt_cdiv(n, 4)), dim3(256), 0, mvt_stream(stream), (const float*)fvec, P, C, keys,
                       n, nseg, frame, feat_out, idx_out);
  else
    hipLaunchKernelGGL(knn1_gather_kernel<0>, dim3((unsigned)mvt_cdiv(n, 4)), dim3(256), 0, mvt_stream(stream), (const float*)fvec, P, C, keys,
                       n, nseg, frame, feat_out, idx_out);
  return mvt_launch_status();
}

extern "C" int mvt_window_corr(const float* fmap, const float* targets, const float* coords, float* out, int BS, int N, int C,
                               int h, int w, int level, int radius, int ldo, int o_off, void* stream) {
  MVT_REQUIRE(fmap && targets && coords && out && BS > 0 && N > 0 && h > 0 && w > 0 && level >= 0 && level < 8);
  MVT_REQUIRE(radius >= 1 && radius <= 7 && o_off >= 0 && ldo >= o_off + (2 * radius + 1) * (2 * radius + 1));
  const unsigned blocks = (unsigned)mvt_cdiv((long long)BS * N, 4);
#define LAUNCH(LPR)                                                                                                             \
  hipLaunchKernelGGL((window_corr_kernel<LPR>), dim3(blocks), dim3(256), 0, mvt_stream(stream), fmap, targets, coords, out, BS, N, h, \
                     w, level, radius, ldo, o_off)
  switch (C) {
    case 32: LAUNCH(8); break;
    case 64: LAUNCH(16); break;
    case 128: LAUNCH(32); break;
    case 256: LAUNCH(64); break;
    default: return MVT_ERR_ARG;
  }
#undef LAUNCH
  return mvt_launch_status();
}

extern "C" int mvt_window_corr_levels(int levels, const void* const* fmaps, int fmap_bf16, const int* hs, const int* ws, const float* targets,
                                      const float* coords, float* out, int BS, int N, int C, int radius, int ldo, int o_off, void* stream) {
  MVT_REQUIRE(levels >= 1 && levels <= 8 && fmaps && hs && ws && targets && coords && out && BS > 0 && N > 0);
  MVT_REQUIRE((fmap_bf16 == 0 || fmap_bf16 == 1) && radius >= 1 && radius <= 7 && o_off >= 0);
  MVT_REQUIRE(ldo >= o_off + levels * (2 * radius + 1) * (2 * radius + 1));
  WinLevels lv{};
  for (int l = 0; l < levels; ++l) {
    MVT_REQUIRE(fmaps[l] && hs[l] > 0 && ws[l] > 0 && ((uintptr_t)fmaps[l] % 16 == 0) && (long long)BS * hs[l] * ws[l] * C < (1LL << 40));
    lv.fmap[l] = fmaps[l];
    lv.h[l] = hs[l];
    lv.w[l] = ws[l];
  }
  MVT_REQUIRE((uintptr_t)targets % 16 == 0);
  const dim3 grid((unsigned)mvt_cdiv((long long)BS * N, 4), (unsigned)levels);
#define LAUNCH(LPR, BF)                                                                                                                  \
  hipLaunchKernelGGL((window_corr_levels_kernel<LPR, BF>), grid, dim3(256), 0, mvt_stream(stream), lv, targets, coords, out, BS, N, radius, \
                     ldo, o_off)
  if (fmap_bf16) {
    switch (C) {
      case 32: LAUNCH(4, 1); break;
      case 64: LAUNCH(8, 1); break;
      case 128: LAUNCH(16, 1); break;
      case 256: LAUNCH(32, 1); break;
      default: return MVT_ERR_ARG;
    }
  } else {
    switch (C) {
      case 32: LAUNCH(8, 0); break;
      case 64: LAUNCH(16, 0); break;
      case 128: LAUNCH(32, 0); break;
      case 256: LAUNCH(64, 0); break;
      default: return MVT_ERR_ARG;
    }
  }
#undef LAUNCH
  return mvt_launch_status();
}

extern "C" int mvt_corr_gather_dot_opts(int levels, const float* const* xyz, const void* const* fvec, int fvec_bf16, const long long* P,
                                        const int* const* idx, int C, const float* targets, const float* coords, int N, int S, int frame0,
                                        int frame_step, int T, int K, int groups, int add_offset, int add_xyz, float* out, int ldo, int o_off,
                                        void* stream) {
  MVT_REQUIRE(levels >= 1 && levels <= 8 && xyz && fvec && P && idx && targets && coords && out);
  MVT_REQUIRE((fvec_bf16 == 0 || fvec_bf16 == 1) && (add_offset == 0 || add_offset == 1) && (add_xyz == 0 || add_xyz == 1));
  MVT_REQUIRE(N > 0 && S > 0 && T > 0 && frame0 >= 0 && frame0 < T && frame_step >= 0 && K >= 1 && K <= 16);
  const int lpr = fvec_bf16 ? C / 8 : C / 4;  // lanes per feature row
  MVT_REQUIRE(groups >= 1 && (groups & (groups - 1)) == 0 && lpr >= 1 && groups <= lpr && lpr % groups == 0);
  const int OW = groups + 3 * add_offset + 3 * add_xyz;
  MVT_REQUIRE(o_off >= 0 && ldo >= o_off + levels * K * OW);
  CorrLevels lv{};
  for (int l = 0; l < levels; ++l) {
    MVT_REQUIRE(xyz[l] && fvec[l] && idx[l] && P[l] >= K && P[l] < (1LL << 31));
    lv.xyz[l] = xyz[l];
    lv.fvec[l] = (const float*)fvec[l];
    lv.idx[l] = idx[l];
    lv.P[l] = P[l];
  }
  const dim3 grid((unsigned)mvt_cdiv((long long)N * S, 4), (unsigned)levels);
#define LAUNCH(LPR, BF)                                                                                                                   \
  hipLaunchKernelGGL((corr_gather_dot_opts_kernel<LPR, BF>), grid, dim3(256), 0, mvt_stream(stream), lv, (const float*)targets, coords, N, S, \
                     frame0, frame_step, T, K, out, ldo, o_off, groups, add_offset, add_xyz)
  if (fvec_bf16) {
    switch (C) {
      case 32: LAUNCH(4, 1); break;
      case 64: LAUNCH(8, 1); break;
      case 128: LAUNCH(16, 1); break;
      case 256: LAUNCH(32, 1); break;
      default: return MVT_ERR_ARG;
    }
  } else {
    switch (C) {
      case 32: LAUNCH(8, 0); break;
      case 64: LAUNCH(16, 0); break;
      case 128: LAUNCH(32, 0); break;
      case 256: LAUNCH(64, 0); break;
      default: return MVT_ERR_ARG;
    }
  }
#undef LAUNCH
  return mvt_launch_status();
}
