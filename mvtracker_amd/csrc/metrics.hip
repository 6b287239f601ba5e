// Per-track evaluation metrics of the reference's post-processing (mvtracker/evaluation/metrics.py:10-58 compute_metrics,
// :61-171 compute_tapvid_metrics with query_mode="first", :327-330 point movement) -- SURVEY.md section 8f rank 4.
// One wave per track walks the T frames (64 per step); every metric of a track is independent of the other tracks, so the
// point-type subsets of evaluate_predictions (:346-392) are plain masked means over this kernel's output.
//   out[n][0]  movement (path length over visible frames >= query frame)   out[n][1]  visible frames >= query frame
//   out[n][2..4]  occlusion accuracy (all / gt occluded / gt visible)       out[n][5], [6] average Jaccard, average pts-within
//   out[n][7..10] MTE (lower median), ATE, FDE (last visible frame), survival
//   out[n][11 + k], out[n][11 + K + k]  pts-within / Jaccard of threshold k
#include "common.h"

namespace {

constexpr int TMAX = 1024;  // frames per clip supported by the in-LDS median

struct MetricArgs {
  const float* gt;
  const float* pred;
  const unsigned char* gt_vis;
  const unsigned char* pred_occ;
  const int* qt;
  float* out;
  int T, N, D, K, ldo;
  float thr[8];
  float survival_thr;
};

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void track_metrics_kernel(MetricArgs p) {
  __shared__ float dl[4][TMAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave;
  if (n >= p.N) return;
  float* d_valid = dl[wave];
  const int T = p.T, D = p.D, K = p.K;
  const int qt = p.qt[n];
  int c_eval = 0, c_agree = 0, c_occ = 0, c_occ_agree = 0, c_vis = 0, c_vis_agree = 0, c_nvis = 0;
  int c_within[8], c_tp[8], c_fp[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) c_within[k] = c_tp[k] = c_fp[k] = 0;
  float move = 0.f;
  double dsum = 0.0;
  int nvalid = 0, last_vis = -1, first_fail = T;
  float carry[3] = {0.f, 0.f, 0.f};  // position of the last visible frame of the previous 64-frame step
  bool have_carry = false;
  for (int f0 = 0; f0 < T; f0 += 64) {
    const int f = f0 + lane;
    const bool in = f < T;
    const long long o = ((long long)(in ? f : 0) * p.N + n);
    float g[3] = {0.f, 0.f, 0.f}, q[3] = {0.f, 0.f, 0.f};
    for (int a = 0; a < D; ++a) {
      g[a] = p.gt[o * D + a];
      q[a] = p.pred[o * D + a];
    }
    const bool vis = in && p.gt_vis[o] != 0 && f >= qt;     // gt visibility masked to frames >= the query frame (:322-323)
    const bool pocc = p.pred_occ[o] != 0;
    const bool gocc = !vis;
    const bool ev = in && f != qt && f >= qt;               // evaluation points (:119-128)
    float d2 = 0.f;
    for (int a = 0; a < D; ++a) d2 = fmaf(q[a] - g[a], q[a] - g[a], d2);
    const float dist = sqrtf(d2);
    c_eval += ev;
    c_agree += ev && (pocc == gocc);
    c_occ += ev && gocc;
    c_occ_agree += ev && gocc && (pocc == gocc);
    c_vis += ev && !gocc;
    c_vis_agree += ev && !gocc && (pocc == gocc);
    c_nvis += vis;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < K) {
        const bool within = dist < p.thr[k];
        c_within[k] += ev && within && !gocc;
        c_tp[k] += ev && within && !pocc && !gocc;
        c_fp[k] += ev && ((!within && !pocc) || (!pocc && gocc));
      }
    }
    // trajectory errors over visible frames (the query frame included, metrics.py:27-31)
    const unsigned long long vb = __ballot(vis);
    if (vis) {
      dsum += (double)dist;
      last_vis = f;
      if (dist > p.survival_thr && f < first_fail) first_fail = f;
    }
    const int below = __popcll(vb & ((1ull << lane) - 1ull));
    if (vis) d_valid[nvalid + below] = dist;
    // movement: distance to the previous visible frame (in this step, or carried over from an earlier one)
    const unsigned long long lower = vb & ((1ull << lane) - 1ull);
    const int prev_lane = lower ? 63 - __clzll(lower) : -1;
    float pg[3];
    for (int a = 0; a < 3; ++a) pg[a] = __shfl(g[a], prev_lane < 0 ? 0 : prev_lane, 64);
    if (vis) {
      bool has = prev_lane >= 0;
      if (!has && have_carry) {
        has = true;
        for (int a = 0; a < 3; ++a) pg[a] = carry[a];
      }
      if (has) {
        float m2 = 0.f;
        for (int a = 0; a < D; ++a) m2 = fmaf(g[a] - pg[a], g[a] - pg[a], m2);
        move += sqrtf(m2);
      }
    }
    if (vb) {
      const int top = 63 - __clzll(vb);
      for (int a = 0; a < 3; ++a) carry[a] = __shfl(g[a], top, 64);
      have_carry = true;
    }
    nvalid += __popcll(vb);
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  // lower median of the nvalid distances: the element whose rank (ties broken by position) is (nvalid - 1) / 2
  float med = __builtin_nanf("");
  const int want = (nvalid - 1) / 2;
  for (int i = lane; i < nvalid; i += 64) {
    const float di = d_valid[i];
    int rank = 0;
    for (int j = 0; j < nvalid; ++j) {
      const float dj = d_valid[j];
      rank += (dj < di) || (dj == di && j < i);
    }
    if (rank == want) med = di;
  }
  // exactly one lane holds the median (ranks are a permutation): broadcast it
  const unsigned long long mb = __ballot(med == med);
  if (mb) med = __shfl(med, __ffsll((long long)mb) - 1, 64);

  // wave reductions
  c_eval = wave_sum_i(c_eval); c_agree = wave_sum_i(c_agree); c_occ = wave_sum_i(c_occ); c_occ_agree = wave_sum_i(c_occ_agree);
  c_vis = wave_sum_i(c_vis); c_vis_agree = wave_sum_i(c_vis_agree); c_nvis = wave_sum_i(c_nvis);
  move = wave_sum(move);
  double ds = dsum;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ds += __shfl_xor(ds, o, 64);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    last_vis = max(last_vis, __shfl_xor(last_vis, o, 64));
    first_fail = min(first_fail, __shfl_xor(first_fail, o, 64));
  }
  float pw[8], jc[8];
  float spw = 0.f, sjc = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    pw[k] = jc[k] = 0.f;
    if (k < K) {
      const int w = wave_sum_i(c_within[k]), tp = wave_sum_i(c_tp[k]), fp = wave_sum_i(c_fp[k]);
      pw[k] = (float)w / (float)c_vis;
      jc[k] = (float)tp / ((float)c_vis + (float)fp);
      spw += pw[k];
      sjc += jc[k];
    }
  }
  if (lane == 0) {
    float* o = p.out + (long long)n * p.ldo;
    o[0] = move;
    o[1] = (float)c_nvis;
    o[2] = (float)c_agree / (float)c_eval;
    o[3] = (float)c_occ_agree / (float)c_occ;
    o[4] = (float)c_vis_agree / (float)c_vis;
    o[5] = sjc / (float)K;
    o[6] = spw / (float)K;
    o[7] = med;
    o[8] = nvalid ? (float)(ds / (double)nvalid) : __builtin_nanf("");
    // FDE: the distance at the last visible frame (metrics.py:39-41; NaN when no frame is visible)
    float fde = __builtin_nanf("");
    if (last_vis >= 0) {
      const long long ol = (long long)last_vis * p.N + n;
      float d2 = 0.f;
      for (int a = 0; a < D; ++a) d2 = fmaf(p.pred[ol * D + a] - p.gt[ol * D + a], p.pred[ol * D + a] - p.gt[ol * D + a], d2);
      fde = sqrtf(d2);
    }
    o[9] = fde;
    o[10] = (float)(first_fail - qt) / (float)(T - qt);
    for (int k = 0; k < K; ++k) {
      o[11 + k] = pw[k];
      o[11 + K + k] = jc[k];
    }
  }
}

}  // namespace

extern "C" int mvt_track_metrics(const float* gt_tracks, const float* pred_tracks, const unsigned char* gt_visible,
                                 const unsigned char* pred_occluded, const int* query_frame, int T, int N, int D,
                                 const float* thresholds /* host */, int K, float survival_threshold, float* out, int ldo, void* stream) {
  MVT_REQUIRE(gt_tracks && pred_tracks && gt_visible && pred_occluded && query_frame && thresholds && out);
  MVT_REQUIRE(T >= 1 && T <= TMAX && N >= 1 && (D == 2 || D == 3) && K >= 1 && K <= 8 && ldo >= 11 + 2 * K);
  MetricArgs a{};
  a.gt = gt_tracks; a.pred = pred_tracks; a.gt_vis = gt_visible; a.pred_occ = pred_occluded; a.qt = query_frame; a.out = out;
  a.T = T; a.N = N; a.D = D; a.K = K; a.ldo = ldo; a.survival_thr = survival_threshold;
  for (int k = 0; k < K; ++k) a.thr[k] = thresholds[k];
  hipLaunchKernelGGL(track_metrics_kernel, dim3((unsigned)mvt_cdiv(N, 4)), dim3(256), 0, mvt_stream(stream), a);
  return mvt_launch_status();
}
