// Token assembly, positional embeddings, LayerNorm, track update and the visibility head.
// All HBM-bound row kernels: one wave per row, coalesced loads, shuffle reductions.
#include "common.h"

namespace {

// pos[n][d], d < D: per axis a = d / A (A = dim3/3), j = d % A: sin(x_a * w_j) for j < A/2 else
// cos(x_a * w_{j-A/2}), w_j = 10000^(-j/(A/2)); all in fp64 like the reference's numpy path.
__global__ void pos_embed_kernel(const float* __restrict__ coords, int N, int S, int D, int dim3, const double* __restrict__ omega_tab,
                                 float* __restrict__ pos) {
  const int A = dim3 / 3, half = A / 2;
  // one thread per (track, axis, frequency): the sine and the cosine of an argument share one fp64 argument reduction (sincos)
  const int per = 3 * half;
  const long long total = (long long)N * per;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % per);
    const long long n = i / per;
    const int a = q / half, jj = q - a * half;
    // omega_j = 10000^(-j / (A/2)): from the caller's table (the reference's numpy values, embeddings.py:95-97; the fp64 pow
    // per element was most of this kernel's time) or computed here
    const double omega = omega_tab ? omega_tab[jj] : 1.0 / pow(10000.0, (double)jj / ((double)A / 2.0));
    const double arg = (double)coords[(n * S) * 3 + a] * omega;
    double sn, cs;
    sincos(arg, &sn, &cs);
    const int ds = a * A + jj, dc = ds + half;
    if (ds < D) pos[n * D + ds] = (float)sn;
    if (dc < D) pos[n * D + dc] = (float)cs;
  }
}

__global__ __launch_bounds__(256) void token_assemble_kernel(const float* __restrict__ coords, const float* __restrict__ fcorr, int Fc,
                                                             const float* __restrict__ ffeats, int C, const float* __restrict__ mask_vis,
                                                             const float* __restrict__ pos, const float* __restrict__ time_embed, int N,
                                                             int S, int E, float* __restrict__ x, int ldx) {
  const long long row = blockIdx.x;  // n * S + s
  const int n = (int)(row / S), s = (int)(row - (long long)n * S);
  const int D = 3 * E + 3 + Fc + C + 2;
  const float* c = coords + row * 3;
  const float* c0 = coords + (long long)n * S * 3;
  const float fl[3] = {c[0] - c0[0], c[1] - c0[1], c[2] - c0[2]};
  const float step = 1000.0f / (float)E;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float v;
    if (d < 3 * E) {
      const int a = d / E, w = d - a * E;
      const float div = (float)(w & ~1) * step;  // arange(0, E, 2) * (1000 / E)
      const float arg = fl[a] * div;
      v = (w & 1) ? cosf(arg) : sinf(arg);
    } else if (d < 3 * E + 3) {
      v = fl[d - 3 * E];
    } else if (d < 3 * E + 3 + Fc) {
      v = fcorr[row * Fc + (d - 3 * E - 3)];
    } else if (d < 3 * E + 3 + Fc + C) {
      v = ffeats[row * C + (d - 3 * E - 3 - Fc)];
    } else {
      v = mask_vis[row * 2 + (d - 3 * E - 3 - Fc - C)];
    }
    x[row * ldx + d] = (v + pos[(long long)n * D + d]) + time_embed[(long long)s * D + d];
  }
  // the pad columns [D, ldx) are read as K padding by the input-transform GEMM: exact zeros, written here (no memset of x)
  if (threadIdx.x < ldx - D) x[row * ldx + D + threadIdx.x] = 0.f;
}

// Per-window track state (mvtracker.py:510-511, 645-680): one launch instead of a dozen tensor ops.
//   coords [n][S][3]: tracks carried over from the previous window (n < p0) continue from its second half, the last estimate
//                     held (:648-651); new tracks start at their query point (:510)
//   mask_vis [n][S][2]: (track_mask, vis_init): the mask is (frame >= query frame) minus what earlier windows already covered
//                     (:505-507, :695), window slots past the clip repeat the last frame (:598-604); vis_init is the previous
//                     window's LOGIT for carried tracks (:653-655), 10 for new ones (:511)
//   ffeats [n][S][C]: the track's initial feature repeated over the window (:645)
__global__ void window_prepare_kernel(const float* __restrict__ qxyz, const int* __restrict__ qt, const float* __restrict__ feat_init,
                                      const float* __restrict__ prev_coords, const float* __restrict__ prev_vis, int n, int p0, int S,
                                      int C, int w, int T, float* __restrict__ coords, float* __restrict__ mask_vis,
                                      float* __restrict__ ffeats) {
  const long long total = (long long)n * S * (C / 4);
  const int half = S / 2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % (C / 4));
    const long long row = i / (C / 4);  // tr * S + s
    const int s = (int)(row % S);
    const int tr = (int)(row / S);
    *reinterpret_cast<f32x4*>(ffeats + row * C + cq * 4) = *reinterpret_cast<const f32x4*>(feat_init + (long long)tr * C + cq * 4);
    if (cq == 0) {
      const int sp = s < half ? half + s : S - 1;  // slot of the previous window this one continues from
      float vis = 10.0f;
      if (tr < p0) {
        vis = prev_vis[(long long)tr * S + sp];
#pragma unroll
        for (int a = 0; a < 3; ++a) coords[row * 3 + a] = prev_coords[((long long)tr * S + sp) * 3 + a];
      } else {
#pragma unroll
        for (int a = 0; a < 3; ++a) coords[row * 3 + a] = qxyz[(long long)tr * 3 + a];
      }
      const int S_local = T - w < S ? T - w : S;
      const int f = w + (s < S_local ? s : S_local - 1);
      const bool on = f >= qt[tr] && !(tr < p0 && f < w + half);
      mask_vis[row * 2] = on ? 1.0f : 0.0f;
      mask_vis[row * 2 + 1] = vis;
    }
  }
}

// Results of one window into the clip-level outputs, in the caller's (unsorted) query order (mvtracker.py:692-693, 710-711):
// traj [T][N][3], vis_logit / vis_prob [T][N]; order[tr] = position of sorted track tr in the caller's query list.
__global__ void window_store_kernel(const float* __restrict__ coords, const float* __restrict__ vis, const long long* __restrict__ order,
                                    int n, int S, int w, int T, int N, float* __restrict__ traj, float* __restrict__ vis_logit,
                                    float* __restrict__ vis_prob) {
  const int S_local = T - w < S ? T - w : S;
  const long long total = (long long)n * S_local;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int s = (int)(i % S_local);
    const int tr = (int)(i / S_local);
    const long long src = (long long)tr * S + s;
    const long long dst = (long long)(w + s) * N + order[tr];
#pragma unroll
    for (int a = 0; a < 3; ++a) traj[dst * 3 + a] = coords[src * 3 + a];
    const float lg = vis[src];
    vis_logit[dst] = lg;
    vis_prob[dst] = 1.0f / (1.0f + expf(-lg));
  }
}

__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ y, int ldy, long long rows, int C,
                                                        float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  float v[8];  // C <= 512
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < C ? xr[c] : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + 64 * i;
    const float d = c < C ? v[i] - mean : 0.f;
    ss += d * d;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + 64 * i;
    if (c < C) {
      float o = (v[i] - mean) * rstd;
      if (w) o = o * w[c] + b[c];
      y[row * ldy + c] = o;
    }
  }
}

__global__ __launch_bounds__(256) void delta_split_kernel(const float* __restrict__ delta, int ldd, const float* __restrict__ gw,
                                                          const float* __restrict__ gb, float* __restrict__ coords, float* __restrict__ dn,
                                                          long long rows, int C, int* __restrict__ nan_flag) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* dr = delta + row * ldd;
  if (lane < 3) {
    const float nc = coords[row * 3 + lane] + dr[lane];
    coords[row * 3 + lane] = nc;
    if (nan_flag && nc != nc) atomicOr(nan_flag, 1);
  }
  float v[4];  // C <= 256
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < C ? dr[3 + c] : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    const float d = c < C ? v[i] - mean : 0.f;
    ss += d * d;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)C + 1e-5f);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < C) dn[row * C + c] = (v[i] - mean) * rstd * gw[c] + gb[c];
  }
}

__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ out, long long rows, int C) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s = fmaf(x[row * ldx + c], w[c], s);
  s = wave_sum(s);
  if (lane == 0) out[row] = s + (b ? b[0] : 0.f);
}

__global__ void broadcast_rows_kernel(const float* __restrict__ v, float* __restrict__ x, int ld, int n, int S, int C) {
  const long long total = (long long)n * S * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long r = i / C;  // n * S + s
    x[r * ld + c] = v[(r / S) * C + c];
  }
}

inline unsigned grid_for(long long total) {
  long long g = mvt_cdiv(total, 256);
  return (unsigned)(g > 256 * 32 ? 256 * 32 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int mvt_pos_embed(const float* coords, int N, int S, int D, int dim_padded, const double* omega, float* pos, void* stream) {
  MVT_REQUIRE(coords && pos && N > 0 && S > 0 && D > 0 && dim_padded % 6 == 0 && D <= dim_padded);
  hipLaunchKernelGGL(pos_embed_kernel, dim3(grid_for((long long)N * (dim_padded / 2))), dim3(256), 0, mvt_stream(stream), coords, N, S, D, dim_padded,
                     omega, pos);
  return mvt_launch_status();
}

extern "C" int mvt_token_assemble(const float* coords, const float* fcorr, int Fc, const float* ffeats, int C,
                                  const float* mask_vis, const float* pos, const float* time_embed, int N, int S, int E, float* x,
                                  int ldx, void* stream) {
  MVT_REQUIRE(coords && fcorr && ffeats && mask_vis && pos && time_embed && x && N > 0 && S > 0 && E > 0 && E % 2 == 0);
  MVT_REQUIRE(Fc > 0 && C > 0 && ldx >= 3 * E + 3 + Fc + C + 2);
  hipLaunchKernelGGL(token_assemble_kernel, dim3((unsigned)((long long)N * S)), dim3(256), 0, mvt_stream(stream), coords, fcorr, Fc,
                     ffeats, C, mask_vis, pos, time_embed, N, S, E, x, ldx);
  return mvt_launch_status();
}

extern "C" int mvt_window_prepare(const float* qxyz, const int* qt, const float* feat_init, const float* prev_coords,
                                  const float* prev_vis, int n, int p0, int S, int C, int w, int T, float* coords, float* mask_vis,
                                  float* ffeats, void* stream) {
  MVT_REQUIRE(qxyz && qt && feat_init && coords && mask_vis && ffeats && n > 0 && p0 >= 0 && p0 <= n && S >= 2 && C > 0 && C % 4 == 0);
  MVT_REQUIRE(w >= 0 && w < T && (p0 == 0 || (prev_coords && prev_vis)));
  hipLaunchKernelGGL(window_prepare_kernel, dim3(grid_for((long long)n * S * (C / 4))), dim3(256), 0, mvt_stream(stream), qxyz, qt,
                     feat_init, prev_coords, prev_vis, n, p0, S, C, w, T, coords, mask_vis, ffeats);
  return mvt_launch_status();
}

extern "C" int mvt_window_store(const float* coords, const float* vis, const long long* order, int n, int S, int w, int T, int N,
                                float* traj, float* vis_logit, float* vis_prob, void* stream) {
  MVT_REQUIRE(coords && vis && order && traj && vis_logit && vis_prob && n > 0 && n <= N && S > 0 && w >= 0 && w < T);
  hipLaunchKernelGGL(window_store_kernel, dim3(grid_for((long long)n * S)), dim3(256), 0, mvt_stream(stream), coords, vis, order, n, S, w, T,
                     N, traj, vis_logit, vis_prob);
  return mvt_launch_status();
}

extern "C" int mvt_layernorm(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, long long rows, int C,
                             float eps, void* stream) {
  MVT_REQUIRE(x && y && rows > 0 && C > 0 && C <= 512 && ldx >= C && ldy >= C && ((w == nullptr) == (b == nullptr)));
  hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)mvt_cdiv(rows, 4)), dim3(256), 0, mvt_stream(stream), x, ldx, w, b, y, ldy, rows,
                     C, eps);
  return mvt_launch_status();
}

extern "C" int mvt_delta_split(const float* delta, int ldd, const float* gw, const float* gb, float* coords, float* dn,
                               long long rows, int C, int* nan_flag, void* stream) {
  MVT_REQUIRE(delta && gw && gb && coords && dn && rows > 0 && C > 0 && C <= 256 && ldd >= 3 + C);
  hipLaunchKernelGGL(delta_split_kernel, dim3((unsigned)mvt_cdiv(rows, 4)), dim3(256), 0, mvt_stream(stream), delta, ldd, gw, gb,
                     coords, dn, rows, C, nan_flag);
  return mvt_launch_status();
}

extern "C" int mvt_rowdot(const float* x, int ldx, const float* w, const float* b, float* out, long long rows, int C,
                          void* stream) {
  MVT_REQUIRE(x && w && out && rows > 0 && C > 0 && ldx >= C);
  hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)mvt_cdiv(rows, 4)), dim3(256), 0, mvt_stream(stream), x, ldx, w, b, out, rows, C);
  return mvt_launch_status();
}

extern "C" int mvt_broadcast_rows(const float* v, float* x, int ld, int n, int S, int C, void* stream) {
  MVT_REQUIRE(v && x && n > 0 && S > 0 && C > 0 && ld >= C);
  hipLaunchKernelGGL(broadcast_rows_kernel, dim3(grid_for((long long)n * S * C)), dim3(256), 0, mvt_stream(stream), v, x, ld, n, S, C);
  return mvt_launch_status();
}
