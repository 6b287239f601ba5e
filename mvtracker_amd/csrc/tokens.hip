// Token assembly, positional embeddings, LayerNorm, track update and the visibility head.
// All HBM-bound row kernels: one wave per row, coalesced loads, shuffle reductions.
#include "common.h"

namespace {

// pos[n][d], d < D: per axis a = d / A (A = dim3/3), j = d % A: sin(x_a * w_j) for j < A/2 else
// cos(x_a * w_{j-A/2}), w_j = 10000^(-j/(A/2)); all in fp64 like the reference's numpy path.
__global__ void pos_embed_kernel(const float* __restrict__ coords, int N, int S, int D, int dim3, float* __restrict__ pos) {
  const int A = dim3 / 3, half = A / 2;
  const long long total = (long long)N * D;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const long long n = i / D;
    const int a = d / A, j = d - a * A;
    const int jj = j < half ? j : j - half;
    const double omega = 1.0 / pow(10000.0, (double)jj / ((double)A / 2.0));
    const double arg = (double)coords[(n * S) * 3 + a] * omega;
    pos[i] = (float)(j < half ? sin(arg) : cos(arg));
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

extern "C" int mvt_pos_embed(const float* coords, int N, int S, int D, int dim_padded, float* pos, void* stream) {
  MVT_REQUIRE(coords && pos && N > 0 && S > 0 && D > 0 && dim_padded % 6 == 0 && D <= dim_padded);
  hipLaunchKernelGGL(pos_embed_kernel, dim3(grid_for((long long)N * D)), dim3(256), 0, mvt_stream(stream), coords, N, S, D, dim_padded, pos);
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
