// Frame-store construction: camera inverses, strided depth, feature pyramid (2x2 average pool)
// and the world-space point cloud of every pyramid level.  HBM-bound, 16 bytes per thread.
#include "common.h"

namespace {

__global__ void invert_cameras_kernel(const float* __restrict__ intrs, const float* __restrict__ extrs, float* __restrict__ kinv,
                                      float* __restrict__ einv, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // K^-1 by the adjugate, fp64
  double k[9];
  for (int e = 0; e < 9; ++e) k[e] = (double)intrs[i * 9 + e];
  double c00 = k[4] * k[8] - k[5] * k[7], c01 = k[5] * k[6] - k[3] * k[8], c02 = k[3] * k[7] - k[4] * k[6];
  double det = k[0] * c00 + k[1] * c01 + k[2] * c02;
  double id = 1.0 / det;
  double inv[9] = {c00 * id, (k[2] * k[7] - k[1] * k[8]) * id, (k[1] * k[5] - k[2] * k[4]) * id,
                   c01 * id, (k[0] * k[8] - k[2] * k[6]) * id, (k[2] * k[3] - k[0] * k[5]) * id,
                   c02 * id, (k[1] * k[6] - k[0] * k[7]) * id, (k[0] * k[4] - k[1] * k[3]) * id};
  for (int e = 0; e < 9; ++e) kinv[i * 9 + e] = (float)inv[e];
  // [A|t; 0 0 0 1]^-1 = [A^-1 | -A^-1 t]
  double a[9], tv[3];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) a[r * 3 + c] = (double)extrs[i * 12 + r * 4 + c];
    tv[r] = (double)extrs[i * 12 + r * 4 + 3];
  }
  double d00 = a[4] * a[8] - a[5] * a[7], d01 = a[5] * a[6] - a[3] * a[8], d02 = a[3] * a[7] - a[4] * a[6];
  double ad = 1.0 / (a[0] * d00 + a[1] * d01 + a[2] * d02);
  double ai[9] = {d00 * ad, (a[2] * a[7] - a[1] * a[8]) * ad, (a[1] * a[5] - a[2] * a[4]) * ad,
                  d01 * ad, (a[0] * a[8] - a[2] * a[6]) * ad, (a[2] * a[3] - a[0] * a[5]) * ad,
                  d02 * ad, (a[1] * a[6] - a[0] * a[7]) * ad, (a[0] * a[4] - a[1] * a[3]) * ad};
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) einv[i * 12 + r * 4 + c] = (float)ai[r * 3 + c];
    einv[i * 12 + r * 4 + 3] = (float)(-(ai[r * 3] * tv[0] + ai[r * 3 + 1] * tv[1] + ai[r * 3 + 2] * tv[2]));
  }
}

__global__ void depth_subsample_kernel(const float* __restrict__ d, float* __restrict__ out, int V, int T, int H, int W, int s) {
  const int hs = H / s, ws = W / s;
  const long long total = (long long)T * V * hs * ws;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int x = (int)(i % ws);
    long long r = i / ws;
    int y = (int)(r % hs);
    long long tv = r / hs;
    int t = (int)(tv / V), v = (int)(tv - (long long)t * V);
    out[i] = d[(((long long)v * T + t) * H + (long long)y * s) * W + (long long)x * s];
  }
}

__global__ void avgpool2_kernel(const float* __restrict__ in, float* __restrict__ out, long long n, int h, int w, int C) {
  const int ho = h / 2, wo = w / 2, cq = C / 4;
  const long long total = n * ho * wo * cq;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int q = (int)(i % cq);
    long long pix = i / cq;
    int x = (int)(pix % wo);
    long long r = pix / wo;
    int y = (int)(r % ho);
    long long img = r / ho;
    const float* b = in + ((img * h + 2 * y) * (long long)w + 2 * x) * C + q * 4;
    f32x4 a = *reinterpret_cast<const f32x4*>(b);
    f32x4 bb = *reinterpret_cast<const f32x4*>(b + C);
    f32x4 c = *reinterpret_cast<const f32x4*>(b + (long long)w * C);
    f32x4 d = *reinterpret_cast<const f32x4*>(b + (long long)w * C + C);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (((a[e] + bb[e]) + c[e]) + d[e]) * 0.25f;  // torch avg_pool2d accumulation order
    *reinterpret_cast<f32x4*>(out + i * 4) = o;
  }
}

__global__ void unproject_kernel(const float* __restrict__ depth_s, const float* __restrict__ kinv, const float* __restrict__ einv,
                                 float* __restrict__ xyz, int V, int T, int hs, int ws, int stride, int level) {
  const int f = 1 << level;
  const int h = hs >> level, w = ws >> level;
  const float st = (float)(stride * f);
  const long long total = (long long)T * V * h * w;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int x = (int)(i % w);
    long long r = i / w;
    int y = (int)(r % h);
    long long tv = r / h;
    int t = (int)(tv / V), v = (int)(tv - (long long)t * V);
    float d = depth_s[(tv * hs + (long long)y * f) * ws + (long long)x * f];
    float px = ((float)x + 0.5f) * st - 0.5f, py = ((float)y + 0.5f) * st - 0.5f;
    const float* K = kinv + ((long long)v * T + t) * 9;
    const float* E = einv + ((long long)v * T + t) * 12;
    float cx = (K[0] * px + K[1] * py + K[2]) * d;
    float cy = (K[3] * px + K[4] * py + K[5]) * d;
    float cz = (K[6] * px + K[7] * py + K[8]) * d;
    f32x4 o;
    o[0] = E[0] * cx + E[1] * cy + E[2] * cz + E[3];
    o[1] = E[4] * cx + E[5] * cy + E[6] * cz + E[7];
    o[2] = E[8] * cx + E[9] * cy + E[10] * cz + E[11];
    o[3] = 0.f;
    *reinterpret_cast<f32x4*>(xyz + i * 4) = o;
  }
}

inline unsigned grid_for(long long total) {
  long long g = mvt_cdiv(total, 256);
  return (unsigned)(g > 256 * 32 ? 256 * 32 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int mvt_invert_cameras(const float* intrs, const float* extrs, float* kinv, float* einv, int n, void* stream) {
  MVT_REQUIRE(intrs && extrs && kinv && einv && n > 0);
  hipLaunchKernelGGL(invert_cameras_kernel, dim3((n + 63) / 64), dim3(64), 0, mvt_stream(stream), intrs, extrs, kinv, einv, n);
  return mvt_launch_status();
}

extern "C" int mvt_depth_subsample(const float* depths, float* out, int V, int T, int H, int W, int s, void* stream) {
  MVT_REQUIRE(depths && out && V > 0 && T > 0 && s > 0 && H >= s && W >= s);
  long long total = (long long)T * V * (H / s) * (W / s);
  hipLaunchKernelGGL(depth_subsample_kernel, dim3(grid_for(total)), dim3(256), 0, mvt_stream(stream), depths, out, V, T, H, W, s);
  return mvt_launch_status();
}

extern "C" int mvt_avgpool2(const float* in, float* out, long long n, int h, int w, int C, void* stream) {
  MVT_REQUIRE(in && out && n > 0 && h >= 2 && w >= 2 && C > 0 && C % 4 == 0);
  long long total = n * (h / 2) * (w / 2) * (C / 4);
  hipLaunchKernelGGL(avgpool2_kernel, dim3(grid_for(total)), dim3(256), 0, mvt_stream(stream), in, out, n, h, w, C);
  return mvt_launch_status();
}

extern "C" int mvt_unproject(const float* depth_s, const float* kinv, const float* einv, float* xyz, int V, int T, int hs, int ws,
                             int stride, int level, void* stream) {
  MVT_REQUIRE(depth_s && kinv && einv && xyz && V > 0 && T > 0 && level >= 0 && level < 8);
  MVT_REQUIRE((hs >> level) > 0 && (ws >> level) > 0 && stride > 0);
  long long total = (long long)T * V * (hs >> level) * (ws >> level);
  hipLaunchKernelGGL(unproject_kernel, dim3(grid_for(total)), dim3(256), 0, mvt_stream(stream), depth_s, kinv, einv, xyz, V, T, hs, ws,
                     stride, level);
  return mvt_launch_status();
}
