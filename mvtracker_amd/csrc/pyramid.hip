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

template <int BF>  // BF: bf16 rows in and out (the bf16 frame store); the 2x2 mean is taken in fp32 either way
__global__ void avgpool2_kernel(const float* __restrict__ in, float* __restrict__ out, long long n, int h, int w, int C) {
  constexpr int E = BF ? 8 : 4;  // elements per 16-byte lane access
  const int ho = h / 2, wo = w / 2, cq = C / E;
  const long long total = n * ho * wo * cq;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int q = (int)(i % cq);
    long long pix = i / cq;
    int x = (int)(pix % wo);
    long long r = pix / wo;
    int y = (int)(r % ho);
    long long img = r / ho;
    const long long b = ((img * h + 2 * y) * (long long)w + 2 * x) * C + q * E;
#pragma unroll
    for (int hf = 0; hf < E / 4; ++hf) {
      const f32x4 a = load_act4(in, b + 4 * hf, BF), bb = load_act4(in, b + C + 4 * hf, BF);
      const f32x4 c = load_act4(in, b + (long long)w * C + 4 * hf, BF), d = load_act4(in, b + (long long)w * C + C + 4 * hf, BF);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (((a[e] + bb[e]) + c[e]) + d[e]) * 0.25f;  // torch avg_pool2d accumulation order
      store_act4(out, i * E + 4 * hf, o, BF);
    }
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

// View assignment of the monocular-to-multi-view adapter (monocular_baselines.py:630-680): project a query into every view
// at its query frame, sample that view's depth map (bilinear_sample2d, model_utils.py:81-165: four clamped taps, weights
// from the unclamped corners), score = sampled depth - camera z (-1e4 - z outside the image or behind the camera) and take
// the first maximum over the views.  One thread per query.
__global__ void adapter_best_view_kernel(const float* __restrict__ depths, const float* __restrict__ intrs, const float* __restrict__ extrs,
                                         const float* __restrict__ qp, int V, int T, int H, int W, int N, int* __restrict__ view_out,
                                         float* __restrict__ xyz_out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  int t = (int)qp[n * 4];  // .long(): truncation toward zero
  t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
  const float px = qp[n * 4 + 1], py = qp[n * 4 + 2], pz = qp[n * 4 + 3];
  float best = -INFINITY;
  int bv = 0;
  for (int v = 0; v < V; ++v) {
    const float* E = extrs + ((long long)v * T + t) * 12;
    const float* K = intrs + ((long long)v * T + t) * 9;
    float cam[3], ph[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) cam[i] = ((E[i * 4] * px + E[i * 4 + 1] * py) + E[i * 4 + 2] * pz) + E[i * 4 + 3];
#pragma unroll
    for (int i = 0; i < 3; ++i) ph[i] = (K[i * 3] * cam[0] + K[i * 3 + 1] * cam[1]) + K[i * 3 + 2] * cam[2];
    const float x = ph[0] / ph[2], y = ph[1] / ph[2], z = cam[2];
    const float xf = floorf(x), yf = floorf(y);
    const int x0 = (int)xf, y0 = (int)yf;
    const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x0 + 1, 0), W - 1);
    const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y0 + 1, 0), H - 1);
    const float* im = depths + ((long long)v * T + t) * H * W;
    const float x1f = (float)(x0 + 1), y1f = (float)(y0 + 1);
    const float w00 = (x1f - x) * (y1f - y), w01 = (x - xf) * (y1f - y), w10 = (x1f - x) * (y - yf), w11 = (x - xf) * (y - yf);
    float d = ((w00 * im[cy0 * W + cx0] + w01 * im[cy0 * W + cx1]) + w10 * im[cy1 * W + cx0]) + w11 * im[cy1 * W + cx1];
    const bool outside = x < 0.f || x >= (float)W || y < 0.f || y >= (float)H || z < 0.f;
    if (outside) d = -1e4f;
    const float score = d - z;
    if (xyz_out) {
      float* o = xyz_out + ((long long)v * N + n) * 3;
      o[0] = x;
      o[1] = y;
      o[2] = z;
    }
    if (score > best) {  // strict: the first maximum wins (torch.argmax)
      best = score;
      bv = v;
    }
  }
  view_out[n] = bv;
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

extern "C" int mvt_avgpool2(const void* in, void* out, long long n, int h, int w, int C, int io_flags, void* stream) {
  const int bf = io_flags == (MVT_IO_IN_BF16 | MVT_IO_OUT_BF16);
  MVT_REQUIRE(io_flags == 0 || bf);  // the pyramid keeps the element type of level 0
  MVT_REQUIRE(in && out && n > 0 && h >= 2 && w >= 2 && C > 0 && C % (bf ? 8 : 4) == 0);
  long long total = n * (h / 2) * (w / 2) * (C / (bf ? 8 : 4));
  if (bf)
    hipLaunchKernelGGL(avgpool2_kernel<1>, dim3(grid_for(total)), dim3(256), 0, mvt_stream(stream), (const float*)in, (float*)out, n, h, w, C);
  else
    hipLaunchKernelGGL(avgpool2_kernel<0>, dim3(grid_for(total)), dim3(256), 0, mvt_stream(stream), (const float*)in, (float*)out, n, h, w, C);
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

extern "C" int mvt_adapter_best_view(const float* depths, const float* intrs, const float* extrs, const float* query_points, int V, int T,
                                     int H, int W, int N, int* view_out, float* xyz_out, void* stream) {
  MVT_REQUIRE(depths && intrs && extrs && query_points && view_out && V > 0 && T > 0 && H > 0 && W > 0 && N > 0);
  hipLaunchKernelGGL(adapter_best_view_kernel, dim3((unsigned)mvt_cdiv(N, 128)), dim3(128), 0, mvt_stream(stream), depths, intrs, extrs,
                     query_points, V, T, H, W, N, view_out, xyz_out);
  return mvt_launch_status();
}
