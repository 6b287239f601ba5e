// HBM-bound encoder-side kernels: input repack, nearest / align-corners bilinear resize,
// InstanceNorm statistics and the fused normalise + ReLU (+ residual) pass.  All tensors are
// channels-last; every thread moves 16 bytes.
#include <stdlib.h>

#include "common.h"

namespace {

template <typename TIN>
__global__ void rgb_to_nhwc4_kernel(const TIN* __restrict__ rgbs, float* __restrict__ out, int V, int T, int H, int W,
                                    long long img0, long long nimg) {
  // images are numbered frame-major: image t * V + v is view v of frame t (the order of the frame store)
  const long long hw = (long long)H * W;
  const long long total = nimg * hw;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long img = i / hw;
    long long pix = i - img * hw;
    int tt = (int)((img0 + img) / V), v = (int)((img0 + img) - (long long)tt * V);
    const TIN* src = rgbs + (((long long)v * T + tt) * 3) * hw + pix;
    f32x4 o;
    o[0] = 2.0f * ((float)src[0] / 255.0f) - 1.0f;
    o[1] = 2.0f * ((float)src[hw] / 255.0f) - 1.0f;
    o[2] = 2.0f * ((float)src[2 * hw] / 255.0f) - 1.0f;
    o[3] = 0.0f;
    *reinterpret_cast<f32x4*>(out + i * 4) = o;
  }
}

__global__ void resize_nearest_kernel(const float* __restrict__ in, float* __restrict__ out, long long planes, int Hi, int Wi,
                                      int Ho, int Wo) {
  // torch nearest: src = min(floor(dst * (float)in/out), in - 1)
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  const long long total = planes * Ho * Wo;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int x = (int)(i % Wo);
    long long r = i / Wo;
    int y = (int)(r % Ho);
    long long pl = r / Ho;
    int sy = min((int)floorf(y * sh), Hi - 1);
    int sx = min((int)floorf(x * sw), Wi - 1);
    out[i] = in[(pl * Hi + sy) * (long long)Wi + sx];
  }
}

// stage 1: block (slab, image) sums channel quads over its pixel slab in fp64
__global__ __launch_bounds__(256) void instnorm_partial_kernel(const float* __restrict__ x, int ldx, double* __restrict__ partial,
                                                               long long HW, int C, int bf) {
  __shared__ double red[256 * 8];
  const int slab = blockIdx.x, img = blockIdx.y;
  const int cq = C / 4;              // channel quads per pixel
  const int lanes = 256 / cq;        // pixel lanes (cq <= 64 -> lanes >= 4)
  const int t = threadIdx.x;
  const int q = t % cq, pl = t / cq;
  const long long per = (HW + MVT_IN_SLABS - 1) / MVT_IN_SLABS;
  const long long p0 = slab * per, p1 = min(HW, p0 + per);
  double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
  if (pl < lanes) {
    const long long base = (long long)img * HW * ldx + q * 4;
    for (long long p = p0 + pl; p < p1; p += lanes) {
      f32x4 v = load_act4(x, base + p * ldx, bf);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[e] += (double)v[e];
        ss[e] += (double)v[e] * (double)v[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[t * 8 + e] = s[e];
    red[t * 8 + 4 + e] = ss[e];
  }
  __syncthreads();
  if (t < cq) {  // fixed-order reduction over pixel lanes -> deterministic
    double a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = 0;
    for (int l = 0; l < lanes; ++l)
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += red[(l * cq + t) * 8 + e];
    double* dst = partial + (((long long)img * MVT_IN_SLABS + slab) * C + t * 4) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dst[e * 2] = a[e];
      dst[e * 2 + 1] = a[4 + e];
    }
  }
}

__global__ void instnorm_finish_kernel(const double* __restrict__ partial, float* __restrict__ mean_rstd, long long HW, int C,
                                       long long total) {
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;  // (img, c)
  if (i >= total) return;
  long long img = i / C;
  int c = (int)(i - img * C);
  double s = 0, ss = 0;
  for (int slab = 0; slab < MVT_IN_SLABS; ++slab) {
    const double* p = partial + ((img * MVT_IN_SLABS + slab) * C + c) * 2;
    s += p[0];
    ss += p[1];
  }
  double mean = s / (double)HW;
  double var = ss / (double)HW - mean * mean;
  if (var < 0) var = 0;
  mean_rstd[i * 2] = (float)mean;
  mean_rstd[i * 2 + 1] = (float)(1.0 / sqrt(var + 1e-5));
}

// block (channel group of FC, image): 1024 / FC slot lanes x FC channels accumulate in fp64, fixed-order LDS reduction (deterministic).
// FC = 8: twice the workgroups of the first version (16 channels) -- at 64 channels x 24 images that one ran on 96 workgroups only.
constexpr int FC = 8, FL = 1024 / FC;
__global__ __launch_bounds__(1024) void instnorm_finish_slots_kernel(const float* __restrict__ partial, int slots,
                                                                     float* __restrict__ mean_rstd, long long HW, int C) {
  __shared__ double red[1024 * 2];
  const int t = threadIdx.x, c = blockIdx.x * FC + (t % FC), l = t / FC;
  const long long img = blockIdx.y;
  double s = 0, ss = 0;
  if (c < C) {
    for (int sl = l; sl < slots; sl += FL) {
      const float2 v = *reinterpret_cast<const float2*>(partial + ((img * slots + sl) * C + c) * 2);
      s += (double)v.x;
      ss += (double)v.y;
    }
  }
  red[t * 2] = s;
  red[t * 2 + 1] = ss;
  __syncthreads();
  // tree over the slot lanes in a fixed order (pairs at distance 64, 32, ... lanes): deterministic
  for (int w = FL / 2; w >= 1; w >>= 1) {
    if (l < w) {
      red[t * 2] += red[(t + w * FC) * 2];
      red[t * 2 + 1] += red[(t + w * FC) * 2 + 1];
    }
    __syncthreads();
  }
  if (t < FC && c < C) {
    s = red[t * 2], ss = red[t * 2 + 1];
    double mean = s / (double)HW;
    double var = ss / (double)HW - mean * mean;
    if (var < 0) var = 0;
    mean_rstd[(img * C + c) * 2] = (float)mean;
    mean_rstd[(img * C + c) * 2 + 1] = (float)(1.0 / sqrt(var + 1e-5));
  }
}

// (a 16-byte-per-lane, 32-bit-index variant of this kernel measured 8 % SLOWER: the per-lane statistics reads double)
__global__ void instnorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ st, const float* __restrict__ skip,
                                      const float* __restrict__ skst, float* __restrict__ y, long long HW, int C, long long total4,
                                      int bf, int skip_relu) {
  const int cq = C / 4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    int q = (int)(i % cq);
    long long img = (i / cq) / HW;
    f32x4 v = load_act4(x, i * 4, bf);
    const float* s = st + (img * C + q * 4) * 2;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaxf((v[e] - s[e * 2]) * s[e * 2 + 1], 0.0f);
    if (skip) {
      f32x4 k = load_act4(skip, i * 4, bf);
      if (skst) {
        const float* ks = skst + (img * C + q * 4) * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) k[e] = (k[e] - ks[e * 2]) * ks[e * 2 + 1];
        if (skip_relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) k[e] = fmaxf(k[e], 0.0f);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf(k[e] + o[e], 0.0f);
    }
    store_act4(y, i * 4, o, bf);
  }
}

// bf16 tensors, C % 8 == 0: a block owns a run of pixels of ONE image (grid.y = image), a thread owns one channel octet --
// its statistics sit in registers, every access is 16 bytes, and there is no integer division in the loop (the generic kernel
// above spends a 64-bit division per element quad and re-reads the statistics for every quad).  UN pixels per thread are in
// flight together.
template <bool SKIP, bool SKST>
__global__ __launch_bounds__(256) void instnorm_apply_bf16_kernel(const unsigned short* __restrict__ x, const float* __restrict__ st,
                                                                  const unsigned short* __restrict__ skip, const float* __restrict__ skst,
                                                                  unsigned short* __restrict__ y, int HW, int C, int ppb, int skip_relu) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  constexpr int UN = 4;
  const int c8n = C >> 3;                       // octets per pixel
  const int lanes = (256 / c8n) * c8n;          // active threads (whole pixels)
  const int t = threadIdx.x;
  if (t >= lanes) return;
  const int oc = t % c8n, p0 = t / c8n, pstep = 256 / c8n;
  const long long img = blockIdx.y;
  float m[8], rs[8], km[8], kr[8];
  {
    const float* s = st + (img * C + oc * 8) * 2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      m[e] = s[2 * e];
      rs[e] = s[2 * e + 1];
    }
    if (SKST) {
      const float* k = skst + (img * C + oc * 8) * 2;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        km[e] = k[2 * e];
        kr[e] = k[2 * e + 1];
      }
    }
  }
  const int pb = blockIdx.x * ppb, pe = min(pb + ppb, HW);
  const long long base = img * (long long)HW * C + oc * 8;
  auto unpack = [](const u32x4& w, float (&f)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f[2 * e] = __uint_as_float(w[e] << 16);
      f[2 * e + 1] = __uint_as_float(w[e] & 0xFFFF0000u);
    }
  };
  for (int p = pb + p0; p < pe; p += pstep * UN) {
    u32x4 xv[UN], kv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int pp = p + u * pstep;
      if (pp < pe) {
        xv[u] = *reinterpret_cast<const u32x4*>(x + base + (long long)pp * C);
        if (SKIP) kv[u] = *reinterpret_cast<const u32x4*>(skip + base + (long long)pp * C);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int pp = p + u * pstep;
      if (pp >= pe) break;
      float f[8], k[8];
      unpack(xv[u], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = fmaxf((f[e] - m[e]) * rs[e], 0.0f);
      if (SKIP) {
        unpack(kv[u], k);
        if (SKST) {
#pragma unroll
          for (int e = 0; e < 8; ++e) k[e] = (k[e] - km[e]) * kr[e];
          if (skip_relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) k[e] = fmaxf(k[e], 0.0f);
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = fmaxf(k[e] + f[e], 0.0f);
      }
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (unsigned)mvt_bf16_bits(f[2 * e]) | ((unsigned)mvt_bf16_bits(f[2 * e + 1]) << 16);
      *reinterpret_cast<u32x4*>(y + base + (long long)pp * C) = o;
    }
  }
}

__global__ void resize_bilinear_ac_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int Hs, int Ws, int C, int Hd,
                                          int Wd, int ldd, int c_off, int bf) {
  // torch upsample_bilinear2d, align_corners=True: src = dst * (in-1)/(out-1)
  const float rh = Hd > 1 ? (float)(Hs - 1) / (float)(Hd - 1) : 0.f;
  const float rw = Wd > 1 ? (float)(Ws - 1) / (float)(Wd - 1) : 0.f;
  const int cq = C / 4;
  const long long total = (long long)n * Hd * Wd * cq;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int q = (int)(i % cq);
    long long pix = i / cq;
    int x = (int)(pix % Wd);
    long long r = pix / Wd;
    int y = (int)(r % Hd);
    long long img = r / Hd;
    float fy = rh * y, fx = rw * x;
    int y0 = (int)fy, x0 = (int)fx;
    int yp = y0 < Hs - 1 ? 1 : 0, xp = x0 < Ws - 1 ? 1 : 0;
    float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
    const long long b = ((img * Hs + y0) * (long long)Ws + x0) * C + q * 4;
    f32x4 a00 = load_act4(src, b, bf);
    f32x4 a01 = load_act4(src, b + (long long)xp * C, bf);
    f32x4 a10 = load_act4(src, b + (long long)yp * Ws * C, bf);
    f32x4 a11 = load_act4(src, b + ((long long)yp * Ws + xp) * C, bf);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = hy * (hx * a00[e] + lx * a01[e]) + ly * (hx * a10[e] + lx * a11[e]);
    store_act4(dst, pix * ldd + c_off + q * 4, o, bf);
  }
}

// The encoder's concat (spatracker/blocks.py:266-276): up to four stage outputs, each resized bilinearly (align_corners=True) to
// the H/4 grid, written side by side into one [n][Hd][Wd][ldd] tensor.  One thread per 16-byte piece of the CONCATENATED pixel
// row, so a wave writes whole contiguous rows (832 B per pixel at 416 bf16 channels): four separate resize launches each wrote a
// 128-192-256-byte slice of every row at offsets that are not multiples of the 128-B line.  Same arithmetic as
// resize_bilinear_ac_kernel.
struct ConcatSrc {
  const float* p[4];
  int Hs[4], Ws[4], C[4], c0[4];  // c0: first output channel of the source
  int nsrc;
};
template <int BF>
__global__ void concat_resize_kernel(ConcatSrc cs, float* __restrict__ dst, int n, int Hd, int Wd, int ldd) {
  constexpr int E = BF ? 8 : 4;  // channels per 16-byte piece
  const int ctot = cs.c0[cs.nsrc - 1] + cs.C[cs.nsrc - 1];
  const int ppr = ctot / E;      // pieces per pixel row
  // (32-bit index arithmetic: the host checks that n * Hd * Wd * ppr fits; 64-bit divisions were a third of the kernel)
  const unsigned total = (unsigned)n * Hd * Wd * ppr;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned pixu = i / (unsigned)ppr;
    const int piece = (int)(i - pixu * (unsigned)ppr);
    const long long pix = pixu;
    const int ch = piece * E;
    int s = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (k < cs.nsrc && ch >= cs.c0[k]) s = k;
    const int Hs = cs.Hs[s], Ws = cs.Ws[s], C = cs.C[s];
    const float* src = cs.p[s];
    const unsigned ru = pixu / (unsigned)Wd;
    const int x = (int)(pixu - ru * (unsigned)Wd);
    const unsigned imgu = ru / (unsigned)Hd;
    const int y = (int)(ru - imgu * (unsigned)Hd);
    const long long img = imgu;
    const float rh = Hd > 1 ? (float)(Hs - 1) / (float)(Hd - 1) : 0.f;
    const float rw = Wd > 1 ? (float)(Ws - 1) / (float)(Wd - 1) : 0.f;
    float fy = rh * y, fx = rw * x;
    int y0 = (int)fy, x0 = (int)fx;
    int yp = y0 < Hs - 1 ? 1 : 0, xp = x0 < Ws - 1 ? 1 : 0;
    float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
    const long long b = ((img * Hs + y0) * (long long)Ws + x0) * C + (ch - cs.c0[s]);
#pragma unroll
    for (int hf = 0; hf < E / 4; ++hf) {
      f32x4 a00 = load_act4(src, b + 4 * hf, BF);
      f32x4 a01 = load_act4(src, b + (long long)xp * C + 4 * hf, BF);
      f32x4 a10 = load_act4(src, b + (long long)yp * Ws * C + 4 * hf, BF);
      f32x4 a11 = load_act4(src, b + ((long long)yp * Ws + xp) * C + 4 * hf, BF);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = hy * (hx * a00[e] + lx * a01[e]) + ly * (hx * a10[e] + lx * a11[e]);
      store_act4(dst, pix * ldd + ch + 4 * hf, o, BF);
    }
  }
}

// bf16 form of the same concat (one block per output image row, grid.y = image): a thread keeps ONE channel octet -- its source
// map, channel offset and vertical taps are fixed for the whole row -- and walks the pixels; 16-byte loads of the four taps, no
// division in the loop (the generic kernel above pays three 32-bit divisions and eight 8-byte loads per piece).
__global__ __launch_bounds__(256) void concat_resize_rows_bf16_kernel(ConcatSrc cs, unsigned short* __restrict__ dst, int Hd, int Wd, int ldd) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const int ctot = cs.c0[cs.nsrc - 1] + cs.C[cs.nsrc - 1];
  const int ppr = ctot >> 3;                  // octets per pixel
  const int pstep = 256 / ppr;                // pixels per sweep of the block
  const int t = threadIdx.x;
  if (t >= pstep * ppr) return;
  const int oc = t % ppr, px0 = t / ppr;
  const int ch = oc * 8;
  int s = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k)
    if (k < cs.nsrc && ch >= cs.c0[k]) s = k;
  const int Hs = cs.Hs[s], Ws = cs.Ws[s], C = cs.C[s];
  const unsigned short* src = reinterpret_cast<const unsigned short*>(cs.p[s]);
  const int y = blockIdx.x;
  const long long img = blockIdx.y;
  const float rh = Hd > 1 ? (float)(Hs - 1) / (float)(Hd - 1) : 0.f;
  const float rw = Wd > 1 ? (float)(Ws - 1) / (float)(Wd - 1) : 0.f;
  const float fy = rh * y;
  const int y0 = (int)fy;
  const int yp = y0 < Hs - 1 ? 1 : 0;
  const float ly = fy - y0, hy = 1.f - ly;
  const unsigned short* r0 = src + ((img * Hs + y0) * (long long)Ws) * C + (ch - cs.c0[s]);
  const unsigned short* r1 = r0 + (long long)yp * Ws * C;
  unsigned short* drow = dst + ((img * Hd + y) * (long long)Wd) * ldd + ch;
  auto unpack = [](const u32x4& w, float (&f)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f[2 * e] = __uint_as_float(w[e] << 16);
      f[2 * e + 1] = __uint_as_float(w[e] & 0xFFFF0000u);
    }
  };
  for (int x = px0; x < Wd; x += pstep) {
    const float fx = rw * x;
    const int x0 = (int)fx;
    const int xp = x0 < Ws - 1 ? 1 : 0;
    const float lx = fx - x0, hx = 1.f - lx;
    const u32x4 w00 = *reinterpret_cast<const u32x4*>(r0 + (long long)x0 * C), w01 = *reinterpret_cast<const u32x4*>(r0 + (long long)(x0 + xp) * C);
    const u32x4 w10 = *reinterpret_cast<const u32x4*>(r1 + (long long)x0 * C), w11 = *reinterpret_cast<const u32x4*>(r1 + (long long)(x0 + xp) * C);
    float a00[8], a01[8], a10[8], a11[8];
    unpack(w00, a00);
    unpack(w01, a01);
    unpack(w10, a10);
    unpack(w11, a11);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v0 = hy * (hx * a00[2 * e] + lx * a01[2 * e]) + ly * (hx * a10[2 * e] + lx * a11[2 * e]);
      const float v1 = hy * (hx * a00[2 * e + 1] + lx * a01[2 * e + 1]) + ly * (hx * a10[2 * e + 1] + lx * a11[2 * e + 1]);
      o[e] = (unsigned)mvt_bf16_bits(v0) | ((unsigned)mvt_bf16_bits(v1) << 16);
    }
    *reinterpret_cast<u32x4*>(drow + (long long)x * ldd) = o;
  }
}

inline unsigned grid_for(long long total, int block = 256) {
  long long g = mvt_cdiv(total, block);
  return (unsigned)(g > 256 * 32 ? 256 * 32 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int mvt_concat_resize_bilinear_ac(int nsrc, const void* const* srcs, const int* Hs, const int* Ws, const int* Cs, void* dst,
                                             int n, int Hd, int Wd, int ldd, int io_flags, void* stream) {
  MVT_REQUIRE(nsrc >= 1 && nsrc <= 4 && srcs && Hs && Ws && Cs && dst && n > 0 && Hd > 0 && Wd > 0);
  const int bf = io_flags == (MVT_IO_IN_BF16 | MVT_IO_OUT_BF16);
  MVT_REQUIRE(io_flags == 0 || bf);
  const int E = bf ? 8 : 4;
  ConcatSrc cs{};
  cs.nsrc = nsrc;
  int c0 = 0;
  for (int k = 0; k < nsrc; ++k) {
    MVT_REQUIRE(srcs[k] && Hs[k] > 0 && Ws[k] > 0 && Cs[k] > 0 && Cs[k] % E == 0 && ((uintptr_t)srcs[k] % 16 == 0));
    cs.p[k] = (const float*)srcs[k]; cs.Hs[k] = Hs[k]; cs.Ws[k] = Ws[k]; cs.C[k] = Cs[k]; cs.c0[k] = c0;
    c0 += Cs[k];
  }
  MVT_REQUIRE(ldd >= c0 && ldd % E == 0 && ((uintptr_t)dst % 16 == 0));
  const long long total = (long long)n * Hd * Wd * (c0 / E);
  MVT_REQUIRE(total < (1LL << 31));
  static const bool generic = getenv("MVT_CONCAT_GENERIC") != nullptr;  // tuning / A-B switch
  if (bf && !generic && c0 / 8 <= 256 && n <= 65535)
    hipLaunchKernelGGL(concat_resize_rows_bf16_kernel, dim3((unsigned)Hd, (unsigned)n), dim3(256), 0, mvt_stream(stream), cs,
                       (unsigned short*)dst, Hd, Wd, ldd);
  else if (bf)
    hipLaunchKernelGGL(concat_resize_kernel<1>, dim3(grid_for(total)), dim3(256), 0, mvt_stream(stream), cs, (float*)dst, n, Hd, Wd, ldd);
  else
    hipLaunchKernelGGL(concat_resize_kernel<0>, dim3(grid_for(total)), dim3(256), 0, mvt_stream(stream), cs, (float*)dst, n, Hd, Wd, ldd);
  return mvt_launch_status();
}

extern "C" int mvt_rgb_to_nhwc4(const float* rgbs, float* out, int V, int T, int H, int W, int t0, int nt, void* stream) {
  MVT_REQUIRE(rgbs && out && V > 0 && T > 0 && H > 0 && W > 0 && t0 >= 0 && nt > 0 && t0 + nt <= T);
  long long total = (long long)nt * V * H * W;
  hipLaunchKernelGGL(rgb_to_nhwc4_kernel<float>, dim3(grid_for(total)), dim3(256), 0, mvt_stream(stream), rgbs, out, V, T, H, W,
                     (long long)t0 * V, (long long)nt * V);
  return mvt_launch_status();
}

extern "C" int mvt_rgb_images_to_nhwc4(const void* rgbs, int is_u8, float* out, int V, int T, int H, int W, long long img0, int nimg,
                                       void* stream) {
  MVT_REQUIRE(rgbs && out && V > 0 && T > 0 && H > 0 && W > 0 && img0 >= 0 && nimg > 0 && img0 + nimg <= (long long)V * T);
  MVT_REQUIRE(is_u8 == 0 || is_u8 == 1);
  long long total = (long long)nimg * H * W;
  if (is_u8)
    hipLaunchKernelGGL(rgb_to_nhwc4_kernel<unsigned char>, dim3(grid_for(total)), dim3(256), 0, mvt_stream(stream),
                       (const unsigned char*)rgbs, out, V, T, H, W, img0, (long long)nimg);
  else
    hipLaunchKernelGGL(rgb_to_nhwc4_kernel<float>, dim3(grid_for(total)), dim3(256), 0, mvt_stream(stream), (const float*)rgbs, out, V, T,
                       H, W, img0, (long long)nimg);
  return mvt_launch_status();
}

extern "C" int mvt_rgb_u8_to_nhwc4(const unsigned char* rgbs, float* out, int V, int T, int H, int W, int t0, int nt, void* stream) {
  MVT_REQUIRE(rgbs && out && V > 0 && T > 0 && H > 0 && W > 0 && t0 >= 0 && nt > 0 && t0 + nt <= T);
  long long total = (long long)nt * V * H * W;
  hipLaunchKernelGGL(rgb_to_nhwc4_kernel<unsigned char>, dim3(grid_for(total)), dim3(256), 0, mvt_stream(stream), rgbs, out, V, T, H, W,
                     (long long)t0 * V, (long long)nt * V);
  return mvt_launch_status();
}

extern "C" int mvt_resize_nearest(const float* in, float* out, long long planes, int Hi, int Wi, int Ho, int Wo, void* stream) {
  MVT_REQUIRE(in && out && planes > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0);
  hipLaunchKernelGGL(resize_nearest_kernel, dim3(grid_for(planes * Ho * Wo)), dim3(256), 0, mvt_stream(stream), in, out, planes, Hi,
                     Wi, Ho, Wo);
  return mvt_launch_status();
}

extern "C" int mvt_instnorm_stats(const void* x, int ldx, double* partial, float* mean_rstd, int n, long long HW, int C,
                                  int io_flags, void* stream) {
  MVT_REQUIRE((io_flags & ~MVT_IO_IN_BF16) == 0);
  MVT_REQUIRE(x && partial && mean_rstd && n > 0 && HW > 0 && C > 0 && C % 4 == 0 && C <= 256 && ldx >= C && ldx % 4 == 0);
  MVT_REQUIRE(256 % (C / 4) == 0 || C / 4 <= 64);
  hipLaunchKernelGGL(instnorm_partial_kernel, dim3(MVT_IN_SLABS, n), dim3(256), 0, mvt_stream(stream), (const float*)x, ldx, partial,
                     HW, C, io_flags & MVT_IO_IN_BF16 ? 1 : 0);
  long long total = (long long)n * C;
  hipLaunchKernelGGL(instnorm_finish_kernel, dim3((unsigned)mvt_cdiv(total, 256)), dim3(256), 0, mvt_stream(stream), partial,
                     mean_rstd, HW, C, total);
  return mvt_launch_status();
}

extern "C" int mvt_instnorm_finish_slots(const float* partial, int slots, float* mean_rstd, int n, long long HW, int C, void* stream) {
  MVT_REQUIRE(partial && mean_rstd && slots > 0 && n > 0 && HW > 0 && C > 0);
  hipLaunchKernelGGL(instnorm_finish_slots_kernel, dim3((unsigned)mvt_cdiv(C, FC), (unsigned)n), dim3(1024), 0, mvt_stream(stream),
                     partial, slots, mean_rstd, HW, C);
  return mvt_launch_status();
}

extern "C" int mvt_instnorm_apply(const void* x, const float* mean_rstd, const void* skip, const float* skip_stats, void* y,
                                  int n, long long HW, int C, int io_flags, void* stream) {
  MVT_REQUIRE(x && mean_rstd && y && n > 0 && HW > 0 && C > 0 && C % 4 == 0);
  const int both = MVT_IO_IN_BF16 | MVT_IO_OUT_BF16;
  const int skip_relu = io_flags & MVT_APPLY_SKIP_RELU ? 1 : 0;
  io_flags &= ~MVT_APPLY_SKIP_RELU;
  MVT_REQUIRE(io_flags == 0 || io_flags == both);  // x, skip and y share one element type
  MVT_REQUIRE(!skip_relu || skip_stats);
  MVT_REQUIRE(skip || !skip_stats);
  static const bool generic = getenv("MVT_APPLY_GENERIC") != nullptr;  // tuning / A-B switch
  if (io_flags && C % 8 == 0 && C <= 2048 && HW < (1LL << 31) && !generic && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) &&
      ((uintptr_t)skip % 16 == 0)) {
    // pixels per block: ~8 waves of blocks per CU-slot keeps the tail short; at least one full sweep of the block's threads
    const int pstep = 256 / (C / 8);
    long long ppb = (HW * n + 8191) / 8192;
    ppb = (ppb + pstep * 4 - 1) / (pstep * 4) * (pstep * 4);
    const dim3 grid((unsigned)mvt_cdiv(HW, ppb), (unsigned)n);
    const unsigned short *xs = (const unsigned short*)x, *ks = (const unsigned short*)skip;
    unsigned short* ys = (unsigned short*)y;
    if (!skip)
      hipLaunchKernelGGL((instnorm_apply_bf16_kernel<false, false>), grid, dim3(256), 0, mvt_stream(stream), xs, mean_rstd, ks, skip_stats, ys,
                         (int)HW, C, (int)ppb, skip_relu);
    else if (!skip_stats)
      hipLaunchKernelGGL((instnorm_apply_bf16_kernel<true, false>), grid, dim3(256), 0, mvt_stream(stream), xs, mean_rstd, ks, skip_stats, ys,
                         (int)HW, C, (int)ppb, skip_relu);
    else
      hipLaunchKernelGGL((instnorm_apply_bf16_kernel<true, true>), grid, dim3(256), 0, mvt_stream(stream), xs, mean_rstd, ks, skip_stats, ys,
                         (int)HW, C, (int)ppb, skip_relu);
    return mvt_launch_status();
  }
  long long total4 = (long long)n * HW * (C / 4);
  hipLaunchKernelGGL(instnorm_apply_kernel, dim3(grid_for(total4)), dim3(256), 0, mvt_stream(stream), (const float*)x, mean_rstd,
                     (const float*)skip, skip_stats, (float*)y, HW, C, total4, io_flags ? 1 : 0, skip_relu);
  return mvt_launch_status();
}

extern "C" int mvt_resize_bilinear_ac(const void* src, void* dst, int n, int Hs, int Ws, int C, int Hd, int Wd, int ldd,
                                      int c_off, int io_flags, void* stream) {
  MVT_REQUIRE(src && dst && n > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && C > 0 && C % 4 == 0);
  MVT_REQUIRE(io_flags == 0 || io_flags == (MVT_IO_IN_BF16 | MVT_IO_OUT_BF16));
  MVT_REQUIRE(ldd % 4 == 0 && c_off % 4 == 0 && c_off >= 0 && c_off + C <= ldd);
  long long total = (long long)n * Hd * Wd * (C / 4);
  hipLaunchKernelGGL(resize_bilinear_ac_kernel, dim3(grid_for(total)), dim3(256), 0, mvt_stream(stream), (const float*)src, (float*)dst,
                     n, Hs, Ws, C, Hd, Wd, ldd, c_off, io_flags ? 1 : 0);
  return mvt_launch_status();
}
