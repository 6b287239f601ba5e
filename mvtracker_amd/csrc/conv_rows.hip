// 3x3 / stride 1 / pad 1 convolution of the encoder on the bf16 matrix cores, second generation
// (reference: the ResidualBlock convs and BasicEncoder.conv2, spatracker/blocks.py:84-128, 246-282).
//
// The first halo kernel (gemm.hip) spent ~17 VALU + 8 SALU instructions per MFMA on branchy loaders and 64-bit
// epilogue addressing and ran at 14-24 % of the MFMA peak.  This one is laid out so that the inner loop is nothing but
// ds_read_b128 (immediate offsets) and MFMAs:
//   * a workgroup owns TR = 8 image rows x 32 columns of output pixels (256 px) and BN output channels; a 32-pixel MFMA
//     row block is one image row segment, so the A fragment of tap (kh, kw) is the patch row shifted by a constant;
//   * the input patch (10 x 34 pixels x 32 channels, bf16, 80-B pixel slots: conflict-free ds_read_b128) is staged once per
//     32-channel chunk -- InstanceNorm + ReLU of the producer applied on the way when in_stats is given -- and the
//     weights one filter row (3 taps) at a time; both come through registers (global loads of the next stage are in
//     flight during the MFMAs of the current one) with branch-free clamped addressing;
//   * wave tiles are TM x TN MFMA blocks (2x2, 2x3 or 4x2), i.e. 1 - 0.75 KiB of LDS reads per MFMA;
//   * the epilogue adds the bias, writes fp32 or bf16 and emits the per-channel (sum, sum of squares) of every 32-pixel
//     row segment for the fused InstanceNorm statistics.
#include <stdlib.h>

#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// Diagnostic build only (-DMVT_STAMPS, tools/stamp_conv.py): s_memtime at the phase boundaries of two workgroups, one slot per
// (workgroup, wave, stamp), in a buffer of its own.  No stamp executes in the shipped library.
#ifdef MVT_STAMPS
#ifndef MVT_STAMP_WG
#define MVT_STAMP_WG 2000
#endif
__device__ unsigned long long mvt_conv_stamp_buf[2 * 8 * 128];
#define CSTAMP(i)                                                                                                  \
  do {                                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    unsigned long long t_;                                                                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    if ((threadIdx.x & 63) == 0 && (blockIdx.x == 0 || blockIdx.x == MVT_STAMP_WG) && (i) < 128)                           \
      mvt_conv_stamp_buf[((blockIdx.x ? 1 : 0) * 8 + (threadIdx.x >> 6)) * 128 + (i)] = t_;                        \
  } while (0)
#else
#define CSTAMP(i) \
  do {            \
  } while (0)
#endif

struct RowsArgs {
  const float* in;   // [n][H][W][Cin] fp32 or bf16
  const unsigned short* w;  // bf16 [Cout][ldw]: row = (kh, kw, cin)
  const float* bias;
  float* out;        // [n][H][W][ldo] fp32 or bf16
  const float* in_stats;  // [n][Cin][2] or null
  float* out_part;        // [n][slots][Cout][2] or null
  int H, W, Cin, Cout, ldw, ldo, slots, in_bf16, out_bf16;
  // stem only: the input is read straight from the clip's planar frames (V,T,3,H,W), fp32 or uint8 in [0, 255] -- image i of the call is
  // view (rgb_img0 + i) % V of frame (rgb_img0 + i) / V -- and normalised on the way (2 (x / 255) - 1, mvtracker.py:455; the
  // arithmetic of rgb_to_nhwc4_kernel, whose [n][H][W][4] fp32 staging tensor and launch this replaces)
  const void* rgb;
  int rgb_u8, rgb_V, rgb_T;
  long long rgb_img0;
  // fused downsample branch (DS, the stride-2 3x3 kernel): the ResidualBlock's 1x1 / stride-2 convolution of the SAME input
  // (blocks.py:112-128) = the centre tap of every output pixel with its own weights [Cout][ldw2], bias, output and statistics
  const unsigned short* w2;
  const float* bias2;
  float* out2;
  float* out_part2;
  int ldw2;
};

constexpr int TR = 8;            // output rows per workgroup (stem and the stride-1 3x3 kernels)
constexpr int TC = 32;           // output columns per workgroup
constexpr int CK = 32;           // channels per chunk
constexpr int LDP = CK + 8;      // bf16 per pixel / weight-row slot (80 B)
constexpr int MAXC = 512;        // input channels the in-kernel normalisation supports

// Shared epilogue of the row-tile kernels: bias, output (fp32 / bf16), per-32-pixel statistics.  acc[i][j] is the 32 x 32
// block (row i of this wave, 32-channel block j); yw = first output row of the wave.
// Elements of the per-wave LDS staging tile of the bf16 epilogue: 32 pixels x (BN + 8) channels
template <int TN> constexpr int stage_elems() { return 32 * (TN * 32 + 8); }

// Statistics: every (tile row, channel) leaves its (sum, sum of squares) over the row's 32 pixels in the workgroup's LDS table
// wst[tile row][BN][2] (pre-zeroed; lrow0 = first tile row of this wave); the kernel adds the rows up in a fixed order and writes
// ONE slot per workgroup (tile) -- eight times fewer partials to write and to reduce than one slot per row segment.
template <int TM, int TN, bool QUADS, bool STAGED, int WLD = TN * 32>
__device__ __forceinline__ void epilogue_rows(const f32x16 (&acc)[TM][TN], const RowsArgs& p, long long img, int Ho, int Wo, int yw, int x0,
                                              int n0, int tiles_x, int tx, int lane, unsigned short* stage, float* wst, int lrow0) {
  constexpr int BNS = WLD;  // channels per row of the statistics table (the workgroup's channel count)
  const int r = lane & 31, h = lane >> 5;
#ifndef MVT_EPI_NO_TR
  if constexpr (STAGED && TN * 32 * 36 <= stage_elems<TN>()) {  // (96-channel tiles: the transposed tile does not fit their staging area)
    // Round 3: bf16 output through a TRANSPOSED staging tile.  The accumulator layout has the channel on the lane and four runs of
    // four consecutive pixels in the registers: as [pixel][channel] staging every value was its own 2-byte LDS store (16 per 32 x 32
    // block and lane; in-kernel stamps: the epilogue of a 64-channel tile cost as much as its MFMAs).  Staged as [channel][pixel]
    // (72-B rows: conflict-free) a lane stores each run of four pixels as ONE 8-byte word -- 4 stores per block -- and the tile is read
    // back with gfx950's transposing ds_read_b64_tr_b16 (per 16-lane group: 4 channel rows x 16 pixels, delivered to lane i as the
    // four channels of pixel i): two such reads give a lane eight consecutive channels of one pixel = one 16-byte global store.
    // Same values, same rounding, same statistics as the [pixel][channel] form: bit-identical outputs.
    // The wave-level fences around the staging tile are scoped to LDS ("local"): unscoped they also wait for the row's GLOBAL
    // stores to complete (s_waitcnt vmcnt(0), a full write round trip per tile row) before the next row may touch the tile.
    constexpr int BN = TN * 32, LDTT = 36;   // staging row = 32 pixels + 4 pad (bf16)
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    const bool interior = x0 + TC <= Wo;
    // Round 4: written as `if (valid) { s1 += v; s2 = fma(v, v, s2); }` per element, every accumulator register became its own
    // divergent branch (EXEC save / restore around two instructions: 170-255 instructions per 32 x 32 block, a quarter of a
    // 64-channel tile's life).  A tile that lies inside the image with all its channels -- wave-uniform, nearly every tile -- takes no
    // predicate at all; the others select 0 into the sums (s + 0 and fma(0, 0, s) leave s unchanged: same statistics).
    const bool full = interior && n0 + TN * 32 <= p.Cout;
    const int gq = lane >> 4, li = lane & 15;  // 16-lane group, lane in group
    // the biases of all channel blocks are requested up front: one memory round trip per tile instead of one per 32 x 32 block
    // (a persistent workgroup has no neighbour to hide them behind)
    float bvs[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bvs[j] = (p.bias && n0 + j * 32 + r < p.Cout) ? p.bias[n0 + j * 32 + r] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int y = yw + i;  // wave-uniform
      if (y >= Ho) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 32 + r;
        const bool nok = n < p.Cout;
        const float bv = bvs[j];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v4;
#pragma unroll
          for (int e = 0; e < 4; ++e) v4[e] = acc[i][j][4 * g + e] + bv;
          if (full) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              s1 += v4[e];
              s2 = fmaf(v4[e], v4[e], s2);
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float vm = (nok && (interior || x0 + 8 * g + 4 * h + e < Wo)) ? v4[e] : 0.f;
              s1 += vm;
              s2 = fmaf(vm, vm, s2);
            }
          }
          *reinterpret_cast<u32x2*>(&stage[(j * 32 + r) * LDTT + 8 * g + 4 * h]) = __builtin_bit_cast(u32x2, __builtin_convertvector(v4, bf16x4));
        }
        if (p.out_part) {  // one writer per (tile row, channel): deterministic
          // (lane pair (l, l ^ 32) summed with v_permlane32_swap: the two results hold the pair's lower and upper value in every
          //  lane, their sum is the same bits as x + shfl_xor(x, 32) without the LDS round trip of a ds_bpermute)
          const auto q1 = __builtin_amdgcn_permlane32_swap(__float_as_int(s1), __float_as_int(s1), false, false);
          const auto q2 = __builtin_amdgcn_permlane32_swap(__float_as_int(s2), __float_as_int(s2), false, false);
          s1 = __int_as_float(q1[0]) + __int_as_float(q1[1]);
          s2 = __int_as_float(q2[0]) + __int_as_float(q2[1]);
          if (h == 0 && nok) {
            float* pp = wst + ((lrow0 + i) * BNS + j * 32 + r) * 2;
            pp[0] = s1;
            pp[1] = s2;
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // one 32 x 32 block at a time (VGPR budget)
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
      unsigned short* rowp = reinterpret_cast<unsigned short*>(p.out) + ((img * Ho + y) * (long long)Wo + x0) * p.ldo + n0;
      // pair k: group gq reads channel octet m = 2 k + (gq >> 1) of pixels 16 (gq & 1) .. + 15; lane 4 q + pp of a group supplies the
      // address of channel row q, pixel columns 4 pp .. 4 pp + 3 (EXEC is all ones here: no lane may be masked for the gather)
      const int ph = gq & 1;
      const unsigned short* tb = stage + (li >> 2) * LDTT + ph * 16 + (li & 3) * 4;
      // (all transposed reads of the row first, then its stores: one LDS latency per row instead of one per pair)
      s16x4 lo[BN / 16], hi[BN / 16];
#pragma unroll
      for (int k = 0; k < BN / 16; ++k) {
        const int m = 2 * k + (gq >> 1);
        lo[k] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tb + (8 * m) * LDTT));
        hi[k] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tb + (8 * m + 4) * LDTT));
      }
      __builtin_amdgcn_sched_barrier(0);
      const int px = ph * 16 + li;
      unsigned short* lanep = rowp + (long long)px * p.ldo;
#pragma unroll
      for (int k = 0; k < BN / 16; ++k) {
        const int m = 2 * k + (gq >> 1);
        if (full || (x0 + px < Wo && n0 + 8 * m < p.Cout)) {
          const u32x2 a = __builtin_bit_cast(u32x2, lo[k]), b2 = __builtin_bit_cast(u32x2, hi[k]);
          *reinterpret_cast<u32x4*>(lanep + 8 * m) = (u32x4){a[0], a[1], b2[0], b2[1]};
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
    }
    return;
  }
#endif
  if (STAGED) {  // (compile-time: with both paths in one kernel the accumulators are copied out ahead of the branch -> +50 VGPRs)
    // bf16 output through LDS: the accumulator layout (channel on the lane, 16 pixels in the registers) would store 2-4 bytes
    // per lane; staged as [pixel][channel], every lane stores 16 contiguous bytes and a pixel's channels leave as whole
    // 128-B lines (the scattered stores were still a quarter of the 64-channel layers' time).  `stage` is wave-private.
    constexpr int BN = TN * 32, LDT = BN + 8, PPX = BN / 8;  // row stride (bf16), 16-B pieces per pixel
    const bool interior = x0 + TC <= Wo;
    const bool full = interior && n0 + TN * 32 <= p.Cout;  // (wave-uniform: no per-element predicates, see the transposed form above)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int y = yw + i;  // wave-uniform
      if (y >= Ho) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 32 + r;
        const bool nok = n < p.Cout;
        const float bv = (p.bias && nok) ? p.bias[n] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int dx = (e & 3) + 8 * (e >> 2) + 4 * h;
          const float v = acc[i][j][e] + bv;
          stage[dx * LDT + j * 32 + r] = mvt_bf16_bits(v);
          const float vm = (full || (nok && (interior || x0 + dx < Wo))) ? v : 0.f;
          s1 += vm;
          s2 = fmaf(vm, vm, s2);
          if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // four accumulator registers at a time (VGPR budget)
        }
        if (p.out_part) {  // one writer per (tile row, channel): deterministic
          s1 += __shfl_xor(s1, 32, 64);
          s2 += __shfl_xor(s2, 32, 64);
          if (h == 0 && nok) {
            float* pp = wst + ((lrow0 + i) * BNS + j * 32 + r) * 2;
            pp[0] = s1;
            pp[1] = s2;
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // one 32 x 32 block at a time: a single VGPR decides the occupancy of these kernels
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");  // the tile is written as 16-bit elements and read as 128-bit vectors
      unsigned short* rowp = reinterpret_cast<unsigned short*>(p.out) + ((img * Ho + y) * (long long)Wo + x0) * p.ldo + n0;
#pragma unroll
      for (int k = 0; k < (32 * PPX + 63) / 64; ++k) {
        const int pc = lane + 64 * k;
        const int px = pc / PPX, c8 = pc - px * PPX;
        if (pc < 32 * PPX && x0 + px < Wo && n0 + c8 * 8 < p.Cout)
          *reinterpret_cast<u32x4*>(rowp + (long long)px * p.ldo + c8 * 8) = *reinterpret_cast<const u32x4*>(&stage[px * LDT + c8 * 8]);
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
    }
    return;
  }
  if (STAGED) return;
  const bool interior = x0 + TC <= Wo;
  const int esz = p.out_bf16 ? 2 : 4;
  const bool pairs = ((p.Cout | p.ldo) & 1) == 0;  // lane pairs (n, n+1) are both valid or both invalid, rows stay 4-B aligned
  const int odd = lane & 1;
  // (the quad transpose costs a few registers: it is compiled in only where it does not lower the occupancy)
  const bool quads = QUADS && ((p.Cout | p.ldo) & 3) == 0 && ((uintptr_t)p.out & 15) == 0;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + j * 32 + r;
    const bool nok = n < p.Cout;
    const float bv = (p.bias && nok) ? p.bias[n] : 0.f;
    const int lane_off = (4 * h * p.ldo + n) * esz;  // bytes
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int y = yw + i;  // wave-uniform
      if (y >= Ho) continue;
      char* rowp = reinterpret_cast<char*>(p.out) + (((img * Ho + y) * (long long)Wo + x0) * p.ldo) * esz;  // wave-uniform
      float s1 = 0.f, s2 = 0.f;
      if (quads) {
        // 4 x 4 transpose inside every lane quad: registers 4g .. 4g+3 are four neighbouring pixels of channel n; after two
        // DPP exchanges lane k of the quad owns channels (n & ~3) .. +3 of pixel 4h + 8g + k: one 8-B (bf16) / 16-B (fp32) store
        const int k4 = lane & 3, k2 = (lane >> 1) & 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[i][j][4 * g + e] + bv;
            const float vm = (nok && (interior || x0 + 4 * h + 8 * g + e < Wo)) ? v[e] : 0.f;  // (a select, not a branch per element)
            s1 += vm;
            s2 = fmaf(vm, vm, s2);
          }
          float lo[2], hi[2];  // per pixel pair pp (pixels 2pp + odd): channels (n & ~1), (n & ~1) + 1
#pragma unroll
          for (int pp = 0; pp < 2; ++pp) {
            const float send = odd ? v[2 * pp] : v[2 * pp + 1];
            const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0xB1, 0xF, 0xF, true));  // [1,0,3,2]
            lo[pp] = odd ? recv : v[2 * pp];
            hi[pp] = odd ? v[2 * pp + 1] : recv;
          }
          // lanes 0,1 of the quad keep pair 0 (pixels 0,1) and send pair 1; lanes 2,3 keep pair 1 (pixels 2,3) and send pair 0
          const float slo = k2 ? lo[0] : lo[1], shi = k2 ? hi[0] : hi[1];
          const float rlo = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(slo), 0x4E, 0xF, 0xF, true));  // [2,3,0,1]
          const float rhi = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(shi), 0x4E, 0xF, 0xF, true));
          const float klo = k2 ? lo[1] : lo[0], khi = k2 ? hi[1] : hi[0];
          // channels in order: the pair that came from the lanes with bit 1 clear first
          const float c0 = k2 ? rlo : klo, c1 = k2 ? rhi : khi, c2 = k2 ? klo : rlo, c3 = k2 ? khi : rhi;
          const int dx = 8 * g + k4;
          if (nok && (interior || x0 + 4 * h + dx < Wo)) {
            char* q = rowp + dx * p.ldo * esz + lane_off - k4 * esz;
            if (p.out_bf16) {
              uint2 w;
              w.x = (unsigned)mvt_bf16_bits(c0) | ((unsigned)mvt_bf16_bits(c1) << 16);
              w.y = (unsigned)mvt_bf16_bits(c2) | ((unsigned)mvt_bf16_bits(c3) << 16);
              *reinterpret_cast<uint2*>(q) = w;
            } else {
              *reinterpret_cast<f32x4*>(q) = (f32x4){c0, c1, c2, c3};
            }
          }
        }
      } else if (pairs) {
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          const int dx = (e & 3) + 8 * (e >> 2);
          const float v0 = acc[i][j][e] + bv, v1 = acc[i][j][e + 1] + bv;
          const bool ok0 = nok && (interior || x0 + 4 * h + dx < Wo), ok1 = nok && (interior || x0 + 4 * h + dx + 1 < Wo);
          const float m0 = ok0 ? v0 : 0.f, m1 = ok1 ? v1 : 0.f;
          s1 += m0;
          s2 = fmaf(m0, m0, s2);
          s1 += m1;
          s2 = fmaf(m1, m1, s2);
          // even lane sends v1 (pixel e+1) and keeps v0; odd lane sends v0 and keeps v1
          const float send = odd ? v0 : v1;
          const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
          const float lo = odd ? recv : v0, hi = odd ? v1 : recv;  // channels (n & ~1, (n & ~1) + 1) of pixel dx + odd
          if (odd ? ok1 : ok0) {
            char* q = rowp + (dx + odd) * p.ldo * esz + lane_off - odd * esz;
            if (p.out_bf16) *reinterpret_cast<unsigned*>(q) = (unsigned)mvt_bf16_bits(lo) | ((unsigned)mvt_bf16_bits(hi) << 16);
            else *reinterpret_cast<float2*>(q) = make_float2(lo, hi);
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int dx = (e & 3) + 8 * (e >> 2);
          if (nok && (interior || x0 + 4 * h + dx < Wo)) {
            const float v = acc[i][j][e] + bv;
            char* q = rowp + dx * p.ldo * esz + lane_off;
            if (p.out_bf16) *reinterpret_cast<unsigned short*>(q) = mvt_bf16_bits(v);
            else *reinterpret_cast<float*>(q) = v;
            s1 += v;
            s2 = fmaf(v, v, s2);
          }
        }
      }
      if (p.out_part) {  // one writer per (tile row, channel): deterministic
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (h == 0 && nok) {
          float* pp = wst + ((lrow0 + i) * BNS + j * 32 + r) * 2;
          pp[0] = s1;
          pp[1] = s2;
        }
      }
    }
  }
}

// After the epilogue (and a barrier): channel c of the tile = sum of its rows' entries in row order -> the workgroup's slot.
template <int ROWS, int BN>
__device__ __forceinline__ void write_tile_stats(const RowsArgs& p, const float* wst, long long img, int tile, int n0, int t) {
  if (p.out_part && t < BN && n0 + t < p.Cout) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
      s1 += wst[(rr * BN + t) * 2];
      s2 += wst[(rr * BN + t) * 2 + 1];
    }
    float* pp = p.out_part + ((img * p.slots + tile) * p.Cout + n0 + t) * 2;
    pp[0] = s1;
    pp[1] = s2;
  }
}

// Geometry of one workgroup: TM row blocks per wave, four waves stacked in y, kernel size KS (1 or 3), stride S (1 or 2).
// The patch is stored row by row in 80-B pixel slots.  For the stride-2 3x3 kernel a patch row holds its even columns
// first and its odd columns after them, so that the pixels 2r + kw of tap kw are consecutive slots again.
template <int TM, int KS, int S, int NW> struct Geo {
  static constexpr int ROWS = NW * TM;                                 // output rows (NW waves stacked in y)
  static constexpr int PR = KS == 1 ? ROWS : S * (ROWS - 1) + 3;       // patch rows
  static constexpr int PC = KS == 1 ? TC : S * (TC - 1) + 3;           // patch columns
  static constexpr int HALF = (PC + 1) / 2;                            // even columns of a split row
  static constexpr bool SPLITROW = KS == 3 && S == 2;
  static constexpr int RS = SPLITROW ? 2 * HALF : PC;                  // slots per patch row
  static constexpr int NSLOT = PR * RS;
  static constexpr int PAD = KS == 3 ? 1 : 0;
  static constexpr int PSTEP = KS == 1 ? S : 1;                        // input pixels per patch pixel (1x1: only the sampled ones)
  __host__ __device__ static constexpr int tap_col(int kw) { return SPLITROW ? (kw & 1) * HALF + (kw >> 1) : kw; }
  __host__ __device__ static constexpr int row_step() { return KS == 1 ? 1 : S; }  // patch rows per output row
};

// (second launch bound = minimum waves per SIMD: the 64-channel bf16 tiles sit one register above the three-workgroup limit)
template <int TM, int TN, int KS, int S, bool INB, int NW, bool STAGED, bool DS = false>
__global__ __launch_bounds__(64 * NW, DS ? 2 : ((TM == 2 && TN == 2 && INB && NW == 4) ? 3 : 1)) void conv_rows_bf16(RowsArgs p) {
  static_assert(!DS || (KS == 3 && S == 2 && INB && STAGED), "the fused downsample branch rides on the stride-2 3x3 kernel");
  using G = Geo<TM, KS, S, NW>;
  constexpr int NT = 64 * NW;
  constexpr int BN = TN * 32;
  constexpr int ppp = INB ? 4 : 8;   // 16-B pieces per pixel
  constexpr int cpp = INB ? 8 : 4;   // channels per piece
  constexpr int NPL = (G::NSLOT * ppp + NT - 1) / NT;    // pieces per thread per chunk
  constexpr int NWF = (KS * BN * 4 + NT - 1) / NT;        // 16-B weight pieces per thread per stage (KS taps x BN rows x 64 B)
  __shared__ __attribute__((aligned(16))) unsigned short Ps[G::NSLOT * LDP];
  __shared__ __attribute__((aligned(16))) unsigned short Ws[KS * BN * LDP];
  static_assert(!STAGED || NW * stage_elems<TN>() <= G::NSLOT * LDP, "the bf16 epilogue's staging tiles reuse the patch LDS");
  __shared__ __attribute__((aligned(16))) float Sst[2 * MAXC];  // (mean, rstd) of the input channels of this image
  __shared__ float wst[G::ROWS * BN * 2];                       // per (tile row, channel) output statistics

  const int t = threadIdx.x;
  const int Ho = (p.H + 2 * G::PAD - KS) / S + 1, Wo = (p.W + 2 * G::PAD - KS) / S + 1;
  const int tiles_x = (Wo + TC - 1) / TC, tiles_y = (Ho + G::ROWS - 1) / G::ROWS;
  const int tiles_n = (p.Cout + BN - 1) / BN;
  // XCD-aware tile order (cdna_hip_programming.md T1): consecutive workgroup ids go round-robin to the 8 XCDs, each with its own
  // L2.  Tiles that share input -- the Cout tiles of one pixel tile read the same patch, neighbouring pixel tiles share their halo
  // rows / columns -- are consecutive in TILE order, so each XCD is given a contiguous run of tiles (bijective for any grid size).
  int b;
  {
    const unsigned nwg = gridDim.x, q = nwg / 8, rr = nwg % 8, xcd = blockIdx.x % 8;
    b = (int)((xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + blockIdx.x / 8);
  }
  const int tn = b % tiles_n; b /= tiles_n;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y;
  const long long img = b / tiles_y;
  const int y0 = ty * G::ROWS, x0 = tx * TC, n0 = tn * BN;
  const long long in_img = img * (long long)p.H * p.W * p.Cin;

  // ---- loader state (branch-free: out-of-image pixels read a clamped address and are zeroed by a select)
  int pg[NPL];              // element offset of the piece inside the image (clamped), without the chunk offset
  bool pk[NPL];             // inside the image
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int f = t + NT * i;
    const int sl = f / ppp, q = f - sl * ppp;
    const int py = sl / G::RS, cs = sl - py * G::RS;
    const int px = G::SPLITROW ? (cs < G::HALF ? 2 * cs : 2 * (cs - G::HALF) + 1) : cs;
    const int gy = (KS == 1 ? S * (y0 + py) : S * y0 - G::PAD + py), gx = (KS == 1 ? S * (x0 + px) : S * x0 - G::PAD + px);
    pk[i] = sl < G::NSLOT && px < G::PC && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    const int cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);
    pg[i] = (cy * p.W + cx) * p.Cin + q * cpp;
  }
  int wg[NWF];
  bool wk[NWF];
#pragma unroll
  for (int i = 0; i < NWF; ++i) {
    const int u = t + NT * i;
    const int row = u >> 2, part = u & 3;
    const int tap = row / BN, n = row - tap * BN;
    wk[i] = row < KS * BN && n0 + n < p.Cout;
    wg[i] = min(n0 + n, p.Cout - 1) * p.ldw + tap * p.Cin + part * 8;
  }

  f32x4 rp[NPL];   // fp32 tensors: 4 channels; bf16 tensors: 8 packed channels (bit pattern)
  u32x4 rw[NWF];
  // (addresses = wave-uniform 64-bit base + UNSIGNED 32-bit lane offset in bytes: the saddr form of global_load.  As signed element
  //  offsets every pg[i] / wg[i] was sign-extended into a register PAIR that lived through the whole main loop)
  auto load_patch = [&](int c0) {
    if (INB) {
      const char* base = reinterpret_cast<const char*>(reinterpret_cast<const unsigned short*>(p.in) + in_img + c0);
#pragma unroll
      for (int i = 0; i < NPL; ++i) rp[i] = *reinterpret_cast<const f32x4*>(base + (unsigned)(pg[i] * 2));
    } else {
      const char* base = reinterpret_cast<const char*>(p.in + in_img + c0);
#pragma unroll
      for (int i = 0; i < NPL; ++i) rp[i] = *reinterpret_cast<const f32x4*>(base + (unsigned)(pg[i] * 4));
    }
  };
  auto norm4 = [&](f32x4 v, const f32x4& m, const f32x4& rs) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf((v[e] - m[e]) * rs[e], 0.f);
    return v;
  };
  auto store_patch = [&](int c0) {
    f32x4 sm0 = {0.f, 0.f, 0.f, 0.f}, sr0 = sm0, sm1 = sm0, sr1 = sm0;
    if (p.in_stats) {
      const float* sp = &Sst[(c0 + (t % ppp) * cpp) * 2];
      const f32x4 a = *reinterpret_cast<const f32x4*>(sp), b4 = *reinterpret_cast<const f32x4*>(sp + 4);
      sm0 = (f32x4){a[0], a[2], b4[0], b4[2]};
      sr0 = (f32x4){a[1], a[3], b4[1], b4[3]};
      if (INB) {
        const f32x4 c4 = *reinterpret_cast<const f32x4*>(sp + 8), d4 = *reinterpret_cast<const f32x4*>(sp + 12);
        sm1 = (f32x4){c4[0], c4[2], d4[0], d4[2]};
        sr1 = (f32x4){c4[1], c4[3], d4[1], d4[3]};
      }
    }
    if (INB) {
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int f = t + NT * i, sl = f / ppp;
        if (sl >= G::NSLOT) continue;
        u32x4 w = __builtin_bit_cast(u32x4, rp[i]);
        if (p.in_stats) {
#ifndef MVT_NO_PACKED_NORM
          // Round 4 (the lever round 2 had to revert for 4-9 spilled registers; the loader's 32-bit lane offsets freed them): per
          // 32-bit word = two channels, packed fp32 arithmetic -- v_pk_add_f32 with (-mean), v_pk_mul_f32 with rstd: the SAME two
          // roundings as (x - mean) * rstd --, one v_cvt_pk_bf16_f32, and the ReLU as a packed 16-bit INTEGER max with 0 on the
          // bf16 pair (a negative bf16 is a negative int16; rounding commutes with max(., 0)): 24 instead of 36 vector
          // instructions per 16-byte piece, bit-identical results for every non-NaN input.
          auto pair = [](unsigned ww, float nm0, float nm1, float r0, float r1) -> unsigned {
            f32x2 v = (f32x2){__uint_as_float(ww << 16), __uint_as_float(ww & 0xFFFF0000u)};
            v = (v + (f32x2){nm0, nm1}) * (f32x2){r0, r1};
            s16x2 q = __builtin_bit_cast(s16x2, __builtin_convertvector(v, bf16x2));
            q = __builtin_elementwise_max(q, (s16x2){0, 0});
            return __builtin_bit_cast(unsigned, q);
          };
          w = (u32x4){pair(w[0], -sm0[0], -sm0[1], sr0[0], sr0[1]), pair(w[1], -sm0[2], -sm0[3], sr0[2], sr0[3]),
                      pair(w[2], -sm1[0], -sm1[1], sr1[0], sr1[1]), pair(w[3], -sm1[2], -sm1[3], sr1[2], sr1[3])};
#else
          f32x4 lo = (f32x4){__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xFFFF0000u), __uint_as_float(w[1] << 16),
                             __uint_as_float(w[1] & 0xFFFF0000u)};
          f32x4 hi = (f32x4){__uint_as_float(w[2] << 16), __uint_as_float(w[2] & 0xFFFF0000u), __uint_as_float(w[3] << 16),
                             __uint_as_float(w[3] & 0xFFFF0000u)};
          const u32x2 a = __builtin_bit_cast(u32x2, __builtin_convertvector(norm4(lo, sm0, sr0), bf16x4));
          const u32x2 c = __builtin_bit_cast(u32x2, __builtin_convertvector(norm4(hi, sm1, sr1), bf16x4));
          w = (u32x4){a[0], a[1], c[0], c[1]};
#endif
        }
        if (!pk[i]) w = (u32x4){0u, 0u, 0u, 0u};  // zero padding applies after the normalisation
        *reinterpret_cast<u32x4*>(&Ps[sl * LDP + (f - sl * ppp) * cpp]) = w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int f = t + NT * i, sl = f / ppp;
        if (sl >= G::NSLOT) continue;
        f32x4 v = rp[i];
        if (p.in_stats) v = norm4(v, sm0, sr0);
        u32x2 w = __builtin_bit_cast(u32x2, __builtin_convertvector(v, bf16x4));
        if (!pk[i]) w = (u32x2){0u, 0u};
        *reinterpret_cast<u32x2*>(&Ps[sl * LDP + (f - sl * ppp) * cpp]) = w;
      }
    }
  };
  // (64-channel tiles of 256 threads: piece i of a thread is tap i of the same weight row -- one offset register instead of NWF)
  constexpr bool WREG1 = BN == 64 && NT == 256 && NWF == KS;
  auto load_w = [&](int c0, int kh) {
    const int base = kh * KS * p.Cin + c0;
#pragma unroll
    for (int i = 0; i < NWF; ++i)
      rw[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.w + base + (WREG1 ? i * p.Cin : 0)) + (unsigned)(wg[WREG1 ? 0 : i] * 2));
  };
  auto store_w = [&]() {
#pragma unroll
    for (int i = 0; i < NWF; ++i) {
      const int u = t + NT * i, row = u >> 2;
      if (row >= KS * BN) continue;
      *reinterpret_cast<u32x4*>(&Ws[row * LDP + (u & 3) * 8]) = wk[WREG1 ? 0 : i] ? rw[i] : (u32x4){0u, 0u, 0u, 0u};
    }
  };

  // the first patch chunk and weight stage are requested before anything waits: their round trip passes under the statistics
  // staging and its barrier
  load_patch(0);
  load_w(0, 0);
  // InstanceNorm statistics: staged in LDS once (they would otherwise occupy 16 registers through the MFMA loop)
  if (p.in_stats) {
    for (int i = t; i < 2 * p.Cin; i += NT) Sst[i] = p.in_stats[img * 2 * p.Cin + i];
    __syncthreads();
  }
  const int lane = t & 63;
  const int wm = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, h = lane >> 5;
  // lane bases; everything else in the inner loop is a compile-time offset
  const unsigned short* Pl = Ps + (G::row_step() * wm * TM * G::RS + r) * LDP + h * 8;
  const unsigned short* Wl = Ws + r * LDP + h * 8;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // fused downsample branch: its own accumulators; one weight piece per thread and chunk ([BN rows][32 channels] of W2)
  constexpr int NWD = DS ? (BN * 4 + NT - 1) / NT : 1;
  f32x16 dacc[DS ? TM : 1][DS ? TN : 1];
  u32x4 rwd[NWD];
  if constexpr (DS) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) dacc[i][j][e] = 0.f;
  }

  for (int i = t; i < G::ROWS * BN * 2; i += NT) wst[i] = 0.f;  // (rows past the image stay zero; the loop's barriers order this)
  const int nchunk = p.Cin / CK;
  CSTAMP(0);
  for (int c = 0; c < nchunk; ++c) {
    CSTAMP(1 + 8 * c);
    store_patch(c * CK);  // (the barrier that ended the previous chunk's last stage made the patch free)
    if constexpr (DS) {  // this chunk's slice of the downsample weights: requested now, staged behind the three filter rows
#pragma unroll
      for (int i = 0; i < NWD; ++i) {
        const int u = t + NT * i, row = u >> 2;
        const int n = min(n0 + row, p.Cout - 1);
        rwd[i] = *reinterpret_cast<const u32x4*>(p.w2 + (long long)n * p.ldw2 + c * CK + (u & 3) * 8);
        if (row >= BN || n0 + row >= p.Cout) rwd[i] = (u32x4){0u, 0u, 0u, 0u};
      }
    }
    CSTAMP(2 + 8 * c);
    const bool more = c + 1 < nchunk;
#pragma unroll
    for (int kh = 0; kh < KS; ++kh) {
      store_w();
      __syncthreads();
      CSTAMP(3 + 8 * c + 2 * kh);
      if (kh + 1 < KS) {
        load_w(c * CK, kh + 1);
      } else if (more) {
        load_w((c + 1) * CK, 0);
        load_patch((c + 1) * CK);
      }
#ifdef MVT_ROWS_PIPE
      // Round 4: the fragments of group g + 1 (a group = one tap, one 16-wide k-step: TM + TN reads, TM x TN MFMAs) are requested
      // BEFORE the MFMAs of group g issue, in two register sets, and the order is pinned: left free, the scheduler sinks every read
      // to just ahead of its first use and each group waits out an LDS round trip (in-kernel stamps: 2 000-2 300 cycles per 24
      // MFMAs that occupy the pipe for 768).
      {
        constexpr int NG = KS * (CK / 16);
        bf16x8 a[2][TM], bb[2][TN];
        auto read_group = [&](int set, int g) {
          const int kw = g / (CK / 16), ks = g % (CK / 16);
#pragma unroll
          for (int i = 0; i < TM; ++i)
            a[set][i] = __builtin_bit_cast(
                bf16x8, *reinterpret_cast<const u32x4*>(Pl + ((G::row_step() * i + kh) * G::RS + G::tap_col(kw)) * LDP + ks * 16));
#pragma unroll
          for (int j = 0; j < TN; ++j)
            bb[set][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Wl + (kw * BN + j * 32) * LDP + ks * 16));
        };
        // (the last filter row of a chunk carries the next chunk's patch in registers -- 24 VGPRs, the kernel's pressure point at the
        //  168 of three workgroups per CU: there a group's reads are issued right ahead of its own MFMAs, one set)
        const bool AHEAD = MVT_ROWS_PIPE >= 2 || kh + 1 < KS || KS == 1;  // (compile time: kh is an unrolled loop index)
        read_group(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (AHEAD && g + 1 < NG) read_group((g + 1) & 1, g + 1);
          if (!AHEAD && g > 0) read_group(g & 1, g);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g & 1][i], bb[g & 1][j], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#else
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) {
#pragma unroll
        for (int ks = 0; ks < CK / 16; ++ks) {
          bf16x8 a[TM], bb[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i)
            a[i] = __builtin_bit_cast(
                bf16x8, *reinterpret_cast<const u32x4*>(Pl + ((G::row_step() * i + kh) * G::RS + G::tap_col(kw)) * LDP + ks * 16));
#pragma unroll
          for (int j = 0; j < TN; ++j)
            bb[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Wl + (kw * BN + j * 32) * LDP + ks * 16));
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
      }
#endif
      CSTAMP(4 + 8 * c + 2 * kh);
      __syncthreads();
    }
    if constexpr (DS) {
      // The downsample branch: the 1x1 / stride-2 convolution samples input pixel (2y, 2x) = tap (kh, kw) = (1, 1) of the 3x3 / stride-2
      // window around output (y, x), which is already in the patch.  One more stage per chunk on the weight buffer's first BN rows;
      // same k order as the stand-alone 1x1 kernel (chunk, k-step): identical accumulators.
#pragma unroll
      for (int i = 0; i < NWD; ++i) {
        const int u = t + NT * i, row = u >> 2;
        if (row < BN) *reinterpret_cast<u32x4*>(&Ws[row * LDP + (u & 3) * 8]) = rwd[i];
      }
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < CK / 16; ++ks) {
        bf16x8 a[TM], bb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Pl + ((G::row_step() * i + 1) * G::RS + G::tap_col(1)) * LDP + ks * 16));
#pragma unroll
        for (int j = 0; j < TN; ++j) bb[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Wl + (j * 32) * LDP + ks * 16));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) dacc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bb[j], dacc[i][j], 0, 0, 0);
      }
      __syncthreads();  // (the next chunk's patch / first weight stage overwrite what these MFMAs read)
    }
  }
  CSTAMP(120);

  // ---- epilogue: D[pixel][cout]: cout on the lanes, pixels (e&3) + 8(e>>2) + 4h in the registers.  A row's address is a
  // wave-uniform base (SGPRs) plus one 32-bit lane offset.  Neighbouring lanes hold neighbouring channels of the same
  // pixel: lane pairs swap one value (DPP) so that the even lane owns channels (n, n+1) of pixel e and the odd lane the
  // same channels of pixel e+1 -- one 4-byte (bf16) / 8-byte (fp32) store per lane and register PAIR instead of a 2-byte
  // store per register (the scalar bf16 stores alone were a third of the 64-channel layers' time).
  // (the loop above ended with a barrier: patch and weights are dead, the staging tiles reuse their LDS)
  epilogue_rows<TM, TN, TN == 3, STAGED>(acc, p, img, Ho, Wo, y0 + wm * TM, x0, n0, tiles_x, tx, lane, Ps + wm * stage_elems<TN>(), wst,
                                         wm * TM);
  CSTAMP(121);
  if (p.out_part) {
    __syncthreads();
    // (the thread index again, behind an opaque move: carried across the main loop, its byte offset was the one register the
    //  kernel spilled at the 168 of three workgroups per CU)
    int t2;
    asm volatile("v_mov_b32 %0, %1" : "=v"(t2) : "v"(threadIdx.x));
    write_tile_stats<G::ROWS, BN>(p, wst, img, ty * tiles_x + tx, n0, t2);
  }
  if constexpr (DS) {  // the branch's own output tensor and statistics, through the same epilogue
    __syncthreads();  // (write_tile_stats has read the table the second epilogue overwrites)
    RowsArgs p2 = p;
    p2.out = p.out2; p2.bias = p.bias2; p2.out_part = p.out_part2;
    epilogue_rows<TM, TN, TN == 3, STAGED>(dacc, p2, img, Ho, Wo, y0 + wm * TM, x0, n0, tiles_x, tx, lane, Ps + wm * stage_elems<TN>(), wst,
                                           wm * TM);
    if (p2.out_part) {
      __syncthreads();
      int t3;
      asm volatile("v_mov_b32 %0, %1" : "=v"(t3) : "v"(threadIdx.x));
      write_tile_stats<G::ROWS, BN>(p2, wst, img, ty * tiles_x + tx, n0, t3);
    }
  }
  CSTAMP(122);
}

// ------------------------------------------------------------------------------------------------
// The 7x7 / stride 2 / pad 3 stem (BasicEncoder.conv1, blocks.py:222) on the same 8 x 32 pixel tiles.  The input is the
// normalised RGB image with channels padded to four (fp32 [n][H][W][4]); the weights are [Cout][kh][8][4] bf16 (zero for
// kw = 7 and c = 3), i.e. K = 7 x 32.  A 32-wide k-group is one filter row; lane (r, h) of a 16-wide MFMA step reads two
// neighbouring input pixels (8 bf16 = 16 B), and neighbouring output pixels are two input pixels apart, so the A fragments
// of a wave are consecutive 16-B words of a patch row: conflict-free ds_read_b128 with no im2col buffer at all.  The
// generic im2col GEMM spent 680 us per 24 images on this layer (45 GFLOP); the output write is the only real cost.
constexpr int SPR = 2 * TR + 5, SPC = 2 * TC + 6;   // 21 x 70 patch pixels (one spare column keeps rows 16-B aligned)
constexpr int SNP = SPR * SPC;
constexpr int SKW = 7 * 32;                         // K
constexpr int SLW = SKW + 8;                        // LDS weight row stride (bf16): 464 B, conflict-free b128 reads

template <int TN, bool STAGED>
__global__ __launch_bounds__(256) void stem7x7_rows_bf16(RowsArgs p) {
  constexpr int TM = 2, BN = TN * 32;
  constexpr int NPL = (SNP + 255) / 256;
  constexpr int NWF = (BN * (SKW / 8) + 255) / 256;
  __shared__ __attribute__((aligned(16))) unsigned short Ps[SNP * 4];
  __shared__ __attribute__((aligned(16))) unsigned short Ws[BN * SLW];
  __shared__ float wst[TR * BN * 2];
  const int t = threadIdx.x;
  for (int i = t; i < TR * BN * 2; i += 256) wst[i] = 0.f;
  const int Ho = (p.H + 6 - 7) / 2 + 1, Wo = (p.W + 6 - 7) / 2 + 1;
  const int tiles_x = (Wo + TC - 1) / TC, tiles_y = (Ho + TR - 1) / TR;
  int b = blockIdx.x;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y;
  const long long img = b / tiles_y;
  const int y0 = ty * TR, x0 = tx * TC;
  const float* in = p.in + img * (long long)p.H * p.W * 4;
  const long long hw = (long long)p.H * p.W;
  long long plane0 = 0;  // first element of the image's R plane in the planar clip
  if (p.rgb) {
    const long long gi = p.rgb_img0 + img;
    const long long tt = gi / p.rgb_V, vv = gi - tt * p.rgb_V;
    plane0 = ((vv * p.rgb_T + tt) * 3) * hw;
  }

  f32x4 rp[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int f = t + 256 * i;
    const int py = f / SPC, px = f - py * SPC;
    const int gy = 2 * y0 - 3 + py, gx = 2 * x0 - 3 + px;
    const bool ok = f < SNP && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    const int cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);
    f32x4 v;
    if (p.rgb) {  // (workgroup-uniform) three planes, one value each; consecutive lanes read consecutive pixels of a row
      const long long o = plane0 + (long long)cy * p.W + cx;
      float c0, c1, c2;
      if (p.rgb_u8) {
        const unsigned char* sp = static_cast<const unsigned char*>(p.rgb) + o;
        c0 = (float)sp[0]; c1 = (float)sp[hw]; c2 = (float)sp[2 * hw];
      } else {
        const float* sp = static_cast<const float*>(p.rgb) + o;
        c0 = sp[0]; c1 = sp[hw]; c2 = sp[2 * hw];
      }
      v = (f32x4){2.0f * (c0 / 255.0f) - 1.0f, 2.0f * (c1 / 255.0f) - 1.0f, 2.0f * (c2 / 255.0f) - 1.0f, 0.0f};
    } else {
      v = *reinterpret_cast<const f32x4*>(in + ((long long)cy * p.W + cx) * 4);
    }
    rp[i] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  u32x4 rw[NWF];
#pragma unroll
  for (int i = 0; i < NWF; ++i) {
    const int u = t + 256 * i;
    const int n = u / (SKW / 8), part = u - n * (SKW / 8);
    rw[i] = *reinterpret_cast<const u32x4*>(p.w + (long long)min(n, p.Cout - 1) * p.ldw + part * 8);
    if (n >= p.Cout) rw[i] = (u32x4){0u, 0u, 0u, 0u};
  }
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int f = t + 256 * i;
    if (f < SNP) *reinterpret_cast<u32x2*>(&Ps[f * 4]) = __builtin_bit_cast(u32x2, __builtin_convertvector(rp[i], bf16x4));
  }
#pragma unroll
  for (int i = 0; i < NWF; ++i) {
    const int u = t + 256 * i;
    const int n = u / (SKW / 8), part = u - n * (SKW / 8);
    if (n < BN) *reinterpret_cast<u32x4*>(&Ws[n * SLW + part * 8]) = rw[i];
  }
  __syncthreads();

  const int lane = t & 63;
  const int wm = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, h = lane >> 5;
  const unsigned short* Pl = Ps + ((wm * TM * 2) * SPC + 2 * r + 2 * h) * 4;
  const unsigned short* Wl = Ws + r * SLW + h * 8;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
  for (int kh = 0; kh < 7; ++kh) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[TM], bb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Pl + ((2 * i + kh) * SPC + 4 * ks) * 4));
#pragma unroll
      for (int j = 0; j < TN; ++j) bb[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Wl + j * 32 * SLW + kh * 32 + ks * 16));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bb[j], acc[i][j], 0, 0, 0);
    }
  }

  __syncthreads();  // every wave is done with the patch / weights: Ws becomes the staging area
  static_assert(4 * stage_elems<TN>() <= BN * SLW, "staging tiles fit the weight buffer");
  epilogue_rows<TM, TN, true, STAGED>(acc, p, img, Ho, Wo, y0 + wm * TM, x0, 0, tiles_x, tx, lane, Ws + wm * stage_elems<TN>(), wst, wm * TM);
  if (p.out_part) {
    __syncthreads();
    write_tile_stats<TR, BN>(p, wst, img, ty * tiles_x + tx, 0, t);
  }
}


// ------------------------------------------------------------------------------------------------
// Round 4: the wide 3x3 / stride-1 convolution (BasicEncoder.conv2, 416 -> 256 channels at H/4: 45 % of the encoder's flops,
// spatracker/blocks.py:246-282) as ONE 512-thread workgroup per CU that owns an 8 x 32 pixel tile and ALL 256 output channels
// of a channel block.  In-kernel stamps of the row-tile kernel on this layer (gpurun_out/r4_stamp_c2.txt): of a 32-channel
// chunk's 13.5 k cycles only the three MFMA segments (7 k) compute -- the rest is the register-staged loader (patch 2-3 k) and two
// barriers per 24 MFMAs (3 k) -- and inside the segments every group of MFMAs waits for the LDS reads issued just ahead of it (no
// registers left to read further ahead at 168 VGPRs / three workgroups per CU).  Here:
//   * patch and weights travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write, no VALU),
//     double-buffered, requested one stage (= one filter row of a 32-channel chunk: 48 MFMAs per wave) ahead; ONE raw barrier per
//     stage, every wave waits for its own DMAs (vmcnt(0)) just before it -- the stage's MFMAs cover their latency;
//   * the LDS images are unpadded 64-B slots (a 1-KiB DMA instruction is 16 whole slots, 4 lanes = one slot = one 64-B global
//     segment: coalesced) with the 16-B pieces of a slot XOR-swizzled by the slot's patch column / weight row (>> 2) & 3 -- applied
//     on the SOURCE address of the DMA and on the read address (cdna_hip_programming.md rule 21): conflict-free ds_read_b128;
//   * wave tile 4 rows x 64 channels (8 accumulators, 0.75 KiB of LDS reads per MFMA), the next group's six fragments are read
//     while the current group's eight MFMAs issue (two fragment sets, 256-VGPR budget at two waves per SIMD);
//   * the input patch is fetched ONCE per pixel tile (the 64-channel tiles re-read it four times through L2).
// Same accumulation order per accumulator (chunk, kh, kw, k-step) and the same epilogue as conv_rows_bf16: bit-identical outputs
// and statistics (tests/test_gpu_ops.py::test_conv_big_tile_bit_identical).
constexpr int BG_BN = 256;                               // output channels per workgroup
constexpr int BG_PC = TC + 2;                            // 34 patch columns (10 patch rows)
constexpr int BG_PSLOTS = (TR + 2) * BG_PC;              // 340 pixel slots of 64 B
constexpr int BG_PINSTR = 24;                            // DMA wave-instructions per patch chunk: 3 per wave (21.25 carry pixels)
constexpr int BG_PBYTES = BG_PINSTR * 1024;              // 24 576 B per patch buffer
constexpr int BG_PROW = BG_PC * 64;                      // 2 176 B per patch row
constexpr int BG_WBYTES = 3 * BG_BN * 64;                // 49 152 B per weight stage (3 taps x 256 rows x 32 channels)
constexpr int BG_WST = TR * BG_BN * 2 * 4;               // 16 384 B: per (tile row, channel) output statistics (reuse the weight buffers)
constexpr int BG_LDS = 2 * BG_PBYTES + 2 * BG_WBYTES;    // 147 456 B of the 163 840 a workgroup may own
static_assert(BG_PSLOTS * 64 <= BG_PBYTES, "patch slots fit the buffer");
static_assert(BG_WBYTES / 1024 == 6 * 8, "six weight DMA wave-instructions per wave and stage");
static_assert(BG_LDS <= 160 * 1024 && BG_WST <= 2 * BG_WBYTES, "one workgroup per CU");
static_assert(8 * stage_elems<2>() * 2 <= 2 * BG_PBYTES, "the epilogue's staging tiles reuse the patch buffers");

__device__ __attribute__((aligned(64))) unsigned mvt_conv_zero_page[16];  // source of the DMA lanes that fall outside the image

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;

__device__ __forceinline__ void dma16(const void* src, unsigned char* lds_dst_uniform) {
  // 64 lanes x 16 B: lane l's bytes land at lds_dst_uniform + 16 l (the destination is a wave-uniform base in M0, the source per
  // lane).  Inline asm, M0 saved / restored around it (cdna_hip_programming.md section 5.7): issued through
  // __builtin_amdgcn_global_load_lds the waitcnt pass treats every later ds_read as unordered against it and waits lgkmcnt(0)
  // where a counted wait would do (the fragment reads of the NEXT group were drained before every group's MFMAs).  The pass does
  // not see these requests at all: the kernel waits for them itself (s_waitcnt vmcnt(0) ahead of each stage's barrier).
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_ptr)lds_dst_uniform);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src), "s"(dst)
               : "memory");
}

__global__ __launch_bounds__(512, 2) void conv3x3_big_bf16(RowsArgs p) {
  constexpr int TM = 4, TN = 2;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[BG_LDS];  // (ONE LDS object: patch x2 | weights x2)
  unsigned char* const Pb = lds;
  unsigned char* const Wb = lds + 2 * BG_PBYTES;
  float* const wst = reinterpret_cast<float*>(Wb);  // after the loop: the statistics table reuses the weight buffers

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave & 1, wn = wave >> 1;   // rows 4 wm .. 4 wm + 3 of the tile, channels 64 wn .. 64 wn + 63 of the block
  const int r = lane & 31, h = lane >> 5;
  const int Ho = p.H, Wo = p.W;
  const int tiles_x = (Wo + TC - 1) / TC, tiles_y = (Ho + TR - 1) / TR;
  const int tiles_n = p.Cout / BG_BN;
  int b;
  {  // XCD-aware tile order, as in conv_rows_bf16: every XCD owns a contiguous run of tiles (neighbours share halo rows in its L2)
    const unsigned nwg = gridDim.x, q = nwg / 8, rr = nwg % 8, xcd = blockIdx.x % 8;
    b = (int)((xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + blockIdx.x / 8);
  }
  const int tn = b % tiles_n; b /= tiles_n;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y;
  const long long img = b / tiles_y;
  const int y0 = ty * TR, x0 = tx * TC, n0 = tn * BG_BN;
  const unsigned char* const in_img = reinterpret_cast<const unsigned char*>(p.in) + img * (long long)p.H * p.W * p.Cin * 2;
  const unsigned char* const wbase = reinterpret_cast<const unsigned char*>(p.w);
  const unsigned char* const zero = reinterpret_cast<const unsigned char*>(mvt_conv_zero_page);

  // ---- DMA lane state.  Patch: instruction k = wave + 8 i (i < 3) fills LDS pieces 64 k .. 64 k + 63 of a patch buffer; piece P
  // = slot P >> 2 (patch row py, column px), stored piece P & 3 holds channel piece (P & 3) ^ ((px >> 2) & 3) of the chunk.
  // (every wave issues all three instructions -- the stage loop stays free of branches, which cost the waitcnt pass its exact
  //  LDS counts; pieces behind the last slot and pixels outside the image read the zero page)
  int poff[3];  // byte offset inside the image (without the chunk offset), or -1: outside the image / behind the last slot
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int P = (wave + 8 * i) * 64 + lane;
    const int slot = P >> 2, py = slot / BG_PC, px = slot - py * BG_PC;
    const int q = (P & 3) ^ ((px >> 2) & 3);
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    const bool ok = slot < BG_PSLOTS && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    poff[i] = ok ? ((gy * p.W + gx) * p.Cin + q * 8) * 2 : -1;
  }
  // Weights: instruction k = wave + 8 i (i < 6) fills rows 16 k .. 16 k + 15 of a stage buffer; row R = tap kw = R >> 8 of the
  // stage's filter row, output channel n0 + (R & 255); stored piece (lane & 3) holds channel piece (lane & 3) ^ ((R >> 2) & 3)
  int woff[6];  // byte offset inside the weight matrix (without the stage's (kh, chunk) offset)
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int R = (wave + 8 * i) * 16 + (lane >> 2);
    const int q = (lane & 3) ^ ((R >> 2) & 3);
    woff[i] = ((n0 + (R & 255)) * p.ldw + (R >> 8) * p.Cin + q * 8) * 2;
  }
  // (one instruction at a time: inside the stage loop the requests are spread over the MFMA groups; `dst_stage` / `dst_c` pick the
  //  buffer, so that behind the last stage / chunk a harmless re-read of the current one goes to the idle buffer: no branches)
  auto dma_weight = [&](int stage, int dst_stage, int i) {  // stage = 3 chunk + kh; i < 6
    const int c = stage / 3, kh = stage - 3 * c;
    dma16(wbase + ((long long)kh * 3 * p.Cin + c * CK) * 2 + woff[i], Wb + (dst_stage & 1) * BG_WBYTES + wave * 1024 + i * 8192);
  };
  auto dma_patch1 = [&](int c, int dst_c, int i) {  // i < 3
    dma16(poff[i] >= 0 ? in_img + c * CK * 2 + poff[i] : zero, Pb + (dst_c & 1) * BG_PBYTES + wave * 1024 + i * 8192);
  };

  const int nchunk = p.Cin / CK;
#pragma unroll
  for (int i = 0; i < 3; ++i) dma_patch1(0, 0, i);
#pragma unroll
  for (int i = 0; i < 6; ++i) dma_weight(0, 0, i);

  // ---- fragment read bases.  A: lane (r, h) of tap kw, k-step ks reads slot (row, kw + r), piece (2 ks + h) ^ (((kw + r) >> 2) & 3);
  // B: row kw * 256 + 64 wn + 32 j + r, piece (2 ks + h) ^ ((r >> 2) & 3).  Rows / taps / channel blocks are immediate offsets.
  int pa[3][2], wl[2];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      pa[kw][ks] = (TM * wm * BG_PC + kw + r) * 64 + (((2 * ks + h) ^ (((kw + r) >> 2) & 3)) << 4);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) wl[ks] = (64 * wn + r) * 64 + (((2 * ks + h) ^ ((r >> 2) & 3)) << 4);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // Software pipeline over the stages (stage = chunk c, filter row kh: six groups g = (kw, k-step) of eight MFMAs per wave).  The
  // barrier of a stage sits between its groups 4 and 5: by then every fragment of the stage is in registers (group 5's reads were
  // issued ahead of group 4's MFMAs), so the stage's buffers are free, and the wave's DMAs of the next stage have landed (counted
  // vmcnt); BEHIND the barrier the first fragments of the next stage are requested and group 5's MFMAs cover their LDS round trip.
  // The next stage's weights are requested behind groups 0-2, the next chunk's patch (kh = 0 only) behind groups 3-4 -- it is first
  // read three stages later, so the wait of its own stage leaves it in flight (vmcnt(3): requests retire in order).
  bf16x8 a[2][TM], bb[2][TN];
  auto read_group = [&](int set, int g, int kh, const unsigned char* Pc, const unsigned char* Wc) {
    const int kw = g >> 1, ks = g & 1;
#pragma unroll
    for (int i = 0; i < TM; ++i)
      a[set][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Pc + pa[kw][ks] + (i + kh) * BG_PROW));
#pragma unroll
    for (int j = 0; j < TN; ++j)
      bb[set][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Wc + wl[ks] + kw * (BG_BN * 64) + j * 2048));
  };
  auto mfma_group = [&](int set) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[set][i], bb[set][j], acc[i][j], 0, 0, 0);
  };
  CSTAMP(0);
#ifdef MVT_BIG_PRIO
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);  // the second-dispatched half loses every arbitration against its older SIMD partner
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_group(0, 0, 0, Pb, Wb);
  for (int c = 0; c < nchunk; ++c) {
    const unsigned char* const Pc = Pb + (c & 1) * BG_PBYTES;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int stage = 3 * c + kh;
      const int stage_n = stage + 1 < 3 * nchunk ? stage + 1 : stage, chunk_n = c + 1 < nchunk ? c + 1 : c;  // sources of the requests
      const unsigned char* const Wc = Wb + (stage & 1) * BG_WBYTES;
      // (group g's fragments sit in set g & 1: six groups per stage keep the parity across stages)
#pragma unroll
      for (int g = 0; g < 5; ++g) {
        read_group((g + 1) & 1, g + 1, kh, Pc, Wc);
#ifdef MVT_BIG_EARLY
        if (g == 0) {
#pragma unroll
          for (int i = 0; i < 6; ++i) dma_weight(stage_n, stage + 1, i);
        } else if (g == 1 && kh == 0) {
#pragma unroll
          for (int i = 0; i < 3; ++i) dma_patch1(chunk_n, c + 1, i);
        }
#else
        if (g < 3) {
          dma_weight(stage_n, stage + 1, 2 * g);
          dma_weight(stage_n, stage + 1, 2 * g + 1);
        } else if (kh == 0) {
          dma_patch1(chunk_n, c + 1, g == 3 ? 0 : 2);
          if (g == 3) dma_patch1(chunk_n, c + 1, 1);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(g & 1);
        __builtin_amdgcn_sched_barrier(0);
        if (stage < 12) CSTAMP(1 + 10 * stage + g);
      }
      if (kh == 0) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if (stage < 12) CSTAMP(1 + 10 * stage + 5);
      __builtin_amdgcn_s_barrier();
      if (stage < 12) CSTAMP(1 + 10 * stage + 6);
      read_group(0, 0, kh == 2 ? 0 : kh + 1, kh == 2 ? Pb + ((c + 1) & 1) * BG_PBYTES : Pc, Wb + ((stage + 1) & 1) * BG_WBYTES);
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(1);
      __builtin_amdgcn_sched_barrier(0);
      if (stage < 12) CSTAMP(1 + 10 * stage + 7);
    }
  }
  CSTAMP(124);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // every wave is done with the patch / weight buffers: the staging tiles reuse the patch buffers
  // this wave's part of the statistics table (its rows x its channels; rows past the image stay zero) -- written below by the same wave
  for (int i = lane; i < TM * 64 * 2; i += 64) wst[((wm * TM + (i >> 7)) * BG_BN + wn * 64) * 2 + (i & 127)] = 0.f;

  epilogue_rows<TM, TN, false, true, BG_BN>(acc, p, img, Ho, Wo, y0 + wm * TM, x0, n0 + wn * 64, tiles_x, tx, lane,
                                             reinterpret_cast<unsigned short*>(Pb) + wave * stage_elems<TN>(), wst + wn * 64 * 2, wm * TM);
  CSTAMP(125);
  if (p.out_part) {
    __syncthreads();
    write_tile_stats<TR, BG_BN>(p, wst, img, ty * tiles_x + tx, n0, t);
  }
  CSTAMP(126);
}


}  // namespace

#ifdef MVT_STAMPS
extern "C" int mvt_debug_read_conv_stamps(unsigned long long* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(mvt_conv_stamp_buf), sizeof(unsigned long long) * 2 * 8 * 128) == hipSuccess ? MVT_OK : MVT_ERR_HIP_BASE;
}
extern "C" int mvt_debug_clear_conv_stamps() {
  static unsigned long long z[2 * 8 * 128];
  return hipMemcpyToSymbol(HIP_SYMBOL(mvt_conv_stamp_buf), z, sizeof(z)) == hipSuccess ? MVT_OK : MVT_ERR_HIP_BASE;
}
#endif

// tile rows of the variant the launcher picks for (kernel size, stride) -- shared with mvt_conv2d_stat_slots
static int rows_nw8() {
  static const int v = getenv("MVT_ROWS_NW8") ? atoi(getenv("MVT_ROWS_NW8")) : 0;  // tuning override
  return v;
}
__attribute__((visibility("hidden"))) int mvt_detail_conv_rows_tile_rows(int ksize, int stride, int Ho) {
  if (ksize == 3 && stride == 1) return (rows_nw8() == 1 && Ho % 16 == 0) ? 16 : (rows_nw8() == 2 ? 4 : 8);
  if (ksize == 3) return 4;
  if (ksize == 7) return TR;  // stem
  return 8;
}
// slots of the fused statistics: one per workgroup tile (tile rows x 32 output pixels)
__attribute__((visibility("hidden"))) int mvt_detail_conv_rows_slots(int Ho, int Wo, int tile_rows) {
  return (int)(mvt_cdiv(Ho, tile_rows) * mvt_cdiv(Wo, TC));
}

// 3x3 (pad 1) or 1x1 (pad 0) convolution, stride 1 or 2, Cin % 32 == 0, bf16 mode
__attribute__((visibility("hidden"))) int mvt_detail_conv_rows(const void* in, const unsigned short* w, int ldw, const float* bias, void* out,
                                                               int n, int H, int W, int Cin, int Cout, int ksize, int stride, int ldo,
                                                               int io_flags, const float* in_stats, float* out_partial,
                                                               hipStream_t stream) {
  MVT_REQUIRE((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2));
  MVT_REQUIRE((!in_stats || Cin <= MAXC) && Cin % CK == 0 && (long long)H * W * Cin < (1LL << 31) && (long long)Cout * ldw < (1LL << 31));
  MVT_REQUIRE((long long)(TC + 8) * ldo * 4 < (1LL << 31));
  const int pad = ksize == 3 ? 1 : 0;
  const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
  RowsArgs a{};
  a.in = (const float*)in; a.w = w; a.bias = bias; a.out = (float*)out; a.in_stats = in_stats; a.out_part = out_partial;
  a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ldw = ldw; a.ldo = ldo;
  a.slots = mvt_detail_conv_rows_slots(Ho, Wo, mvt_detail_conv_rows_tile_rows(ksize, stride, Ho));
  a.in_bf16 = io_flags & MVT_IO_IN_BF16 ? 1 : 0;
  a.out_bf16 = io_flags & MVT_IO_OUT_BF16 ? 1 : 0;
  // (no dynamic LDS anywhere in this library: every kernel's group segment is its static size, checked at compile time against
  //  the 160 KiB a workgroup may own -- see DESIGN.md section 5, "a queue abort worth remembering")
  const bool n96 = Cout % 64 != 0 && Cout % 96 == 0;
  const int bn = n96 ? 96 : 64;
  // (64-channel tiles keep three workgroups per CU with bf16 tensors: measured 1.4x faster than 128-channel tiles at one per CU)
  // bf16 output goes through the LDS-staged epilogue when the tensor allows 16-byte pieces (and the staging tiles fit the patch)
  const bool st_ok = a.out_bf16 && Cout % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)out & 15) == 0;
#define LAUNCH2(TM_, TN_, KS_, S_, NW_, INB_)                                                                                       \
  do {                                                                                                                             \
    constexpr bool fits = NW_ * stage_elems<TN_>() <= Geo<TM_, KS_, S_, NW_>::NSLOT * LDP;                                          \
    if (fits && st_ok) hipLaunchKernelGGL((conv_rows_bf16<TM_, TN_, KS_, S_, INB_, NW_, fits>), grid, dim3(64 * NW_), 0, stream, a); \
    else hipLaunchKernelGGL((conv_rows_bf16<TM_, TN_, KS_, S_, INB_, NW_, false>), grid, dim3(64 * NW_), 0, stream, a);            \
  } while (0)
#define LAUNCH(TM_, KS_, S_, NW_)                                                                                                   \
  do {                                                                                                                             \
    const long long tiles = (long long)n * mvt_cdiv(Ho, NW_ * TM_) * mvt_cdiv(Wo, TC) * mvt_cdiv(Cout, bn);                        \
    MVT_REQUIRE(tiles < (1LL << 31));                                                                                              \
    const dim3 grid((unsigned)tiles);                                                                                              \
    if (n96) {                                                                                                                     \
      if (a.in_bf16) LAUNCH2(TM_, 3, KS_, S_, NW_, true);                                                                          \
      else LAUNCH2(TM_, 3, KS_, S_, NW_, false);                                                                                   \
    } else {                                                                                                                       \
      if (a.in_bf16) LAUNCH2(TM_, 2, KS_, S_, NW_, true);                                                                          \
      else LAUNCH2(TM_, 2, KS_, S_, NW_, false);                                                                                   \
    }                                                                                                                              \
  } while (0)
  const int nw8 = rows_nw8();
  // wide 3x3 / stride-1 layers without normalise-on-load (conv2): one 512-thread workgroup per 8 x 32 pixel tile and 256-channel
  // block, LDS-DMA staging (conv3x3_big_bf16).  MVT_CONV_BIG=0 keeps the 64-channel row tiles (read per call: A/B runs, tests)
  if (ksize == 3 && stride == 1 && !in_stats && !(io_flags & MVT_IO_SHORT_WG) && a.in_bf16 && st_ok && Cout % BG_BN == 0 && ((uintptr_t)in & 15) == 0 && Cin % 8 == 0 &&
      ldw % 8 == 0 && ((uintptr_t)w & 15) == 0 && (long long)Cout * ldw * 2 < (1LL << 31) && (long long)H * W * Cin * 2 < (1LL << 31)) {
    const char* e = getenv("MVT_CONV_BIG");
    if (!e || atoi(e) != 0) {
      const long long tiles = (long long)n * mvt_cdiv(Ho, TR) * mvt_cdiv(Wo, TC) * (Cout / BG_BN);
      MVT_REQUIRE(tiles < (1LL << 31));
      hipLaunchKernelGGL(conv3x3_big_bf16, dim3((unsigned)tiles), dim3(512), 0, stream, a);
      return mvt_launch_status();
    }
  }
  if (ksize == 3 && stride == 1 && nw8 == 1 && Ho % 16 == 0) LAUNCH(2, 3, 1, 8);
  else if (ksize == 3 && stride == 1 && nw8 == 2) LAUNCH(1, 3, 1, 4);
  else if (ksize == 3 && stride == 1) LAUNCH(2, 3, 1, 4);
  else if (ksize == 3) LAUNCH(1, 3, 2, 4);   // the stride-2 patch is four times larger per output pixel: 4 x 32 pixel tiles
  else if (stride == 1) LAUNCH(2, 1, 1, 4);
  else LAUNCH(2, 1, 2, 4);
#undef LAUNCH2
#undef LAUNCH
  return mvt_launch_status();
}

// The stride-2 3x3 convolution of a ResidualBlock together with the block's 1x1 / stride-2 downsample branch of the same input
// (blocks.py:112-128) in ONE launch: the downsample is the centre tap of the 3x3 window with its own weights, so the input patch is
// staged once, one launch, one InstanceNorm-finish and one full read of the block input fewer per block.
__attribute__((visibility("hidden"))) int mvt_detail_conv3x3s2_down(const void* in, const unsigned short* w3, int ldw3, const float* b3,
                                                                    const unsigned short* wd, int ldwd, const float* bd, void* out3,
                                                                    void* outd, int n, int H, int W, int Cin, int Cout, int ldo,
                                                                    float* part3, float* partd, hipStream_t stream) {
  MVT_REQUIRE(in && w3 && wd && out3 && outd && n > 0 && Cin % CK == 0 && Cout % 32 == 0 && Cout % 8 == 0 && ldo % 8 == 0 && ldo >= Cout);
  MVT_REQUIRE((long long)H * W * Cin < (1LL << 31) && (long long)Cout * ldw3 < (1LL << 31) && ldwd >= Cin && ldw3 >= 9 * Cin);
  MVT_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)out3 & 15) == 0 && ((uintptr_t)outd & 15) == 0 && ((uintptr_t)w3 & 15) == 0 &&
              ((uintptr_t)wd & 15) == 0 && ldw3 % 8 == 0 && ldwd % 8 == 0 && (!part3) == (!partd));
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  RowsArgs a{};
  a.in = (const float*)in; a.w = w3; a.bias = b3; a.out = (float*)out3; a.in_stats = nullptr; a.out_part = part3;
  a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ldw = ldw3; a.ldo = ldo;
  a.slots = mvt_detail_conv_rows_slots(Ho, Wo, 4);
  a.in_bf16 = 1; a.out_bf16 = 1;
  a.w2 = wd; a.bias2 = bd; a.out2 = (float*)outd; a.out_part2 = partd; a.ldw2 = ldwd;
  const bool n96 = Cout % 64 != 0 && Cout % 96 == 0;
  const long long tiles = (long long)n * mvt_cdiv(Ho, 4) * mvt_cdiv(Wo, TC) * mvt_cdiv(Cout, n96 ? 96 : 64);
  MVT_REQUIRE(tiles < (1LL << 31));
  static_assert(4 * stage_elems<2>() <= Geo<1, 3, 2, 4>::NSLOT * LDP && 4 * stage_elems<3>() <= Geo<1, 3, 2, 4>::NSLOT * LDP, "staged epilogue");
  if (n96) hipLaunchKernelGGL((conv_rows_bf16<1, 3, 3, 2, true, 4, true, true>), dim3((unsigned)tiles), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((conv_rows_bf16<1, 2, 3, 2, true, 4, true, true>), dim3((unsigned)tiles), dim3(256), 0, stream, a);
  return mvt_launch_status();
}

// 7x7 stride-2 pad-3 stem with Cin padded to 4 and Cout <= 64 (bf16 mode); the input is either the normalised [n][H][W][4] fp32
// tensor `in`, or (in == null) the planar clip `rgb` (V,T,3,H,W), fp32 or uint8, images rgb_img0 .. rgb_img0 + n - 1 frame-major
__attribute__((visibility("hidden"))) int mvt_detail_stem7x7_rows_src(const float* in, const void* rgb, int rgb_u8, int rgb_V, int rgb_T,
                                                                      long long rgb_img0, const unsigned short* w, int ldw,
                                                                      const float* bias, void* out, int n, int H, int W, int Cout, int ldo,
                                                                      int io_flags, float* out_partial, hipStream_t stream);
__attribute__((visibility("hidden"))) int mvt_detail_stem7x7_rows(const float* in, const unsigned short* w, int ldw, const float* bias,
                                                                  void* out, int n, int H, int W, int Cout, int ldo, int io_flags,
                                                                  float* out_partial, hipStream_t stream) {
  return mvt_detail_stem7x7_rows_src(in, nullptr, 0, 0, 0, 0, w, ldw, bias, out, n, H, W, Cout, ldo, io_flags, out_partial, stream);
}
__attribute__((visibility("hidden"))) int mvt_detail_stem7x7_rows_src(const float* in, const void* rgb, int rgb_u8, int rgb_V, int rgb_T,
                                                                      long long rgb_img0, const unsigned short* w, int ldw,
                                                                      const float* bias, void* out, int n, int H, int W, int Cout, int ldo,
                                                                      int io_flags, float* out_partial, hipStream_t stream) {
  MVT_REQUIRE(Cout > 0 && Cout <= 64 && ldw >= SKW && (long long)H * W * 4 < (1LL << 31) && !(io_flags & MVT_IO_IN_BF16));
  MVT_REQUIRE((in != nullptr) != (rgb != nullptr));
  MVT_REQUIRE(!rgb || (rgb_V > 0 && rgb_T > 0 && rgb_img0 >= 0 && rgb_img0 + n <= (long long)rgb_V * rgb_T && (rgb_u8 == 0 || rgb_u8 == 1)));
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  RowsArgs a{};
  a.rgb = rgb; a.rgb_u8 = rgb_u8; a.rgb_V = rgb_V; a.rgb_T = rgb_T; a.rgb_img0 = rgb_img0;
  a.in = in; a.w = w; a.bias = bias; a.out = (float*)out; a.in_stats = nullptr; a.out_part = out_partial;
  a.H = H; a.W = W; a.Cin = 4; a.Cout = Cout; a.ldw = ldw; a.ldo = ldo;
  a.slots = mvt_detail_conv_rows_slots(Ho, Wo, TR);
  a.in_bf16 = 0;
  a.out_bf16 = io_flags & MVT_IO_OUT_BF16 ? 1 : 0;
  const long long tiles = (long long)n * mvt_cdiv(Ho, TR) * mvt_cdiv(Wo, TC);
  MVT_REQUIRE(tiles < (1LL << 31) && (long long)(TC + 8) * ldo * 4 < (1LL << 31));
  const bool st_ok = a.out_bf16 && Cout % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)out & 15) == 0;
  if (Cout <= 32) {
    if (st_ok) hipLaunchKernelGGL((stem7x7_rows_bf16<1, true>), dim3((unsigned)tiles), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((stem7x7_rows_bf16<1, false>), dim3((unsigned)tiles), dim3(256), 0, stream, a);
  } else {
    if (st_ok) hipLaunchKernelGGL((stem7x7_rows_bf16<2, true>), dim3((unsigned)tiles), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((stem7x7_rows_bf16<2, false>), dim3((unsigned)tiles), dim3(256), 0, stream, a);
  }
  return mvt_launch_status();
}
