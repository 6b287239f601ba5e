// Shared helpers for the gfx950 kernels of libmvtracker_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mvtracker_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MVT_WAVE 64

#define MVT_REQUIRE(cond) \
  do {                    \
    if (!(cond)) return MVT_ERR_ARG; \
  } while (0)

static inline int mvt_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MVT_OK : MVT_ERR_HIP_BASE + (int)e;
}

static inline hipStream_t mvt_stream(void* s) { return (hipStream_t)s; }

static inline long long mvt_cdiv(long long a, long long b) { return (a + b - 1) / b; }

__device__ __forceinline__ float mvt_gelu_tanh(float x) {
  // 0.5 x (1 + tanh(k)) = x * sigmoid(2k), k = sqrt(2/pi) (x + 0.044715 x^3).  ocml tanhf costs ~150 instructions
  // and dominated the fc1 epilogue; v_exp_f32 + v_rcp_f32 are 1-ulp hardware ops (relative error ~2e-7).
  const float k2 = 2.0f * 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return x * __builtin_amdgcn_rcpf(1.0f + __expf(-k2));
}

__device__ __forceinline__ float mvt_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }

__device__ __forceinline__ float mvt_act(float x, int act) {
  switch (act) {
    case MVT_ACT_RELU: return fmaxf(x, 0.0f);
    case MVT_ACT_GELU_TANH: return mvt_gelu_tanh(x);
    case MVT_ACT_GELU_ERF: return mvt_gelu_erf(x);
    default: return x;
  }
}

// Activation tensors of the encoder are fp32 or, in bf16 mode, bf16 (MVT_IO_* flags).  Element offsets, fp32 values.
// Plain cast (v_cvt_pk_bf16_f32, round to nearest even): a NaN stays a NaN.  Rounding by integer arithmetic on the f32 bits
// turns some NaNs into 0 / inf (MI355X_MICROARCH.md, bf16 conversion pitfall), which would launder a NaN born in the bf16
// encoder / updater before the deferred NaN guard (delta_split) sees it.
__device__ __forceinline__ unsigned short mvt_bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
__device__ __forceinline__ f32x4 load_act4(const float* base, long long off, int is_bf16) {
  if (!is_bf16) return *reinterpret_cast<const f32x4*>(base + off);
  const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + off);
  return (f32x4){__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xFFFF0000u), __uint_as_float(w.y << 16),
                 __uint_as_float(w.y & 0xFFFF0000u)};
}
__device__ __forceinline__ void store_act(float* base, long long off, float v, int is_bf16) {
  if (!is_bf16) base[off] = v;
  else reinterpret_cast<unsigned short*>(base)[off] = mvt_bf16_bits(v);
}
__device__ __forceinline__ void store_act4(float* base, long long off, const f32x4& v, int is_bf16) {
  if (!is_bf16) {
    *reinterpret_cast<f32x4*>(base + off) = v;
  } else {
    uint2 w;
    w.x = (unsigned)mvt_bf16_bits(v[0]) | ((unsigned)mvt_bf16_bits(v[1]) << 16);
    w.y = (unsigned)mvt_bf16_bits(v[2]) | ((unsigned)mvt_bf16_bits(v[3]) << 16);
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + off) = w;
  }
}

// Key-split attention partials in global memory (attention_mfma.hip writes, its merge variants and block_fused.hip's ATT 3 read):
// one record per (split, chunk) = 17 quads x 64 lanes x 4 floats, QUAD-major, so that every access of a wave is one contiguous
// 1-KiB piece (lane-major 68-float records made every load and store of a wave touch 64 different cache lines).  Quad 0 =
// (m[0], m[1], l[0], l[1]); quad 1 + (mb*2 + db)*4 + g = accumulator registers 4g..4g+3 of O^T block (mb, db).
#define MVT_PART_FLOATS (17 * 64 * 4)
__device__ __forceinline__ long long mvt_part_off(long long rec, int quad, int lane) { return (rec * 17 + quad) * 256 + lane * 4; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int o) {
  unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
  lo = __shfl_xor(lo, o, 64);
  hi = __shfl_xor(hi, o, 64);
  return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long w = shfl_xor_u64(v, o);
    v = w < v ? w : v;
  }
  return v;
}
