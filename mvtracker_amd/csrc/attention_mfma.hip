// Flash-style attention on the bf16 matrix cores for the updater's space attention (virtual<-point: 64 queries x N
// keys; point<-virtual: N queries x 64 keys; virtual self: 64 x 64; cotracker2/blocks.py:258-271), head width 48.
//
// A wave owns 64 queries (two 32-wide MFMA blocks) of one (group, head) and walks the keys 32 at a time:
//     S^T (key x query)  = K_blk (32 x 48) . Q^T        A = K rows straight from global (fp32 -> bf16), B = Q rows
//     online softmax over the key index = over the 16 accumulator registers (+ the partner half-wave, one shuffle)
//     O^T (d x query)   += V^T (d x key) . P^T          B = P^T taken from the accumulator (registers 8s..8s+7 are the
//                                                       B fragment of k-step s), A = V^T from a small transposed LDS
//                                                       image read in the accumulator's key order
// so scores and probabilities never leave registers and the output lands with the query on the lane (row stores).
// KS > 1 splits the keys of one query chunk over KS waves; their (m, l, O^T) states are merged through LDS.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int DH = 48;
constexpr int LDV = 40;  // bf16 row stride of the V^T image: 32 keys + 8 pad (80 B)

// eight consecutive elements at element offset `off` of a fp32 or bf16 tensor, as bf16
__device__ __forceinline__ bf16x8 load8(const float* base, long long off, int bf) {
  if (bf) return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(base) + off));
  const float* p = base + off;
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  const bf16x4 x = __builtin_convertvector(a, bf16x4), y = __builtin_convertvector(b, bf16x4);
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    o[e] = x[e];
    o[4 + e] = y[e];
  }
  return o;
}

// (BFIN: bf16 q / k / v tensors -- a compile-time flag: as a run-time one every load sat in its own branch with its own full wait)
template <int KS, bool BFIN>
__global__ __launch_bounds__(KS == 1 ? 256 : KS * 64) void attention_mfma_kernel(
    const float* __restrict__ q, int ldq, long long q_gs, long long q_is, const float* __restrict__ k, const float* __restrict__ v,
    int ldkv, long long k_gs, long long k_is, float* __restrict__ o, int ldo, int groups, int nq, int nk, int heads, int bf_in_unused,
    int bf_out, float* __restrict__ ws) {
  constexpr int bf_in = BFIN ? 1 : 0;
  // ws != null: the keys are additionally cut over gridDim.y workgroups; wave 0 leaves its (m, l, O) partial in ws and
  // attention_merge_kernel finishes the softmax (the 64 virtual x 1024 point attention is only 72 (frame, head) chunks)
  constexpr int NW = KS == 1 ? 4 : KS;
  constexpr int VT = 64 * LDV;                              // bf16 elements of one wave's V^T image (64 d rows)
  constexpr int MERGE = KS > 1 ? (KS - 1) * 64 * 68 : 0;    // floats: per lane m[2], l[2], 64 accumulators
  __shared__ __attribute__((aligned(16))) unsigned short vts[NW * VT];
  __shared__ float red[MERGE > 0 ? MERGE : 1];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  unsigned short* vt = vts + wave * VT;

  const int chunks = (nq + 63) / 64;
  const long long nchunk = (long long)groups * heads * chunks;
  // Key-split-over-workgroups form (ws != null): placed XCD-AWARE.  The hardware deals consecutive linear workgroup ids round-robin
  // to the 8 XCDs; chunk = (frame, head), so with the plain mapping the six heads of a frame -- which share every 128-byte line of
  // the frame's K / V rows -- sit on six different XCDs and each pulls those rows through the fabric into its own L2.  Remapped, XCD c
  // owns the chunk-major run [c, c+1) * total/8: whole frames (the same frames as in the block kernels that produced K / V's
  // consumers' inputs and that read the partial records next, block_fused.hip).
  unsigned bxl = blockIdx.x, byl = blockIdx.y;
#ifndef MVT_ATTN_NO_XCD  // (A/B builds, tools/build_variant.sh)
  if (KS > 1 && ws) {
    const unsigned total = gridDim.x * gridDim.y;
    if (total % 8 == 0) {
      const unsigned L = blockIdx.x + gridDim.x * blockIdx.y;
      const unsigned V = (L % 8) * (total / 8) + L / 8;
      bxl = V / gridDim.y;
      byl = V % gridDim.y;
    }
  }
#endif
  const long long chunk_id = KS == 1 ? (long long)bxl * 4 + wave : (long long)bxl;
  const bool chunk_ok = chunk_id < nchunk;
  const long long cid = chunk_ok ? chunk_id : 0;
  const int qc = (int)(cid % chunks);
  const int hd = (int)((cid / chunks) % heads);
  const long long g = cid / ((long long)chunks * heads);

  const int nkb = (nk + 31) / 32;                       // key blocks
  const int gper = ws ? (nkb + (int)gridDim.y - 1) / (int)gridDim.y : nkb;   // ... of this workgroup
  const int g0 = ws ? (int)byl * gper : 0;
  const int g1 = g0 + gper < nkb ? g0 + gper : nkb;
  const int per = (gper + KS - 1) / KS;
  const int kb0 = KS == 1 ? g0 : g0 + wave * per;
  const int kb1 = KS == 1 ? g1 : (kb0 + per < g1 ? kb0 + per : g1);
  const long long kvbase = (g * k_gs) * ldkv + hd * DH;  // element offset (fp32 or bf16 tensors)
  const long long kstep = k_is * ldkv;
  // K / V operands of one 32-key block as they come from memory: the block after the current one is always in flight (and the
  // first one is requested ahead of the Q conversion), so a wave pays one memory round trip, not one per key block
  struct KV {
    bf16x8 kf[3];   // A fragments: key row r, k = d
    f32x4 vraw[6];  // V[key][24h + 4i .. +3]: fp32 values, or (bf16 tensors) the two packed words in [0], [1]
  };
  auto load_kv = [&](int kb, KV& t) {
    const int key = kb * 32 + r;
    const long long ko = kvbase + (long long)(key < nk ? key : nk - 1) * kstep;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) t.kf[ks] = load8(k, ko + ks * 16 + 8 * h, bf_in);
    const long long vo = ko + 24 * h;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (bf_in) {
        const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(v) + vo + 4 * i);
        t.vraw[i] = (f32x4){__uint_as_float(w.x), __uint_as_float(w.y), 0.f, 0.f};
      } else {
        t.vraw[i] = *reinterpret_cast<const f32x4*>(v + vo + 4 * i);
      }
    }
  };
  KV cur;
  if (kb0 < kb1) load_kv(kb0, cur);

  // Q^T fragments (B operand): lane (r, h) of block mb holds Q[query qc*64 + mb*32 + r][ks*16 + 8h ..], pre-scaled
  const float scale = 1.0f / sqrtf((float)DH);
  bf16x8 qf[2][3];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int qi = qc * 64 + mb * 32 + r;
    const long long qo = (g * q_gs + (long long)(qi < nq ? qi : nq - 1) * q_is) * ldq + hd * DH;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      f32x4 a = load_act4(q, qo + ks * 16 + 8 * h, bf_in), b = load_act4(q, qo + ks * 16 + 8 * h + 4, bf_in);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[e] *= scale;
        b[e] *= scale;
      }
      const bf16x4 x = __builtin_convertvector(a, bf16x4), y = __builtin_convertvector(b, bf16x4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        qf[mb][ks][e] = x[e];
        qf[mb][ks][4 + e] = y[e];
      }
    }
  }
  // rows 48..63 of the V^T image stay zero (d padding of the second 32-row block)
  for (int i = lane; i < 16 * LDV; i += 64) vt[48 * LDV + i] = 0;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");

  f32x16 oacc[2][2];  // [mb][db]: O^T rows d = db*32 + .., column = query
  float m[2], l[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    m[mb] = -INFINITY;
    l[mb] = 0.f;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[mb][db][e] = 0.f;
  }

#pragma unroll 1
  for (int kb = kb0; kb < kb1; ++kb) {
    KV nxt = cur;
    if (kb + 1 < kb1) load_kv(kb + 1, nxt);
    // ---- K block: A fragments (key row r of this block, k = d)
    const int key = kb * 32 + r;
    bf16x8 kf[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) kf[ks] = cur.kf[ks];
    // ---- V block -> transposed LDS image vt[d][key]: lane handles key (lane & 31), d range 24 * (lane >> 5) .. +23
    {
      // (the image is written as 16-bit elements and read back as 32-bit vectors: a wavefront-scope fence keeps the
      //  compiler's type-based alias analysis from reordering the two)
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        f32x4 a = cur.vraw[i];
        if (bf_in) {
          const unsigned wx = __float_as_uint(a[0]), wy = __float_as_uint(a[1]);
          a = (f32x4){__uint_as_float(wx << 16), __uint_as_float(wx & 0xFFFF0000u), __uint_as_float(wy << 16), __uint_as_float(wy & 0xFFFF0000u)};
        }
        if (key >= nk) a = (f32x4){0.f, 0.f, 0.f, 0.f};
        // (16-bit element extraction from the packed bf16 vector is done on the 32-bit words: hipcc 7.2 emits
        //  ds_write_b16 of the same low half for every element of `b[e]` otherwise)
        const u32x2 w = __builtin_bit_cast(u32x2, __builtin_convertvector(a, bf16x4));
#pragma unroll
        for (int e = 0; e < 4; ++e) vt[(24 * h + 4 * i + e) * LDV + r] = (unsigned short)((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xFFFFu));
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    // ---- S^T = K . Q^T, online softmax, O^T += V^T . P^T
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      f32x16 s;
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[mb][ks], s, 0, 0, 0);
      float mx = -INFINITY;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int kk = kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;  // key index of this register
        if (kk >= nk) s[e] = -INFINITY;
        mx = fmaxf(mx, s[e]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m[mb], mx);
      const float corr = (m[mb] == -INFINITY) ? 0.f : __expf(m[mb] - mn);
      float ps = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        s[e] = __expf(s[e] - mn);
        ps += s[e];
      }
      l[mb] = l[mb] * corr + ps;  // per-lane partial (this half-wave's key rows); halves are added at the end
      m[mb] = mn;
      if (__ballot(corr != 1.0f)) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int e = 0; e < 16; ++e) oacc[mb][db][e] *= corr;
      }
#pragma unroll
      for (int sk = 0; sk < 2; ++sk) {  // two k-steps of 16 keys: P^T fragment = registers 8sk .. 8sk+7
        f32x4 lo4, hi4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          lo4[e] = s[8 * sk + e];
          hi4[e] = s[8 * sk + 4 + e];
        }
        const bf16x4 bl = __builtin_convertvector(lo4, bf16x4), bh = __builtin_convertvector(hi4, bf16x4);
        bf16x8 pb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pb[e] = bl[e];
          pb[4 + e] = bh[e];
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          // V^T fragment in the accumulator's key order: keys 16sk + 4h + (0..3) and + 8, row d = db*32 + r
          const unsigned short* vr = &vt[(db * 32 + r) * LDV + 16 * sk + 4 * h];
          const u32x2 a0 = *reinterpret_cast<const u32x2*>(vr), a1 = *reinterpret_cast<const u32x2*>(vr + 8);
          const bf16x8 va = __builtin_bit_cast(bf16x8, (u32x4){a0[0], a0[1], a1[0], a1[1]});
          oacc[mb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb, oacc[mb][db], 0, 0, 0);
        }
      }
    }
    cur = nxt;
  }

  if (KS > 1) {
    if (wave > 0) {
      float* rr = red + ((wave - 1) * 64 + lane) * 68;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        rr[mb] = m[mb];
        rr[2 + mb] = l[mb];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int e = 0; e < 16; ++e) rr[4 + (mb * 2 + db) * 16 + e] = oacc[mb][db][e];
      }
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll 1
    for (int w = 0; w < KS - 1; ++w) {
      const float* rr = red + (w * 64 + lane) * 68;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const float mw = rr[mb];
        // both half-waves of a query share m (it is exchanged above), so the merge is lane-local
        const float mn = fmaxf(m[mb], mw);
        const float ca = (m[mb] == -INFINITY) ? 0.f : __expf(m[mb] - mn);
        const float cb = (mw == -INFINITY) ? 0.f : __expf(mw - mn);
        l[mb] = fmaf(l[mb], ca, rr[2 + mb] * cb);  // (explicit: every merge variant rounds identically)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int e = 0; e < 16; ++e) oacc[mb][db][e] = fmaf(oacc[mb][db][e], ca, rr[4 + (mb * 2 + db) * 16 + e] * cb);
        m[mb] = mn;
      }
    }
  }
  if (!chunk_ok) return;
  if (ws) {
    {
      const long long rec = byl * nchunk + chunk_id;
      *reinterpret_cast<f32x4*>(ws + mvt_part_off(rec, 0, lane)) = (f32x4){m[0], m[1], l[0], l[1]};
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq)
            *reinterpret_cast<f32x4*>(ws + mvt_part_off(rec, 1 + (mb * 2 + db) * 4 + gq, lane)) =
                (f32x4){oacc[mb][db][4 * gq], oacc[mb][db][4 * gq + 1], oacc[mb][db][4 * gq + 2], oacc[mb][db][4 * gq + 3]};
    }
    return;  // attention_merge_kernel, or the prologue of the consuming block kernel (MVT_ATTN_PARTIALS), finishes the softmax
  }
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const float lt = l[mb] + __shfl_xor(l[mb], 32, 64);
    const float inv = 1.0f / lt;
    const int qi = qc * 64 + mb * 32 + r;
    if (qi >= nq) continue;
    const long long oo = (g * q_gs + (long long)qi * q_is) * ldo + hd * DH;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int d = db * 32 + 8 * gq + 4 * h;
        if (d < DH) {
          f32x4 t;
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = oacc[mb][db][4 * gq + e] * inv;
          store_act4(o, oo + d, t, bf_out);
        }
      }
    }
  }
}

// Combine the per-workgroup partials of a key-split attention (fixed order: deterministic) and write the output.
__global__ __launch_bounds__(256) void attention_merge_kernel(const float* __restrict__ ws, int nsplit, long long nchunk, long long q_gs,
                                                              long long q_is, float* __restrict__ o, int ldo, int nq, int heads,
                                                              int bf_out) {
  const int lane = threadIdx.x & 63;
  const long long cid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (cid >= nchunk) return;
  const int r = lane & 31, h = lane >> 5;
  const int chunks = (nq + 63) / 64;
  const int qc = (int)(cid % chunks);
  const int hd = (int)((cid / chunks) % heads);
  const long long g = cid / ((long long)chunks * heads);
  float m[2], l[2];
  f32x16 oacc[2][2];
  auto ld = [&](long long rec, int quad) { return *reinterpret_cast<const f32x4*>(ws + mvt_part_off(rec, quad, lane)); };
  {
    const f32x4 q0 = ld(cid, 0);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      m[mb] = q0[mb];
      l[mb] = q0[2 + mb];
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const f32x4 t = ld(cid, 1 + (mb * 2 + db) * 4 + gq);
#pragma unroll
          for (int e = 0; e < 4; ++e) oacc[mb][db][4 * gq + e] = t[e];
        }
    }
  }
#pragma unroll 1
  for (int w = 1; w < nsplit; ++w) {
    const long long rec = w * nchunk + cid;
    const f32x4 q0 = ld(rec, 0);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const float mw = q0[mb];
      const float mn = fmaxf(m[mb], mw);
      const float ca = (m[mb] == -INFINITY) ? 0.f : __expf(m[mb] - mn);
      const float cb = (mw == -INFINITY) ? 0.f : __expf(mw - mn);
      l[mb] = fmaf(l[mb], ca, q0[2 + mb] * cb);  // (explicit: every merge variant rounds identically)
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const f32x4 t = ld(rec, 1 + (mb * 2 + db) * 4 + gq);
#pragma unroll
          for (int e = 0; e < 4; ++e) oacc[mb][db][4 * gq + e] = fmaf(oacc[mb][db][4 * gq + e], ca, t[e] * cb);
        }
      m[mb] = mn;
    }
  }
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const float lt = l[mb] + __shfl_xor(l[mb], 32, 64);
    const float inv = 1.0f / lt;
    const int qi = qc * 64 + mb * 32 + r;
    if (qi >= nq) continue;
    const long long oo = (g * q_gs + (long long)qi * q_is) * ldo + hd * DH;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int d = db * 32 + 8 * gq + 4 * h;
        if (d < DH) {
          f32x4 t;
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = oacc[mb][db][4 * gq + e] * inv;
          store_act4(o, oo + d, t, bf_out);
        }
      }
    }
  }
}

}  // namespace

extern "C" int mvt_attention_bf16(const void* q, int ldq, long long q_gs, long long q_is, const void* k, const void* v,
                                  int ldkv, long long k_gs, long long k_is, void* o, int ldo, int groups, int nq, int nk,
                                  int heads, int dh, int io_flags, float* workspace, void* stream) {
  MVT_REQUIRE((io_flags & ~(MVT_IO_IN_BF16 | MVT_IO_OUT_BF16 | MVT_ATTN_PARTIALS_ONLY)) == 0);
  const bool partials_only = (io_flags & MVT_ATTN_PARTIALS_ONLY) != 0;
  const int bf_in = io_flags & MVT_IO_IN_BF16 ? 1 : 0, bf_out = io_flags & MVT_IO_OUT_BF16 ? 1 : 0;
  MVT_REQUIRE(!bf_in || (ldq % 8 == 0 && ldkv % 8 == 0));  // 16-B aligned rows
  MVT_REQUIRE(!bf_out || ldo % 8 == 0);
  MVT_REQUIRE(q && k && v && o && groups > 0 && nq > 0 && nk > 0 && heads > 0 && dh == DH);
  MVT_REQUIRE(ldq % 4 == 0 && ldkv % 4 == 0 && ldo % 4 == 0 && ldq >= heads * dh && ldkv >= heads * dh && ldo >= heads * dh);
  MVT_REQUIRE(((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) && ((uintptr_t)o % 16 == 0));
  const long long nchunk = (long long)groups * heads * ((nq + 63) / 64);
#define LAUNCH_T(KS, BF, BLOCKS, THREADS, WSP)                                                                                        \
  hipLaunchKernelGGL((attention_mfma_kernel<KS, BF>), BLOCKS, dim3(THREADS), 0, mvt_stream(stream), (const float*)q, ldq, q_gs, q_is, \
                     (const float*)k, (const float*)v, ldkv, k_gs, k_is, (float*)o, ldo, groups, nq, nk, heads, bf_in, bf_out, WSP)
#define LAUNCH(KS, BLOCKS, THREADS)                                         \
  do {                                                                      \
    if (bf_in) LAUNCH_T(KS, true, dim3((unsigned)(BLOCKS)), THREADS, WS);  \
    else LAUNCH_T(KS, false, dim3((unsigned)(BLOCKS)), THREADS, WS);       \
  } while (0)
  constexpr int NSPLIT = MVT_ATTN_NSPLIT;
  MVT_REQUIRE(!partials_only || (nk >= 512 && nchunk < 256 && workspace && ((nk + 31) / 32) % NSPLIT == 0));
  if (nk >= 512 && nchunk < 256 && workspace && ((nk + 31) / 32) % NSPLIT == 0) {
    // too few (group, head) chunks to fill the chip: cut the keys over NSPLIT workgroups per chunk as well
    MVT_REQUIRE((uintptr_t)workspace % 16 == 0);
#define WS workspace
    if (bf_in) LAUNCH_T(4, true, dim3((unsigned)nchunk, NSPLIT), 256, workspace);
    else LAUNCH_T(4, false, dim3((unsigned)nchunk, NSPLIT), 256, workspace);
#undef WS
    if (!partials_only)
      hipLaunchKernelGGL(attention_merge_kernel, dim3((unsigned)mvt_cdiv(nchunk, 4)), dim3(256), 0, mvt_stream(stream), workspace, NSPLIT,
                         nchunk, q_gs, q_is, (float*)o, ldo, nq, heads, bf_out);
    return mvt_launch_status();
  }
#define WS nullptr
  if (nk >= 512 && nchunk < 1024) {
    LAUNCH(4, nchunk, 256);
  } else {
    LAUNCH(1, mvt_cdiv(nchunk, 4), 256);
  }
#undef WS
#undef LAUNCH
#undef LAUNCH_T
  return mvt_launch_status();
}
