// Composite entry point of the updater transformer: the launch sequence of EfficientUpdateFormer.forward
// (cotracker2/blocks.py:455-494) as ONE C call.  No kernels of its own: it sequences the library's fused kernels
// (mvt_gemm_bf16, mvt_ln_proj_bf16, mvt_attention_bf16, mvt_block_fused_bf16) on the caller's stream over a caller-provided
// workspace.  Per layer: 4 attention launches (+1 merge) and 6 block launches:
//   time attention -> block over ALL rows (epilogue projects v2p k|v, p2v q for the point rows, v2p q for the virtual rows)
//   virtual<-point attention (keys cut over 4 workgroups + merge) -> virtual block, two-launch split path (-> virtual self q|k|v)
//   virtual self attention -> virtual block (-> p2v k|v and the NEXT layer's time q|k|v of the virtual rows)
//   point<-virtual attention -> point block (-> the next layer's time q|k|v of the point rows)
#include "common.h"
#include <stdlib.h>

namespace {

inline long long align256(long long b) { return (b + 255) & ~255LL; }

struct Layout {
  long long tok, qkv_a, qkv_b, qp, att, split_ws, attn_ws, h1, h2, total;
  int ldh;
};

constexpr int H_ = 256, HEADS = 6, DH_ = 48, INNER = HEADS * DH_, NV = 64, MLP = 1024, OUT = 131;

inline Layout layout(long long n, long long S) {
  const long long Mp = n * S, Mv = (long long)NV * S, M = Mp + Mv;
  Layout L{};
  long long o = 0;
  auto take = [&](long long bytes) { const long long at = o; o += align256(bytes); return at; };
  L.tok = take(M * H_ * 4);
  L.qkv_a = take(M * 3 * INNER * 2);
  L.qkv_b = take(M * 3 * INNER * 2);
  L.qp = take(Mp * INNER * 2);
  L.att = take(M * INNER * 2);
  L.split_ws = take((MLP / 256 + 1) * Mv * H_ * 4);
  L.attn_ws = take(4LL * S * HEADS * 1 * 64 * 68 * 4);  // key-split partials of the 64-query virtual<-point attention
  L.ldh = (OUT + 3) / 4 * 4;
  L.h1 = take(Mp * L.ldh * 4);
  L.h2 = take(Mp * L.ldh * 4);
  L.total = o;
  return L;
}

inline mvt_block_next next_of(const mvt_lin_frag& l, void* y, int ldy, long long lo, long long hi, const float* lnw = nullptr,
                              const float* lnb = nullptr, float eps = 1e-6f) {
  mvt_block_next nx{};
  nx.w = l.w; nx.b = l.b; nx.lnw = lnw; nx.lnb = lnb; nx.y = (float*)y; nx.ldw = l.K; nx.N = l.N; nx.ldy = ldy; nx.eps = eps;
  nx.row_lo = lo; nx.row_hi = hi; nx.y_bf16 = 1;
  return nx;
}

inline bool lin_ok(const mvt_lin_frag& l, int N, int K) { return l.w && l.b && l.N == N && l.K == K; }

}  // namespace

extern "C" long long mvt_updateformer_workspace_bytes(int n, int S) {
  if (n <= 0 || S <= 0) return -1;
  return layout(n, S).total;
}

#define MVT_TRY(call)          \
  do {                         \
    const int rc_ = (call);    \
    if (rc_ != MVT_OK) return rc_; \
  } while (0)

static int updater_run(const mvt_updater_weights* w, const float* x, int ldx, const mvt_token_inputs* ti, int n, float* delta, int ldd,
                       float* coords, float* ffeats, int* nan_flag, void* workspace, long long workspace_bytes, void* stream) {
  MVT_REQUIRE(w && (x || ti) && workspace && n > 0 && (delta || coords));
  if (!x) {
    MVT_REQUIRE(ti->coords && ti->fcorr && ti->ffeats && ti->mask_vis && ti->pos && ti->time_embed && w->input_frag.w);
    MVT_REQUIRE(3 * ti->E + 3 + ti->Fc + ti->Cf + 2 == w->token_dim);
    ldx = (w->token_dim + 3) / 4 * 4;
  }
  MVT_REQUIRE(!coords || (ffeats && w->flow0_frag.w && w->flow2_frag.w && w->flow4_frag.w && w->ffeats_updater.w && w->ffeats_norm_w &&
                          w->ffeats_norm_b));
  MVT_REQUIRE(w->hidden == H_ && w->heads == HEADS && w->dim_head == DH_ && w->n_virtual == NV && w->out_dim == OUT);
  MVT_REQUIRE(w->depth >= 1 && w->depth <= MVT_UPDATER_MAX_DEPTH && w->S >= 1 && w->virtual_tokens);
  const int S = w->S;
  const Layout L = layout(n, S);
  MVT_REQUIRE(workspace_bytes >= L.total && ((uintptr_t)workspace % 256) == 0);
  MVT_REQUIRE(ldx % 4 == 0 && ldx >= w->token_dim && (!delta || ldd >= OUT));
  MVT_REQUIRE(w->input_transform.w && w->input_transform.N == H_ && w->input_transform.K == w->token_dim);
  MVT_REQUIRE(w->flow0.w && w->flow0.N == L.ldh && w->flow0.K == H_ && w->flow2.w && w->flow2.N == L.ldh && w->flow2.K == OUT);
  MVT_REQUIRE(w->flow4.w && w->flow4.N == OUT && w->flow4.K == OUT);
  for (int i = 0; i < w->depth; ++i) {
    MVT_REQUIRE(lin_ok(w->time_blk[i].qkv, 3 * INNER, H_) && lin_ok(w->vself[i].qkv, 3 * INNER, H_));
    MVT_REQUIRE(lin_ok(w->v2p[i].q, INNER, H_) && lin_ok(w->v2p[i].kv, 2 * INNER, H_) && w->v2p[i].ctx_ln_w && w->v2p[i].ctx_ln_b);
    MVT_REQUIRE(lin_ok(w->p2v[i].q, INNER, H_) && lin_ok(w->p2v[i].kv, 2 * INNER, H_) && w->p2v[i].ctx_ln_w && w->p2v[i].ctx_ln_b);
    for (const mvt_updater_block* b : {&w->time_blk[i], &w->v2p[i], &w->vself[i], &w->p2v[i]})
      MVT_REQUIRE(lin_ok(b->out, H_, INNER) && lin_ok(b->fc1, MLP, H_) && lin_ok(b->fc2, H_, MLP));
  }
  char* base = (char*)workspace;
  float* tok = (float*)(base + L.tok);
  unsigned short* qkv = (unsigned short*)(base + L.qkv_a);
  unsigned short* qkv_nx = (unsigned short*)(base + L.qkv_b);
  unsigned short* qp = (unsigned short*)(base + L.qp);
  unsigned short* att = (unsigned short*)(base + L.att);
  float* split_ws = (float*)(base + L.split_ws);
  float* attn_ws = (float*)(base + L.attn_ws);
  float* h1 = (float*)(base + L.h1);
  float* h2 = (float*)(base + L.h2);
  const long long Mp = (long long)n * S, Mv = (long long)NV * S, M = Mp + Mv;
  const int ld3 = 3 * INNER;
  const int BF = MVT_IO_IN_BF16 | MVT_IO_OUT_BF16;
  float* vt = tok + Mp * H_;

  auto block = [&](const mvt_updater_block& b, float* xr, long long rows, const unsigned short* a, const mvt_block_next* nx, int nn,
                   float* ws) {
    return mvt_block_fused_bf16(xr, H_, a, 1, INNER, INNER, b.out.w, INNER, b.out.b, b.fc1.w, H_, b.fc1.b, b.fc2.w, MLP, b.fc2.b, MLP, nx,
                                nn, rows, H_, ws, stream);
  };
  // the same block with its attention computed inside the kernel
  auto attn_block = [&](const mvt_updater_block& b, float* xr, long long rows, int kind, const unsigned short* q, int ldq,
                        const unsigned short* k, const unsigned short* v, int nkeys, const mvt_block_next* nx, int nn, float* ws) {
    mvt_block_attn at{};
    at.kind = kind; at.S = S; at.n_keys = nkeys; at.heads = HEADS; at.dim_head = DH_; at.ldq = ldq; at.ldkv = ld3; at.q = q; at.k = k; at.v = v;
    return mvt_attn_block_fused_bf16(xr, H_, &at, b.out.w, b.out.b, b.fc1.w, b.fc1.b, b.fc2.w, b.fc2.b, MLP, nx, nn, rows, H_, ws, stream);
  };
  const bool fuse_time = (w->fuse_attention & 1) && S <= 32, fuse_p2v = (w->fuse_attention & 2) != 0, fuse_vs = (w->fuse_attention & 4) != 0;
  // bit 5: the virtual-self block's pass 2 runs inside the point<-virtual block (needs both attentions in their block kernels)
  // (the first workgroups of a frame also evaluate the next layer's time q|k|v of the virtual rows, 8 column blocks each: enough tiles)
  // (not in the small-M form: while 32-token tiles of the point<-virtual block fit one round of workgroups -- ceil(n / 32) * S <= 256,
  //  the same rule as in mvt_attn_block_fused_bf16 -- that block runs on them, twice the workgroups, and the virtual-self block keeps
  //  its own second launch: measured faster at 400 / 512 tracks, 958 -> 906 / 929 -> 882 us per call)
  static const bool small_m = !(getenv("MVT_FRAME_NMB1") && atoi(getenv("MVT_FRAME_NMB1")) == 0);
  const bool small_form = small_m && (long long)n * S >= 4096 && (long long)((n + 31) / 32) * S <= 256;
  const bool fold_vs = (w->fuse_attention & 32) && fuse_p2v && fuse_vs && (long long)n * S >= 4096 && !small_form && MLP / 256 <= 4 &&
                       8 * ((n + 63) / 64) >= (3 * INNER + 31) / 32;

  // tokens: input transform of the point rows, learned virtual tokens repeated over the S frames (blocks.py:456-459), and the
  // first time-attention q|k|v projection -- one launch when the fragment-major input weights are available
  {
    const mvt_block_next nx = next_of(w->time_blk[0].qkv, qkv, ld3, 0, 0);
    if (!x) {  // token rows assembled inside the kernel: no token matrix at all
      MVT_REQUIRE(w->token_dim <= 592 && w->input_frag.K == 592);
      MVT_TRY(mvt_token_input_proj_bf16(ti->coords, ti->fcorr, ti->Fc, ti->ffeats, ti->Cf, ti->mask_vis, ti->pos, ti->time_embed, n, S, ti->E,
                                        w->input_frag.w, w->input_frag.b, w->virtual_tokens, tok, H_, &nx, 1, M, H_, stream));
    } else if (w->input_frag.w && w->token_dim <= 592 && w->input_frag.K == 592) {
      MVT_TRY(mvt_input_proj_bf16(x, ldx, w->token_dim, Mp, w->input_frag.w, w->input_frag.b, w->virtual_tokens, S, tok, H_, &nx, 1, M, H_,
                                  stream));
    } else {
      MVT_TRY(mvt_gemm_bf16(x, ldx, w->input_transform.w, nullptr, w->input_transform.ldw, w->input_transform.b, nullptr, 0, tok, H_, (int)Mp,
                            H_, w->token_dim, MVT_ACT_NONE, 0, stream));
      MVT_TRY(mvt_broadcast_rows(w->virtual_tokens, vt, H_, NV, S, H_, stream));
      MVT_TRY(mvt_ln_proj_bf16(tok, H_, &nx, 1, M, H_, stream));
    }
  }
  for (int i = 0; i < w->depth; ++i) {
    const bool last = i + 1 == w->depth;
    const mvt_updater_block &tb = w->time_blk[i], &v2p = w->v2p[i], &vs = w->vself[i], &p2v = w->p2v[i];
    // ---- time attention over the S frames of every (point or virtual) track; group = track, item stride 1 row
    {
      // (fused: the tile's own q|k|v rows are read before its epilogue overwrites them with the v2p projections -- every
      //  workgroup owns whole tracks, and a tile's projection stores follow its attention)
      const mvt_block_next nx[3] = {next_of(v2p.kv, qkv + INNER, ld3, 0, Mp, v2p.ctx_ln_w, v2p.ctx_ln_b, 1e-5f),
                                    next_of(p2v.q, qp, INNER, 0, Mp), next_of(v2p.q, qkv, ld3, Mp, M)};
      if (fuse_time) {
        MVT_TRY(attn_block(tb, tok, M, MVT_ATTN_TIME, qkv, ld3, qkv + INNER, qkv + 2 * INNER, S, nx, 3, nullptr));
      } else {
        MVT_TRY(mvt_attention_bf16(qkv, ld3, S, 1, qkv + INNER, qkv + 2 * INNER, ld3, S, 1, att, INNER, n + NV, S, S, HEADS, DH_, BF, nullptr,
                                   stream));
        MVT_TRY(block(tb, tok, M, att, nx, 3, nullptr));
      }
    }
    // ---- virtual <- point cross attention, per frame: group = frame (stride 1 row), items stride S rows
    unsigned short* qv = qkv + Mp * ld3;  // q|k|v rows of the virtual tokens
    unsigned short* av = att + Mp * INNER;
    // the key-split path of mvt_attention_bf16 needs >= 512 keys in 4 x whole 32-key blocks; otherwise the plain form
    const bool parts = (w->fuse_attention & 16) && n >= 512 && ((n + 31) / 32) % MVT_ATTN_NSPLIT == 0;
    MVT_TRY(mvt_attention_bf16(qv, ld3, 1, S, qkv + INNER, qkv + 2 * INNER, ld3, 1, S, av, INNER, S, NV, n, HEADS, DH_,
                               BF | (parts ? MVT_ATTN_PARTIALS_ONLY : 0), attn_ws, stream));
    {
      const mvt_block_next nx = next_of(vs.qkv, qv, ld3, 0, 0);
      if (parts) {
        mvt_block_attn at{};
        at.kind = MVT_ATTN_PARTIALS; at.S = S; at.n_keys = NV; at.heads = HEADS; at.dim_head = DH_; at.partials = attn_ws;
        at.n_splits = MVT_ATTN_NSPLIT;
        MVT_TRY(mvt_attn_block_fused_bf16(vt, H_, &at, v2p.out.w, v2p.out.b, v2p.fc1.w, v2p.fc1.b, v2p.fc2.w, v2p.fc2.b, MLP, &nx, 1, Mv, H_,
                                          split_ws, stream));
      } else {
        MVT_TRY(block(v2p, vt, Mv, av, &nx, 1, split_ws));
      }
    }
    // ---- virtual self attention, per frame
    {
      mvt_block_next nx[2] = {next_of(p2v.kv, qv + INNER, ld3, 0, 0, p2v.ctx_ln_w, p2v.ctx_ln_b, 1e-5f), {}};
      int nn = 1;
      if (!last) nx[nn++] = next_of(w->time_blk[i + 1].qkv, qkv_nx + Mp * ld3, ld3, 0, 0);  // virtual rows are final for this layer
      if (fold_vs) {
        // pass 1 only: the block is finished -- x, the p2v k|v and the next layer's time q|k|v of the virtual rows -- by the
        // workgroups of the point<-virtual block below (MVT_ATTN_FRAME_CTX): one launch less on the serial chain
        mvt_block_attn at{};
        at.kind = MVT_ATTN_FRAME; at.S = S; at.n_keys = NV; at.heads = HEADS; at.dim_head = DH_; at.ldq = ld3; at.ldkv = ld3;
        at.q = qv; at.k = qv + INNER; at.v = qv + 2 * INNER; at.defer_pass2 = 1;
        MVT_TRY(mvt_attn_block_fused_bf16(vt, H_, &at, vs.out.w, vs.out.b, vs.fc1.w, vs.fc1.b, vs.fc2.w, vs.fc2.b, MLP, nx, nn, Mv, H_, split_ws,
                                          stream));
      } else if (fuse_vs) {
        // (pass 1 reads the virtual q|k|v; the p2v k|v projection that overwrites k|v is written by pass 2, a later launch)
        MVT_TRY(attn_block(vs, vt, Mv, MVT_ATTN_FRAME, qv, ld3, qv + INNER, qv + 2 * INNER, NV, nx, nn, split_ws));
      } else {
        MVT_TRY(mvt_attention_bf16(qv, ld3, 1, S, qv + INNER, qv + 2 * INNER, ld3, 1, S, av, INNER, S, NV, NV, HEADS, DH_, BF, nullptr, stream));
        MVT_TRY(block(vs, vt, Mv, av, nx, nn, split_ws));
      }
    }
    // ---- point <- virtual cross attention, per frame
    {
      const mvt_block_next nx = last ? mvt_block_next{} : next_of(w->time_blk[i + 1].qkv, qkv_nx, ld3, 0, 0);
      if (fold_vs) {
        mvt_block_ctx cx{};
        cx.ws = split_ws; cx.chunks = MLP / 256; cx.b2 = vs.fc2.b; cx.x = vt; cx.ldx = H_;
        cx.kv = next_of(p2v.kv, nullptr, ld3, 0, 0, p2v.ctx_ln_w, p2v.ctx_ln_b, 1e-5f);
        if (!last) cx.next = next_of(w->time_blk[i + 1].qkv, qkv_nx + Mp * ld3, ld3, 0, 0);
        mvt_block_attn at{};
        at.kind = MVT_ATTN_FRAME_CTX; at.S = S; at.n_keys = NV; at.heads = HEADS; at.dim_head = DH_; at.ldq = INNER; at.ldkv = ld3;
        at.q = qp; at.ctx = &cx;
        MVT_TRY(mvt_attn_block_fused_bf16(tok, H_, &at, p2v.out.w, p2v.out.b, p2v.fc1.w, p2v.fc1.b, p2v.fc2.w, p2v.fc2.b, MLP, &nx, last ? 0 : 1,
                                          Mp, H_, nullptr, stream));
      } else if (fuse_p2v && Mp >= 4096) {
        MVT_TRY(attn_block(p2v, tok, Mp, MVT_ATTN_FRAME, qp, INNER, qv + INNER, qv + 2 * INNER, NV, &nx, last ? 0 : 1, nullptr));
      } else {
        MVT_TRY(mvt_attention_bf16(qp, INNER, 1, S, qv + INNER, qv + 2 * INNER, ld3, 1, S, att, INNER, S, n, NV, HEADS, DH_, BF, nullptr, stream));
        MVT_TRY(block(p2v, tok, Mp, att, &nx, last ? 0 : 1, nullptr));
      }
    }
    unsigned short* t_ = qkv; qkv = qkv_nx; qkv_nx = t_;
  }
  if (coords)  // flow head + track / feature update in one launch; delta only when the caller wants to see it
    return mvt_update_head_bf16(tok, H_, w->flow0_frag.w, w->flow0_frag.b, w->flow2_frag.w, w->flow2_frag.b, w->flow4_frag.w, w->flow4_frag.b,
                                w->ffeats_norm_w, w->ffeats_norm_b, w->ffeats_updater.w, w->ffeats_updater.b, coords, ffeats, delta, ldd, Mp,
                                H_, OUT, nan_flag, stream);
  // flow head (blocks.py:489): 256 -> 131 -> 131 -> 131 with ReLU; hidden activations padded to ldh columns
  MVT_TRY(mvt_gemm_bf16(tok, H_, w->flow0.w, nullptr, w->flow0.ldw, w->flow0.b, nullptr, 0, h1, L.ldh, (int)Mp, L.ldh, H_, MVT_ACT_RELU, 0, stream));
  MVT_TRY(mvt_gemm_bf16(h1, L.ldh, w->flow2.w, nullptr, w->flow2.ldw, w->flow2.b, nullptr, 0, h2, L.ldh, (int)Mp, L.ldh, OUT, MVT_ACT_RELU, 0, stream));
  MVT_TRY(mvt_gemm_bf16(h2, L.ldh, w->flow4.w, nullptr, w->flow4.ldw, w->flow4.b, nullptr, 0, delta, ldd, (int)Mp, OUT, OUT, MVT_ACT_NONE, 0, stream));
  return MVT_OK;
}

extern "C" int mvt_updateformer_forward(const mvt_updater_weights* w, const float* x, int ldx, int n, float* delta, int ldd,
                                        float* coords, float* ffeats, int* nan_flag, void* workspace, long long workspace_bytes,
                                        void* stream) {
  MVT_REQUIRE(x);
  return updater_run(w, x, ldx, nullptr, n, delta, ldd, coords, ffeats, nan_flag, workspace, workspace_bytes, stream);
}

extern "C" int mvt_updateformer_forward_tokens(const mvt_updater_weights* w, const mvt_token_inputs* tokens, int n, float* delta, int ldd,
                                               float* coords, float* ffeats, int* nan_flag, void* workspace, long long workspace_bytes,
                                               void* stream) {
  MVT_REQUIRE(tokens);
  return updater_run(w, nullptr, 0, tokens, n, delta, ldd, coords, ffeats, nan_flag, workspace, workspace_bytes, stream);
}
