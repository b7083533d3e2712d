// Student forward / backward step (BASELINE cfg 4) and the dimension-generic encoder forward.
//
// Replaces what the reference's KD training step runs inside torch autograd for
// StudentModel.encode_with_gradients (reference: src/kd/train.py:176-210: two encodes with
// gradients -> q @ d.T -> CombinedKDLoss -> backward): a post-LN BERT encoder forward that SAVES
// its activations, and the matching backward producing fp32 parameter gradients.  bf16
// activations and MFMA operands, fp32 accumulation, fp32 LayerNorm / softmax statistics, fp32
// gradients.  Dimension-generic (hidden <= 1024, any head count / width that are multiples of 32):
// the teacher cross-encoder (XLM-R-large shape) runs the same forward.
//
// Activations are ROW-MAJOR [tokens, features] here (the inference encoder's fragment order is
// tied to hidden 384).  Every GEMM is the one NT kernel of generic.hip; operands whose reduction
// dimension is not contiguous are transposed first by a bandwidth-bound kernel.
#include "generic.h"

#include <cmath>
#include <vector>

using namespace sskd_generic;

namespace {

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Dims {
  int B, S, H, NH, DH, F, L;
  int64_t M;
  bool training;  // false: nothing is kept for a backward pass (fused attention, GELU in the GEMM epilogue)
};

struct LayerSaved {
  bf16_t *qkv, *P, *ctx, *z1, *x1, *u, *hmid, *z2, *x2;
  float *mean1, *rstd1, *mean2, *rstd2;
  float* lse;  // fused attention (training): log-sum-exp per (row, head, query) instead of P
};

struct Saved {
  bf16_t *z0, *x0;
  float *mean0, *rstd0, *pooled;
  LayerSaved* layer;  // host array
  // scratch
  bf16_t *tH0, *tH1, *tH2, *vt, *tF0, *tP0, *tP1, *tA, *tB, *t3H;
  size_t bytes;
};

// carve the workspace; `layers_out` must hold cfg->layers entries (host memory)
Saved carve(void* base, const Dims& d, LayerSaved* layers_out, bool training) {
  char* p = static_cast<char*>(base);
  auto take_b = [&](size_t elems) {
    bf16_t* r = reinterpret_cast<bf16_t*>(p);
    p += align256(elems * sizeof(bf16_t));
    return r;
  };
  auto take_f = [&](size_t elems) {
    float* r = reinterpret_cast<float*>(p);
    p += align256(elems * sizeof(float));
    return r;
  };
  const size_t M = (size_t)d.M, MH = M * d.H, MF = M * d.F, PP = (size_t)d.B * d.NH * d.S * d.S;
  Saved s{};
  s.layer = layers_out;
  // training with the fused attention kernels keeps lse [B, heads, S] instead of the S x S probabilities
  const bool flash = training && attention_bwd_supported(d.S, d.DH);
  s.z0 = take_b(MH);
  s.x0 = take_b(MH);
  s.mean0 = take_f(M);
  s.rstd0 = take_f(M);
  s.pooled = take_f((size_t)d.B * d.H);
  const int nl = training ? d.L : (d.L > 0 ? 1 : 0);  // inference re-uses one layer's buffers
  for (int l = 0; l < nl; ++l) {
    LayerSaved& ls = layers_out[l];
    ls.qkv = take_b(3 * MH);
    ls.P = (training && !flash) ? take_b(PP) : nullptr;  // fused attention: no score matrix
    ls.lse = flash ? take_f((size_t)d.B * d.NH * d.S) : nullptr;
    ls.ctx = take_b(MH);
    ls.z1 = training ? take_b(MH) : nullptr;
    ls.x1 = take_b(MH);
    ls.u = training ? take_b(MF) : nullptr;     // inference: GELU in the GEMM epilogue
    ls.hmid = take_b(MF);
    ls.z2 = training ? take_b(MH) : nullptr;
    ls.x2 = take_b(MH);
    ls.mean1 = take_f(M);
    ls.rstd1 = take_f(M);
    ls.mean2 = take_f(M);
    ls.rstd2 = take_f(M);
  }
  if (!training && d.L > 1) {
    // ping-pong the layer output so that layer l reads x2 of layer l-1 while writing its own
    for (int l = 1; l < d.L; ++l) {
      layers_out[l] = layers_out[0];
    }
    bf16_t* alt = take_b(MH);
    for (int l = 1; l < d.L; l += 2) layers_out[l].x2 = alt;
  }
  s.tH0 = take_b(MH);
  s.vt = training ? take_b(MH) : nullptr;
  if (training) {
    s.tH1 = take_b(MH);
    s.tH2 = take_b(MH);
    s.tF0 = take_b(MF);
    s.tP0 = flash ? nullptr : take_b(PP);
    s.tP1 = flash ? nullptr : take_b(PP);
    const size_t wide = (size_t)(3 * d.H > d.F ? 3 * d.H : d.F);
    s.tA = take_b(wide * M);
    s.tB = take_b((size_t)(d.H > d.F ? d.H : d.F) * M);
    s.t3H = take_b(3 * MH);
  }
  s.bytes = (size_t)(p - static_cast<char*>(base));
  return s;
}

int check(const sskd_generic_config* cfg, const sskd_generic_weights* w, int B, int S, Dims* d) {
  SSKD_REQUIRE(cfg && w, "generic encoder: null config / weights");
  SSKD_REQUIRE(cfg->hidden > 0 && cfg->hidden % 32 == 0 && cfg->hidden <= 1024,
               "generic encoder: hidden=%d must be a multiple of 32, at most 1024", cfg->hidden);
  SSKD_REQUIRE(cfg->heads > 0 && cfg->hidden % cfg->heads == 0 && (cfg->hidden / cfg->heads) % 32 == 0,
               "generic encoder: head width %d/%d must be a multiple of 32", cfg->hidden, cfg->heads);
  SSKD_REQUIRE(cfg->intermediate > 0 && cfg->intermediate % 32 == 0, "generic encoder: intermediate must be a multiple of 32");
  SSKD_REQUIRE(cfg->layers >= 0 && cfg->vocab_size > 0, "generic encoder: bad layers / vocab");
  SSKD_REQUIRE(B >= 0 && S >= 32 && S % 32 == 0 && S <= 512, "generic encoder: S=%d must be a multiple of 32 in [32, 512]", S);
  SSKD_REQUIRE(S + cfg->pos_offset <= cfg->max_positions, "generic encoder: S=%d + offset %d exceeds max_positions %d", S,
               cfg->pos_offset, cfg->max_positions);
  d->B = B;
  d->S = S;
  d->H = cfg->hidden;
  d->NH = cfg->heads;
  d->DH = cfg->hidden / cfg->heads;
  d->F = cfg->intermediate;
  d->L = cfg->layers;
  d->M = (int64_t)B * S;
  return SSKD_OK;
}

// C[M, N] = A[M, K] B[N, K]^T (+ bias), plain 2-D
int gemm(const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int N, int K,
         const float* bias, bool c_f32, bool acc, hipStream_t st) {
  GemmArgs g{};
  g.A = A;
  g.B = B;
  g.C = C;
  g.bias = bias;
  g.M = (int)M;
  g.N = N;
  g.K = K;
  g.lda = lda;
  g.ldb = ldb;
  g.ldc = ldc;
  g.batch1 = g.batch2 = 1;
  g.alpha = 1.0f;
  g.c_is_f32 = c_f32;
  g.accumulate = acc;
  // weight-gradient products (few output tiles, K = every token of the step): cut K over enough
  // workgroups to fill the chip; the slices meet through fp32 atomics
  if (acc && c_f32 && !bias) {
    const int64_t tiles = sskd::ceil_div(M, 128) * sskd::ceil_div(N, 128);
#ifndef SSKD_SPLITK_TARGET
#define SSKD_SPLITK_TARGET 256  // one workgroup per CU: every extra slice adds a 128 x 128 tile of device-scope atomics (1024: 104 us, 256: 65 us for 384 x 384 x 65 k)
#endif
    int split = (int)(SSKD_SPLITK_TARGET / (tiles > 0 ? tiles : 1));
    const int max_split = K / 256;  // at least 256 of K per slice
    if (split > max_split) split = max_split;
    g.split_k = split > 1 ? split : 1;
  }
  return launch_gemm_nt(g, st);
}

int transpose2d(const bf16_t* in, int64_t R, int C, int64_t ld_in, bf16_t* out, int64_t ld_out, hipStream_t st,
                float* colsum = nullptr) {
  TransposeArgs t{};
  t.colsum = colsum;
  t.in = in;
  t.out = out;
  t.R = (int)R;
  t.C = C;
  t.ld_in = ld_in;
  t.ld_out = ld_out;
  t.batch1 = t.batch2 = 1;
  return launch_transpose(t, st);
}

int gemm(const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int N, int K,
         const float* bias, bool c_f32, bool acc, hipStream_t st);

// dW[M, N] += dY[T, M]^T X[T, N] and db[M] += column sums of dY.  The student's shapes go to the TN kernel, which
// reads both operands as they lie in memory; others are transposed into tA / tB and take the NT kernel.
int weight_grad(const bf16_t* dY, int M, const bf16_t* X, int N, int64_t T, float* dW, float* db, bf16_t* tA, bf16_t* tB,
                hipStream_t st) {
  if (gemm_tn_supported(T, M, N, M, N)) {
    if (db) {  // nullptr: the kernel that produced dY already summed its columns
      int rc = launch_colsum(dY, T, M, M, db, st);
      if (rc != SSKD_OK) return rc;
    }
    return launch_gemm_tn(dY, M, X, N, dW, N, T, M, N, st);
  }
  int rc = transpose2d(dY, T, M, M, tA, T, st, db);  // [M, T]; db = column sums on the way
  if (rc != SSKD_OK) return rc;
  rc = transpose2d(X, T, N, N, tB, T, st);            // [N, T]
  if (rc != SSKD_OK) return rc;
  return gemm(tA, T, tB, T, dW, N, M, N, (int)T, nullptr, true, true, st);
}

#define TRY(expr)                     \
  do {                                \
    int rc_ = (expr);                 \
    if (rc_ != SSKD_OK) return rc_;   \
  } while (0)

// ---- forward of one layer: x -> ls.x2, saving what the backward needs -------------------------
// everything after attention of a TRAINING forward (activations saved): out-projection, LN, FFN, LN
int layer_forward_tail(const Dims& d, const sskd_generic_layer_weights& lw, float eps, const bf16_t* x, LayerSaved& ls,
                       Saved& sv, hipStream_t st) {
  const int H = d.H, F = d.F;
  const int64_t M = d.M;
  TRY(gemm(ls.ctx, H, static_cast<const bf16_t*>(lw.wo), H, sv.tH0, H, M, H, H, lw.bo, false, false, st));
  TRY(launch_add_ln_fwd(x, sv.tH0, lw.ln1_g, lw.ln1_b, eps, M, H, ls.x1, ls.z1, ls.mean1, ls.rstd1, st));
  TRY(gemm(ls.x1, H, static_cast<const bf16_t*>(lw.w1), H, ls.u, F, M, F, H, lw.b1, false, false, st));
  TRY(launch_gelu_fwd(ls.u, ls.hmid, M * F, st));
  TRY(gemm(ls.hmid, F, static_cast<const bf16_t*>(lw.w2), F, sv.tH0, H, M, H, F, lw.b2, false, false, st));
  TRY(launch_add_ln_fwd(ls.x1, sv.tH0, lw.ln2_g, lw.ln2_b, eps, M, H, ls.x2, ls.z2, ls.mean2, ls.rstd2, st));
  return SSKD_OK;
}

int layer_forward(const Dims& d, const sskd_generic_layer_weights& lw, float eps, const bf16_t* x, const int32_t* mask,
                  LayerSaved& ls, Saved& sv, hipStream_t st) {
  const int H = d.H, S = d.S, DH = d.DH, NH = d.NH, F = d.F;
  const int64_t M = d.M;
  TRY(gemm(x, H, static_cast<const bf16_t*>(lw.wqkv), H, ls.qkv, 3 * H, M, 3 * H, H, lw.bqkv, false, false, st));
  if (!d.training) {
    // inference: scores, softmax and P V in one kernel (no [B, heads, S, S] matrix), GELU inside FFN1's epilogue
    TRY(launch_attention_fwd(ls.qkv, mask, d.B, S, NH, DH, 1.0f / sqrtf((float)DH), ls.ctx, nullptr, st));
    TRY(gemm(ls.ctx, H, static_cast<const bf16_t*>(lw.wo), H, sv.tH0, H, M, H, H, lw.bo, false, false, st));
    TRY(launch_add_ln_fwd(x, sv.tH0, lw.ln1_g, lw.ln1_b, eps, M, H, ls.x1, nullptr, nullptr, nullptr, st));
    GemmArgs f1{};
    f1.A = ls.x1;
    f1.B = static_cast<const bf16_t*>(lw.w1);
    f1.C = ls.hmid;
    f1.bias = lw.b1;
    f1.M = (int)M;
    f1.N = F;
    f1.K = H;
    f1.lda = H;
    f1.ldb = H;
    f1.ldc = F;
    f1.batch1 = f1.batch2 = 1;
    f1.alpha = 1.0f;
    f1.act = 1;
    TRY(launch_gemm_nt(f1, st));
    TRY(gemm(ls.hmid, F, static_cast<const bf16_t*>(lw.w2), F, sv.tH0, H, M, H, F, lw.b2, false, false, st));
    TRY(launch_add_ln_fwd(ls.x1, sv.tH0, lw.ln2_g, lw.ln2_b, eps, M, H, ls.x2, nullptr, nullptr, nullptr, st));
    return SSKD_OK;
  }
  if (ls.lse) {
    // fused attention that keeps the log-sum-exp for the fused backward
    TRY(launch_attention_fwd(ls.qkv, mask, d.B, S, NH, DH, 1.0f / sqrtf((float)DH), ls.ctx, ls.lse, st));
    return layer_forward_tail(d, lw, eps, x, ls, sv, st);
  }
  // scores[b, h] = Q_bh K_bh^T
  GemmArgs g{};
  g.A = ls.qkv;
  g.B = ls.qkv + H;
  g.C = ls.P;
  g.M = S;
  g.N = S;
  g.K = DH;
  g.lda = g.ldb = 3 * H;
  g.ldc = S;
  g.batch1 = d.B;
  g.batch2 = NH;
  g.sA1 = g.sB1 = (int64_t)S * 3 * H;
  g.sA2 = g.sB2 = DH;
  g.sC1 = (int64_t)NH * S * S;
  g.sC2 = (int64_t)S * S;
  g.alpha = 1.0f;
  TRY(launch_gemm_nt(g, st));
  TRY(launch_softmax_fwd(ls.P, mask, d.B, NH, S, 1.0f / sqrtf((float)DH), st));
  // V_bh^T [DH, S]
  TransposeArgs t{};
  t.in = ls.qkv + 2 * H;
  t.out = sv.vt;
  t.R = S;
  t.C = DH;
  t.ld_in = 3 * H;
  t.ld_out = S;
  t.batch1 = d.B;
  t.batch2 = NH;
  t.sI1 = (int64_t)S * 3 * H;
  t.sI2 = DH;
  t.sO1 = (int64_t)NH * DH * S;
  t.sO2 = (int64_t)DH * S;
  TRY(launch_transpose(t, st));
  // ctx_bh = P_bh V_bh
  GemmArgs c{};
  c.A = ls.P;
  c.B = sv.vt;
  c.C = ls.ctx;
  c.M = S;
  c.N = DH;
  c.K = S;
  c.lda = S;
  c.ldb = S;
  c.ldc = H;
  c.batch1 = d.B;
  c.batch2 = NH;
  c.sA1 = (int64_t)NH * S * S;
  c.sA2 = (int64_t)S * S;
  c.sB1 = (int64_t)NH * DH * S;
  c.sB2 = (int64_t)DH * S;
  c.sC1 = (int64_t)S * H;
  c.sC2 = DH;
  c.alpha = 1.0f;
  TRY(launch_gemm_nt(c, st));
  return layer_forward_tail(d, lw, eps, x, ls, sv, st);
}

int forward_all(const sskd_generic_config* cfg, const sskd_generic_weights* w, const Dims& d, const int32_t* ids,
                const int32_t* mask, Saved& sv, hipStream_t st, const bf16_t** final_hidden) {
  TRY(launch_embed_fwd(ids, mask, static_cast<const bf16_t*>(w->word_emb), static_cast<const bf16_t*>(w->pos_emb),
                       static_cast<const bf16_t*>(w->type_emb), d.B, d.S, d.H, cfg->vocab_size, cfg->pos_offset, sv.z0, st));
  TRY(launch_add_ln_fwd(sv.z0, nullptr, w->emb_ln_g, w->emb_ln_b, cfg->layer_norm_eps, d.M, d.H, sv.x0, sv.z0, sv.mean0,
                        sv.rstd0, st));
  const bf16_t* x = sv.x0;
  for (int l = 0; l < d.L; ++l) {
    TRY(layer_forward(d, w->layers[l], cfg->layer_norm_eps, x, mask, sv.layer[l], sv, st));
    x = sv.layer[l].x2;
  }
  *final_hidden = x;
  return SSKD_OK;
}

// last stage of a layer's backward: qkv = x Wqkv^T + bqkv.  The layer's input gradient is sv.tH1 + dz1 (dz1 = the
// residual branch of LN1, in sv.tH0): the SUM is formed by the consumer - the LayerNorm backward that opens the layer
// below (or the embedding LayerNorm's) - not by a pass of its own.
int layer_backward_qkv(const Dims& d, const sskd_generic_layer_weights& lw, const sskd_generic_layer_grads& gw,
                       const bf16_t* x_in, Saved& sv, bf16_t* dqkv, hipStream_t st) {
  const int H = d.H;
  const int64_t M = d.M;
  TRY(weight_grad(dqkv, 3 * H, x_in, H, M, gw.wqkv, gw.bqkv, sv.tA, sv.tB, st));
  TRY(gemm(dqkv, 3 * H, static_cast<const bf16_t*>(lw.wqkv_t), 3 * H, sv.tH1, H, M, H, 3 * H, nullptr, false, false, st));
  return SSKD_OK;
}

// ---- backward of one layer: dx2 + dx2b (gradient of the layer output; dx2b optional) -> gradient of its input ----
// The result is sv.tH1 + sv.tH0 (the caller hands both to the next consumer).
int layer_backward(const Dims& d, const sskd_generic_layer_weights& lw, const sskd_generic_layer_grads& gw,
                   const bf16_t* x_in, const int32_t* mask, const LayerSaved& ls, Saved& sv, const bf16_t* dx2,
                   const bf16_t* dx2b, hipStream_t st) {
  const int H = d.H, S = d.S, DH = d.DH, NH = d.NH, F = d.F;
  const int64_t M = d.M;
  bf16_t* dz2 = sv.tH0;   // may alias dx2b: ln_bwd reads a row before it writes it
  TRY(launch_ln_bwd(dx2, ls.z2, ls.mean2, ls.rstd2, lw.ln2_g, M, H, dz2, gw.ln2_g, gw.ln2_b, st, gw.b2, dx2b));  // + db2
  // y = hmid W2^T + b2
  TRY(weight_grad(dz2, H, ls.hmid, F, M, gw.w2, nullptr, sv.tA, sv.tB, st));
  TRY(gemm(dz2, H, static_cast<const bf16_t*>(lw.w2_t), H, sv.tF0, F, M, F, H, nullptr, false, false, st));  // dhmid
  TRY(launch_gelu_bwd_colsum(ls.u, sv.tF0, sv.tF0, gw.b1, M, F, st));  // du (in place) + db1
  // u = x1 W1^T + b1
  TRY(weight_grad(sv.tF0, F, ls.x1, H, M, gw.w1, nullptr, sv.tA, sv.tB, st));
  bf16_t* dx1 = sv.tH1;
  TRY(gemm(sv.tF0, F, static_cast<const bf16_t*>(lw.w1_t), F, dx1, H, M, H, F, nullptr, false, false, st));
  bf16_t* dz1 = sv.tH0;   // aliases dz2, the residual branch of LN2 that this call adds to dx1 on the fly
  TRY(launch_ln_bwd(dx1, ls.z1, ls.mean1, ls.rstd1, lw.ln1_g, M, H, dz1, gw.ln1_g, gw.ln1_b, st, gw.bo, dz2));  // + dbo
  // attn_out = ctx Wo^T + bo
  TRY(weight_grad(dz1, H, ls.ctx, H, M, gw.wo, nullptr, sv.tA, sv.tB, st));
  bf16_t* dctx = sv.tH2;
  TRY(gemm(dz1, H, static_cast<const bf16_t*>(lw.wo_t), H, dctx, H, M, H, H, nullptr, false, false, st));

  // ---- attention, per (batch row, head) ----
  if (ls.lse) {
    bf16_t* dqkv_f = sv.t3H;
    TRY(launch_attention_bwd(ls.qkv, mask, ls.ctx, dctx, ls.lse, d.B, S, NH, DH, 1.0f / sqrtf((float)DH), dqkv_f, st));
    return layer_backward_qkv(d, lw, gw, x_in, sv, dqkv_f, st);
  }
  const int64_t bS3H = (int64_t)S * 3 * H, bPP = (int64_t)NH * S * S, hPP = (int64_t)S * S;
  // dP = dctx_bh V_bh^T
  GemmArgs g{};
  g.A = dctx;
  g.lda = H;
  g.sA1 = (int64_t)S * H;
  g.sA2 = DH;
  g.B = ls.qkv + 2 * H;
  g.ldb = 3 * H;
  g.sB1 = bS3H;
  g.sB2 = DH;
  g.C = sv.tP0;
  g.ldc = S;
  g.sC1 = bPP;
  g.sC2 = hPP;
  g.M = S;
  g.N = S;
  g.K = DH;
  g.batch1 = d.B;
  g.batch2 = NH;
  g.alpha = 1.0f;
  TRY(launch_gemm_nt(g, st));
  // dV_bh = P_bh^T dctx_bh: needs P^T [S(j), S(i)] and dctx_bh^T [DH, S(i)]
  TransposeArgs t{};
  t.in = ls.P;
  t.out = sv.tP1;
  t.R = S;
  t.C = S;
  t.ld_in = t.ld_out = S;
  t.batch1 = d.B;
  t.batch2 = NH;
  t.sI1 = t.sO1 = bPP;
  t.sI2 = t.sO2 = hPP;
  TRY(launch_transpose(t, st));
  TransposeArgs tc{};
  tc.in = dctx;
  tc.out = sv.vt;  // [B, H, S]
  tc.R = S;
  tc.C = H;
  tc.ld_in = H;
  tc.ld_out = S;
  tc.batch1 = d.B;
  tc.batch2 = 1;
  tc.sI1 = (int64_t)S * H;
  tc.sO1 = (int64_t)H * S;
  TRY(launch_transpose(tc, st));
  bf16_t* dqkv = sv.t3H;
  GemmArgs gv{};
  gv.A = sv.tP1;
  gv.lda = S;
  gv.sA1 = bPP;
  gv.sA2 = hPP;
  gv.B = sv.vt;
  gv.ldb = S;
  gv.sB1 = (int64_t)H * S;
  gv.sB2 = (int64_t)DH * S;
  gv.C = dqkv + 2 * H;
  gv.ldc = 3 * H;
  gv.sC1 = bS3H;
  gv.sC2 = DH;
  gv.M = S;
  gv.N = DH;
  gv.K = S;
  gv.batch1 = d.B;
  gv.batch2 = NH;
  gv.alpha = 1.0f;
  TRY(launch_gemm_nt(gv, st));
  // dS = scale * P * (dP - rowsum(dP P))
  TRY(launch_softmax_bwd(sv.tP0, ls.P, (int64_t)d.B * NH * S, S, 1.0f / sqrtf((float)DH), st));
  // Q^T, K^T, V^T of every row: qkv [S, 3H] -> [3H, S]
  TransposeArgs tq{};
  tq.in = ls.qkv;
  tq.out = sv.tA;  // [B, 3H, S]
  tq.R = S;
  tq.C = 3 * H;
  tq.ld_in = 3 * H;
  tq.ld_out = S;
  tq.batch1 = d.B;
  tq.batch2 = 1;
  tq.sI1 = bS3H;
  tq.sO1 = (int64_t)3 * H * S;
  TRY(launch_transpose(tq, st));
  // dQ_bh = dS K_bh  (B operand = K_bh^T [DH, S])
  GemmArgs gq{};
  gq.A = sv.tP0;
  gq.lda = S;
  gq.sA1 = bPP;
  gq.sA2 = hPP;
  gq.B = sv.tA + (int64_t)H * S;
  gq.ldb = S;
  gq.sB1 = (int64_t)3 * H * S;
  gq.sB2 = (int64_t)DH * S;
  gq.C = dqkv;
  gq.ldc = 3 * H;
  gq.sC1 = bS3H;
  gq.sC2 = DH;
  gq.M = S;
  gq.N = DH;
  gq.K = S;
  gq.batch1 = d.B;
  gq.batch2 = NH;
  gq.alpha = 1.0f;
  TRY(launch_gemm_nt(gq, st));
  // dK_bh = dS^T Q_bh  (A = dS^T, B operand = Q_bh^T [DH, S])
  TransposeArgs ts = t;
  ts.in = sv.tP0;
  ts.out = sv.tP1;
  TRY(launch_transpose(ts, st));
  GemmArgs gk = gq;
  gk.A = sv.tP1;
  gk.B = sv.tA;
  gk.C = dqkv + H;
  TRY(launch_gemm_nt(gk, st));
  return layer_backward_qkv(d, lw, gw, x_in, sv, dqkv, st);
}

// Cross-encoder head, one workgroup per sequence, ALL in fp32 (RobertaClassificationHead with one label):
//   logit = out_w . tanh(dense_w . h[<s>] + dense_b) + out_b
// A reranker's product is an ordering of near-equal logits, and the head is 2 H^2 FLOPs per pair (0.002 % of the
// encoder's work): there is nothing to gain from bf16 here, and the round-2 bf16 head (bf16 dense output, bf16
// tanh, bf16 weights) added its own rounding on top of the encoder's.  16 waves per workgroup; a wave computes dense
// rows w, w + 16, ... FOUR at a time (16-byte loads of whole fp32 rows, the <s> state in LDS; the first version - 4
// waves, one row and 4-byte loads at a time - was latency-bound at 1.4 ms per launch), then a fixed-order block
// reduction, so a logit is bit-reproducible from run to run.
constexpr int HEAD_WAVES = 16;
__global__ __launch_bounds__(HEAD_WAVES * 64) void teacher_head_kernel(const bf16_t* __restrict__ hidden, int S, int H,
                                                                      const float* __restrict__ dense_w,
                                                                      const float* __restrict__ dense_b,
                                                                      const float* __restrict__ out_w,
                                                                      const float* __restrict__ out_b, float* __restrict__ logits) {
  __shared__ __attribute__((aligned(16))) float xs[1024];
  __shared__ float part[HEAD_WAVES];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bf16_t* h = hidden + (int64_t)b * S * H;   // token 0 of sequence b
  for (int c = tid; c < H; c += HEAD_WAVES * 64) xs[c] = (float)h[c];
  __syncthreads();
  float acc = 0.f;   // this wave's share of sum_r out_w[r] tanh(...), rows in increasing order
  const int H4 = H / 4;   // H % 32 == 0
  for (int r0 = 4 * wave; r0 < H; r0 += 4 * HEAD_WAVES) {
    float d[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = lane; c < H4; c += 64) {
      const float4 x = reinterpret_cast<const float4*>(xs)[c];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 w = reinterpret_cast<const float4*>(dense_w + (int64_t)(r0 + u) * H)[c];
        d[u] = fmaf(w.x, x.x, fmaf(w.y, x.y, fmaf(w.z, x.z, fmaf(w.w, x.w, d[u]))));
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) d[u] += __shfl_xor(d[u], o);
      acc = fmaf(out_w[r0 + u], tanhf(d[u] + dense_b[r0 + u]), acc);   // identical in every lane of the wave
    }
  }
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int w = 0; w < HEAD_WAVES; ++w) s += part[w];
    logits[b] = s + out_b[0];
  }
}

}  // namespace

extern "C" {

static size_t workspace_one_part(const sskd_generic_config* cfg, int B, int S, int training) {
  Dims d{};
  sskd_generic_weights dummy{};
  if (!cfg || B <= 0 || check(cfg, &dummy, B, S, &d) != SSKD_OK) return 0;
  std::vector<LayerSaved> tmp((size_t)(d.L > 0 ? d.L : 1));
  return carve(nullptr, d, tmp.data(), training != 0).bytes;
}

// Batches of >= 2 x 16 384 tokens run as TWO halves on two streams (sskd::run_parts_on_streams: a side stream forked from /
// joined into the caller's; rows do not interact).  The rule depends on (cfg, B, S, training) alone, so the forward that
// saves activations and the backward that reads them cut the batch - and the workspace - the same way.  Each half keeps
// its token count a multiple of 256 (the 256-row GEMM tiles).  Training splits only where every weight gradient goes
// through the atomically accumulating product (gemm_tn_supported): the halves add into the same gradient buffers.
static int generic_parts(const sskd_generic_config* cfg, int B, int S, int training) {
  if (sskd::forward_stream_parts() < 2 || !cfg || B < 2 || B % 2 != 0) return 1;
  const int64_t T = (int64_t)(B / 2) * S;
  if (T % 256 != 0 || T < 16384) return 1;
  if (training) {
    const int H = cfg->hidden, F = cfg->intermediate;
    if (!(gemm_tn_supported(T, 3 * H, H, 3 * H, H) && gemm_tn_supported(T, H, H, H, H) && gemm_tn_supported(T, F, H, F, H) &&
          gemm_tn_supported(T, H, F, H, F)))
      return 1;
  }
  return 2;
}
static size_t part_stride(const sskd_generic_config* cfg, int B, int S, int training) {   // bytes between the halves
  return (workspace_one_part(cfg, B / 2, S, training) + 255) & ~(size_t)255;
}

size_t sskd_generic_workspace_bytes(const sskd_generic_config* cfg, int B, int S, int training) {
  const size_t whole = workspace_one_part(cfg, B, S, training);
  if (whole == 0 || generic_parts(cfg, B, S, training) == 1) return whole;
  const size_t split = 2 * part_stride(cfg, B, S, training);
  return split > whole ? split : whole;
}

static int prepare(const sskd_generic_config* cfg, const sskd_generic_weights* w, int B, int S, int training,
                   void* d_workspace, size_t workspace_bytes, Dims* d, std::vector<LayerSaved>* layers, Saved* sv) {
  int rc = check(cfg, w, B, S, d);
  if (rc != SSKD_OK) return rc;
  d->training = training != 0;
  SSKD_REQUIRE(w->word_emb && w->pos_emb && w->type_emb && w->emb_ln_g && w->emb_ln_b && (cfg->layers == 0 || w->layers),
               "generic encoder: null weight pointer");
  if (B == 0) return SSKD_OK;
  const size_t need = workspace_one_part(cfg, B, S, training);
  if (!d_workspace || workspace_bytes < need)
    return sskd::fail(SSKD_ERR_WORKSPACE, "generic encoder: workspace %zu B < required %zu B", workspace_bytes, need);
  layers->resize((size_t)(d->L > 0 ? d->L : 1));
  *sv = carve(d_workspace, *d, layers->data(), training != 0);
  return SSKD_OK;
}

static int generic_forward_rows(const sskd_generic_config* cfg, const sskd_generic_weights* w, const int32_t* d_ids,
                                const int32_t* d_mask, int B, int S, int training, int pool, int normalize, void* d_out,
                                void* d_workspace, size_t workspace_bytes, hipStream_t st) {
  Dims d{};
  std::vector<LayerSaved> layers;
  Saved sv{};
  int rc = prepare(cfg, w, B, S, training, d_workspace, workspace_bytes, &d, &layers, &sv);
  if (rc != SSKD_OK || B == 0) return rc;
  const bf16_t* fin = nullptr;
  TRY(forward_all(cfg, w, d, d_ids, d_mask, sv, st, &fin));
  if (pool) return launch_pool_fwd(fin, d_mask, B, S, d.H, normalize, static_cast<float*>(d_out), sv.pooled, st);
  // raw final hidden states, bf16 [B, S, H]
  if (hipMemcpyAsync(d_out, fin, (size_t)d.M * d.H * sizeof(bf16_t), hipMemcpyDeviceToDevice, st) != hipSuccess)
    return sskd::fail(SSKD_ERR_HIP, "generic_forward: copy of the hidden states failed");
  return SSKD_OK;
}

int sskd_generic_forward(const sskd_generic_config* cfg, const sskd_generic_weights* w, const int32_t* d_ids,
                         const int32_t* d_mask, int B, int S, int training, int pool, int normalize, void* d_out,
                         void* d_workspace, size_t workspace_bytes, void* stream) {
  {   // validation of the whole call (and its workspace) before anything is enqueued
    Dims d{};
    int rc = check(cfg, w, B, S, &d);
    if (rc != SSKD_OK || B == 0) return rc;
    const size_t need = sskd_generic_workspace_bytes(cfg, B, S, training);
    if (!d_workspace || workspace_bytes < need)
      return sskd::fail(SSKD_ERR_WORKSPACE, "generic encoder: workspace %zu B < required %zu B", workspace_bytes, need);
  }
  SSKD_REQUIRE(d_ids && d_mask && d_out, "generic_forward: null pointer");
  hipStream_t st = sskd::as_stream(stream);
  const int parts = generic_parts(cfg, B, S, training);
  if (parts == 1)
    return generic_forward_rows(cfg, w, d_ids, d_mask, B, S, training, pool, normalize, d_out, d_workspace, workspace_bytes, st);
  const int Bp = B / 2;
  const size_t stride = part_stride(cfg, B, S, training);
  const size_t out_row = pool ? (size_t)cfg->hidden * sizeof(float) : (size_t)S * cfg->hidden * sizeof(bf16_t);
  return sskd::run_parts_on_streams(2, st, [&](int i, hipStream_t s) {
    return generic_forward_rows(cfg, w, d_ids + (int64_t)i * Bp * S, d_mask + (int64_t)i * Bp * S, Bp, S, training, pool,
                                normalize, static_cast<char*>(d_out) + (size_t)i * Bp * out_row,
                                static_cast<char*>(d_workspace) + i * stride, stride, s);
  });
}

static int generic_backward_rows(const sskd_generic_config* cfg, const sskd_generic_weights* w, const sskd_generic_grads* grads,
                                 const int32_t* d_ids, const int32_t* d_mask, int B, int S, int normalize, const float* d_dout,
                                 void* d_workspace, size_t workspace_bytes, hipStream_t st) {
  Dims d{};
  std::vector<LayerSaved> layers;
  Saved sv{};
  int rc = prepare(cfg, w, B, S, 1, d_workspace, workspace_bytes, &d, &layers, &sv);
  if (rc != SSKD_OK || B == 0) return rc;
  SSKD_REQUIRE(grads && d_ids && d_mask && d_dout, "generic_backward: null pointer");
  SSKD_REQUIRE(grads->word_emb && grads->pos_emb && grads->type_emb && grads->emb_ln_g && grads->emb_ln_b &&
                   (cfg->layers == 0 || grads->layers),
               "generic_backward: null gradient pointer");
  // gradient flowing into the current layer's output = dx + dxb (dxb: the residual share, null at the top)
  const bf16_t* dx = sv.tH2;
  const bf16_t* dxb = nullptr;
  TRY(launch_pool_bwd(d_dout, sv.pooled, d_mask, B, S, d.H, normalize, sv.tH2, st));
  for (int l = d.L - 1; l >= 0; --l) {
    const sskd_generic_layer_weights& lw = w->layers[l];
    SSKD_REQUIRE(lw.wqkv_t && lw.wo_t && lw.w1_t && lw.w2_t, "generic_backward: layer %d lacks transposed weights", l);
    const bf16_t* x_in = l == 0 ? sv.x0 : sv.layer[l - 1].x2;
    TRY(layer_backward(d, lw, grads->layers[l], x_in, d_mask, sv.layer[l], sv, dx, dxb, st));
    // the layer's input gradient is tH1 + tH0: the layer below consumes both in its first kernel (which only READS tH1
    // and rewrites tH0 row by row), so nothing is copied and nothing is added in a pass of its own
    dx = sv.tH1;
    dxb = sv.tH0;
  }
  TRY(launch_ln_bwd(dx, sv.z0, sv.mean0, sv.rstd0, w->emb_ln_g, d.M, d.H, sv.tH0, grads->emb_ln_g, grads->emb_ln_b, st, nullptr, dxb));
  return launch_embed_bwd(d_ids, d_mask, sv.tH0, B, S, d.H, cfg->vocab_size, cfg->pos_offset, grads->word_emb,
                          grads->pos_emb, grads->type_emb, st);
}

int sskd_generic_backward(const sskd_generic_config* cfg, const sskd_generic_weights* w, const sskd_generic_grads* grads,
                          const int32_t* d_ids, const int32_t* d_mask, int B, int S, int normalize, const float* d_dout,
                          void* d_workspace, size_t workspace_bytes, void* stream) {
  {
    Dims d{};
    int rc = check(cfg, w, B, S, &d);
    if (rc != SSKD_OK || B == 0) return rc;
    const size_t need = sskd_generic_workspace_bytes(cfg, B, S, 1);
    if (!d_workspace || workspace_bytes < need)
      return sskd::fail(SSKD_ERR_WORKSPACE, "generic encoder: workspace %zu B < required %zu B", workspace_bytes, need);
  }
  hipStream_t st = sskd::as_stream(stream);
  const int parts = generic_parts(cfg, B, S, 1);
  if (parts == 1)
    return generic_backward_rows(cfg, w, grads, d_ids, d_mask, B, S, normalize, d_dout, d_workspace, workspace_bytes, st);
  // the halves ADD into the same gradient buffers: every accumulation of the backward is atomic (generic_parts)
  const int Bp = B / 2;
  const size_t stride = part_stride(cfg, B, S, 1);
  return sskd::run_parts_on_streams(2, st, [&](int i, hipStream_t s) {
    return generic_backward_rows(cfg, w, grads, d_ids + (int64_t)i * Bp * S, d_mask + (int64_t)i * Bp * S, Bp, S, normalize,
                                 d_dout + (int64_t)i * Bp * cfg->hidden, static_cast<char*>(d_workspace) + i * stride, stride, s);
  });
}

size_t sskd_teacher_workspace_bytes(const sskd_generic_config* cfg, int B, int S) {
  return sskd_generic_workspace_bytes(cfg, B, S, 0);
}

// Cross-encoder score: generic encoder -> hidden state of token 0 (<s>) -> dense + tanh -> out_proj
// (XLMRobertaForSequenceClassification's RobertaClassificationHead with num_labels = 1); the head runs in fp32.
static int teacher_score_rows(const sskd_generic_config* cfg, const sskd_generic_weights* w, const float* d_head_dense_w,
                              const float* d_head_dense_b, const float* d_head_out_w, const float* d_head_out_b,
                              const int32_t* d_ids, const int32_t* d_mask, int B, int S, float* d_logits, void* d_workspace,
                              size_t workspace_bytes, hipStream_t st) {
  Dims d{};
  std::vector<LayerSaved> layers;
  Saved sv{};
  int rc = prepare(cfg, w, B, S, 0, d_workspace, workspace_bytes, &d, &layers, &sv);
  if (rc != SSKD_OK) return rc;
  const bf16_t* fin = nullptr;
  TRY(forward_all(cfg, w, d, d_ids, d_mask, sv, st, &fin));
  hipLaunchKernelGGL(teacher_head_kernel, dim3(B), dim3(HEAD_WAVES * 64), 0, st, fin, S, d.H, d_head_dense_w, d_head_dense_b,
                     d_head_out_w, d_head_out_b, d_logits);
  return sskd::check_launch("teacher_head_kernel");
}

// Batches of >= 2 x 16 384 tokens run as two halves on two streams (generic_parts above: + 1.5 ... 2.7 % at 128 pairs x 256
// tokens, tools/two_stream_teacher_probe.py; a pair's score does not depend on its batch-mates: bit-identical).
int sskd_teacher_score(const sskd_generic_config* cfg, const sskd_generic_weights* w, const float* d_head_dense_w,
                       const float* d_head_dense_b, const float* d_head_out_w, const float* d_head_out_b,
                       const int32_t* d_ids, const int32_t* d_mask, int B, int S, float* d_logits, void* d_workspace,
                       size_t workspace_bytes, void* stream) {
  {   // validation of the whole call (and its workspace) before anything is enqueued
    Dims d{};
    int rc = check(cfg, w, B, S, &d);
    if (rc != SSKD_OK || B == 0) return rc;
    const size_t need = sskd_teacher_workspace_bytes(cfg, B, S);
    if (!d_workspace || workspace_bytes < need)
      return sskd::fail(SSKD_ERR_WORKSPACE, "teacher_score: workspace %zu B < required %zu B", workspace_bytes, need);
  }
  SSKD_REQUIRE(d_head_dense_w && d_head_dense_b && d_head_out_w && d_head_out_b && d_ids && d_mask && d_logits,
               "teacher_score: null pointer");
  hipStream_t st = sskd::as_stream(stream);
  const int parts = generic_parts(cfg, B, S, 0);
  const int Bp = B / 2;
  const size_t part_bytes = parts == 2 ? part_stride(cfg, B, S, 0) : 0;
  if (parts == 1)
    return teacher_score_rows(cfg, w, d_head_dense_w, d_head_dense_b, d_head_out_w, d_head_out_b, d_ids, d_mask, B, S,
                              d_logits, d_workspace, workspace_bytes, st);
  return sskd::run_parts_on_streams(2, st, [&](int i, hipStream_t s) {
    return teacher_score_rows(cfg, w, d_head_dense_w, d_head_dense_b, d_head_out_w, d_head_out_b,
                              d_ids + (int64_t)i * Bp * S, d_mask + (int64_t)i * Bp * S, Bp, S, d_logits + (int64_t)i * Bp,
                              static_cast<char*>(d_workspace) + i * part_bytes, part_bytes, s);
  });
}

// test hook: the NT GEMM by itself
int sskd_gemm_backend(int mode) {
  if (mode == 0 || mode == 1) sskd_generic::set_gemm_backend(mode);
  return sskd_generic::gemm_backend();
}

int sskd_gemm_nt_bf16(const void* d_a, const void* d_b, void* d_c, const float* d_bias, int M, int N, int K, int c_is_f32,
                      int accumulate, void* stream) {
  return gemm(static_cast<const bf16_t*>(d_a), K, static_cast<const bf16_t*>(d_b), K, d_c, N, M, N, K, d_bias,
              c_is_f32 != 0, accumulate != 0, sskd::as_stream(stream));
}

int sskd_gemm_tn_bf16(const void* d_a, const void* d_b, float* d_c, int64_t T, int M, int N, void* stream) {
  if (!gemm_tn_supported(T, M, N, M, N))
    return sskd::fail(SSKD_ERR_UNSUPPORTED, "gemm_tn: T=%lld M=%d N=%d not served (M %% 384, N %% 128, T %% 64)", (long long)T, M, N);
  return launch_gemm_tn(static_cast<const bf16_t*>(d_a), M, static_cast<const bf16_t*>(d_b), N, d_c, N, T, M, N,
                        sskd::as_stream(stream));
}

}  // extern "C"
