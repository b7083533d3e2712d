// Dimension-generic bf16 building blocks (row-major activations) shared by the student
// forward/backward step (train.hip) and the teacher cross-encoder (teacher.hip).
// The inference encoder (encoder.hip) is specialised for hidden 384 and keeps its own
// fragment-order layouts; these kernels trade some of that speed for arbitrary shapes.
#pragma once
#include "common.h"

namespace sskd_generic {

typedef __bf16 bf16_t;

// C[M, N] (+)= alpha * A[M, K] . B[N, K]^T + bias[N]; both operands have the reduction dimension
// contiguous ("NT").  Two-level batching: z = b1 * batch2 + b2 with element strides per level.
struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const float* bias;   // [N] or nullptr
  int M, N, K;         // K % 32 == 0
  int64_t lda, ldb, ldc;  // elements; lda, ldb multiples of 8
  int batch1, batch2;
  int64_t sA1, sA2, sB1, sB2, sC1, sC2;
  float alpha;
  int c_is_f32;        // 0: bf16 output, 1: fp32 output
  int accumulate;      // fp32 output only: C += result
  int split_k;         // > 1 (unbatched, fp32 accumulate only): K is cut into split_k slices, one workgroup
                       // layer each, partial products meet through fp32 atomics (order not reproducible)
  int act;             // bf16 output only: 0 = none, 1 = erf-GELU applied to (alpha * acc + bias) in the epilogue
};

int launch_gemm_nt(const GemmArgs& a, hipStream_t st);
// blaslt.hip: plain large-K products (K, N >= 1024, bf16 out, bias only) go through hipBLASLt; returns false when the
// product is not one of those (the caller launches the hand-written kernel).  set_gemm_backend(1) keeps everything on
// the hand-written kernels (tests, A/B probes); 0 = automatic.
bool blaslt_gemm_nt(const GemmArgs& a, hipStream_t st, int* rc);
void set_gemm_backend(int mode);
int gemm_backend();

// Weight-gradient product with NO operand transposes: C[M, N] (fp32) += A[T, M]^T B[T, N], the reduction (token)
// dimension being the ROW of both row-major operands.  T is cut over workgroups, partial tiles meet through fp32
// atomics.  Served when M % 384 == 0, N % 128 == 0, T % 64 == 0 (gemm_tn_supported): the student's shapes.
bool gemm_tn_supported(int64_t T, int M, int N, int64_t lda, int64_t ldb);
int launch_gemm_tn(const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, float* C, int64_t ldc, int64_t T, int M,
                   int N, hipStream_t st);

// out[C, R] = in[R, C]^T, batched like the GEMM (element strides).
struct TransposeArgs {
  const bf16_t* in;
  bf16_t* out;
  int R, C;
  int64_t ld_in, ld_out;
  int batch1, batch2;
  int64_t sI1, sI2, sO1, sO2;
  float* colsum;   // optional (unbatched): colsum[c] += sum_r in[r][c] (fp32 atomics) - the bias gradient of the
                   // product whose operand is being transposed, without a second pass over it
};
int launch_transpose(const TransposeArgs& a, hipStream_t st);

// z = a (+ b); y = LayerNorm(z) * gamma + beta.  Optionally saves z (bf16), mean and rstd (fp32 [M]).
int launch_add_ln_fwd(const bf16_t* a, const bf16_t* b, const float* gamma, const float* beta, float eps,
                      int64_t M, int H, bf16_t* y, bf16_t* z_save, float* mean, float* rstd, hipStream_t st);
// dz = LN backward of dy; dgamma / dbeta (fp32 [H]) are ACCUMULATED (atomics); dz_colsum (optional, fp32 [H]) +=
// column sums of dz (the bias gradient of the product whose output was normalised).
int launch_ln_bwd(const bf16_t* dy, const bf16_t* z, const float* mean, const float* rstd, const float* gamma,
                  int64_t M, int H, bf16_t* dz, float* dgamma, float* dbeta, hipStream_t st, float* dz_colsum = nullptr,
                  const bf16_t* dy2 = nullptr);   // incoming gradient = dy + dy2 (residual branch), dz may alias either

// Inference attention, fused (no score matrix in memory): for every (batch row, head)
//   ctx[b, :, h*DH : (h+1)*DH] = softmax(scale * Q K^T + (key masked ? -inf : 0)) V
// with Q / K / V the column blocks [0, H) / [H, 2H) / [2H, 3H) of the row-major qkv [B*S, 3H].
// DH in {32, 64, 128}; S a multiple of 32, at most 512.
// `lse` (optional, fp32 [B, heads, S]): log2-domain log-sum-exp of every query's scaled scores - all the
// backward kernel needs to rebuild the probabilities.
int launch_attention_fwd(const bf16_t* qkv, const int32_t* key_mask, int B, int S, int heads, int DH, float scale,
                         bf16_t* ctx, float* lse, hipStream_t st);
// Fused attention backward (probabilities recomputed from qkv + lse, no S x S matrix in memory):
//   dqkv[:, 0:H | H:2H | 2H:3H] = dQ | dK | dV  given dctx, with ctx the forward output.
// One workgroup per (batch row, head) keeps the head's Q, K, V, dO (and three transposed copies) in LDS:
// served for S * DH <= 8192 (attention_bwd_supported), e.g. the student (DH 32, S <= 256).
bool attention_bwd_supported(int S, int DH);
int launch_attention_bwd(const bf16_t* qkv, const int32_t* key_mask, const bf16_t* ctx, const bf16_t* dctx,
                         const float* lse, int B, int S, int heads, int DH, float scale, bf16_t* dqkv, hipStream_t st);

// rows of S scores (bf16, in place): P = softmax(scale * s + (key masked ? -inf : 0)).
int launch_softmax_fwd(bf16_t* scores, const int32_t* key_mask, int B, int heads, int S, float scale, hipStream_t st);
// in place on dP: dS = scale * P * (dP - sum_j dP_j P_j)
int launch_softmax_bwd(bf16_t* dP, const bf16_t* P, int64_t rows, int S, float scale, hipStream_t st);

int launch_gelu_fwd(const bf16_t* u, bf16_t* h, int64_t n, hipStream_t st);
int launch_gelu_bwd(const bf16_t* u, const bf16_t* dh, bf16_t* du, int64_t n, hipStream_t st);
// the same over a [M, F] matrix, plus db[c] += column sums of du (the bias gradient of the product that made u)
int launch_gelu_bwd_colsum(const bf16_t* u, const bf16_t* dh, bf16_t* du, float* db, int64_t M, int F, hipStream_t st);

// db[N] += sum_m dY[m, N]  (fp32, accumulated with atomics)
int launch_colsum(const bf16_t* dY, int64_t M, int N, int64_t ld, float* db, hipStream_t st);
// c = a + b (bf16)
int launch_add(const bf16_t* a, const bf16_t* b, bf16_t* c, int64_t n, hipStream_t st);

// embeddings: z[m] = word[id] + pos[pos_offset + t] + type[0]  (bf16 out; padding rows -> 0 when zero_pad)
int launch_embed_fwd(const int32_t* ids, const int32_t* mask, const bf16_t* word, const bf16_t* pos, const bf16_t* type0,
                     int B, int S, int H, int vocab, int pos_offset, bf16_t* z, hipStream_t st);
int launch_embed_bwd(const int32_t* ids, const int32_t* mask, const bf16_t* dz, int B, int S, int H, int vocab,
                     int pos_offset, float* dword, float* dpos, float* dtype0, hipStream_t st);

// masked mean pool + optional L2 normalise of row-major bf16 hidden [B, S, H]; saves pooled (pre-normalise) rows
int launch_pool_fwd(const bf16_t* hidden, const int32_t* mask, int B, int S, int H, int normalize, float* out,
                    float* pooled_save, hipStream_t st);
int launch_pool_bwd(const float* dout, const float* pooled, const int32_t* mask, int B, int S, int H, int normalize,
                    bf16_t* dhidden, hipStream_t st);

}  // namespace sskd_generic
