// Plain library GEMMs for the large-K products of the generic path (host code only, no kernel of ours in this file).
//
// The teacher cross-encoder (reference: src/mining/miners.py:128-151, CrossEncoder.predict on XLM-R-large) spends 80 %
// of a layer in four PLAIN products: C[M, N] = A[M, K] . W[N, K]^T + bias[N] with K, N in {1024, 3072, 4096}.  Three of
// them have no fused epilogue beyond the bias (QKV, attention output, FFN2), and on those shapes the vendor library's
// tuned kernels beat this repo's 256 x 256 MFMA kernel by 15-25 % (tools/gemm_probe.py, round 4, same box: 1 207-1 481
// against 1 022-1 181 TFLOP/s).  They go through hipBLASLt; everything with a fused epilogue (FFN1 + erf-GELU - the
// library's GELU is the tanh form -, attention, LayerNorm), every K = 384 product of the student (where the 256-tile
// kernel is the faster one: 819 / 641 / 839 against 570 / 488 / 773 TFLOP/s) and every batched / fp32 / accumulating
// product stay on the hand-written kernels (generic.hip).
//
// Row-major NT in the library's column-major terms: C^T[N, M] = op_T(W as K x N, ld = ldb) . (A as K x M, ld = lda),
// bias along the rows of C^T.  One plan (descriptor, layouts, heuristic's first algorithm) per shape, cached; no
// workspace is requested (algorithms that need one are not offered), nothing is allocated or synchronised per call.
#include "generic.h"

#include <hipblaslt/hipblaslt.h>

#include <map>
#include <mutex>
#include <tuple>

namespace sskd_generic {

namespace {
struct Plan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t lw = nullptr, la = nullptr, lc = nullptr;
  hipblasLtMatmulAlgo_t algo{};
  bool ok = false;
};
using Key = std::tuple<int, int, int, int, int64_t, int64_t, int64_t, bool>;   // device, M, N, K, lda, ldb, ldc, bias

std::mutex g_mu;
std::map<int, hipblasLtHandle_t> g_handles;
std::map<Key, Plan> g_plans;
int g_backend = 0;   // 0 = automatic (library for the plain large products), 1 = hand-written kernels only

bool lt_ok(hipblasStatus_t s) { return s == HIPBLAS_STATUS_SUCCESS; }

Plan* plan_for(int dev, hipblasLtHandle_t h, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, bool bias) {
  const Key key{dev, M, N, K, lda, ldb, ldc, bias};
  auto it = g_plans.find(key);
  if (it != g_plans.end()) return &it->second;
  Plan p;
  const hipblasOperation_t op_t = HIPBLAS_OP_T, op_n = HIPBLAS_OP_N;
  const hipblasLtEpilogue_t epi = bias ? HIPBLASLT_EPILOGUE_BIAS : HIPBLASLT_EPILOGUE_DEFAULT;
  const int32_t bias_type = HIP_R_32F;
  hipblasLtMatmulPreference_t pref = nullptr;
  const uint64_t no_ws = 0;
  hipblasLtMatmulHeuristicResult_t found{};
  int n_found = 0;
  bool ok = lt_ok(hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F)) &&
            lt_ok(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &op_t, sizeof(op_t))) &&
            lt_ok(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &op_n, sizeof(op_n))) &&
            lt_ok(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi))) &&
            (!bias || lt_ok(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bias_type,
                                                            sizeof(bias_type)))) &&
            lt_ok(hipblasLtMatrixLayoutCreate(&p.lw, HIP_R_16BF, (uint64_t)K, (uint64_t)N, ldb)) &&
            lt_ok(hipblasLtMatrixLayoutCreate(&p.la, HIP_R_16BF, (uint64_t)K, (uint64_t)M, lda)) &&
            lt_ok(hipblasLtMatrixLayoutCreate(&p.lc, HIP_R_16BF, (uint64_t)N, (uint64_t)M, ldc)) &&
            lt_ok(hipblasLtMatmulPreferenceCreate(&pref)) &&
            lt_ok(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &no_ws, sizeof(no_ws))) &&
            lt_ok(hipblasLtMatmulAlgoGetHeuristic(h, p.desc, p.lw, p.la, p.lc, p.lc, pref, 1, &found, &n_found)) &&
            n_found > 0;
  if (pref) (void)hipblasLtMatmulPreferenceDestroy(pref);
  if (ok) p.algo = found.algo;
  p.ok = ok;
  return &(g_plans[key] = p);   // a shape the library does not serve is remembered too (the caller's own kernel runs)
}
}  // namespace

void set_gemm_backend(int mode) {
  std::lock_guard<std::mutex> lock(g_mu);
  g_backend = mode;
}

int gemm_backend() { return g_backend; }

// true when the product went through the library; false = not a shape for it (the caller launches its own kernel)
bool blaslt_gemm_nt(const GemmArgs& a, hipStream_t st, int* rc) {
  *rc = SSKD_OK;
  if (g_backend == 1) return false;
  const bool plain = a.batch1 * a.batch2 == 1 && a.split_k <= 1 && !a.c_is_f32 && !a.accumulate && a.act == 0 &&
                     a.alpha == 1.0f && a.K >= 1024 && a.N >= 1024 && a.M >= 1024 && a.K % 64 == 0 && a.N % 64 == 0;
  if (!plain) return false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> lock(g_mu);
  hipblasLtHandle_t h = nullptr;
  auto hit = g_handles.find(dev);
  if (hit == g_handles.end()) {
    if (!lt_ok(hipblasLtCreate(&h))) h = nullptr;
    g_handles[dev] = h;
  } else {
    h = hit->second;
  }
  if (!h) return false;
  Plan* p = plan_for(dev, h, a.M, a.N, a.K, a.lda, a.ldb, a.ldc, a.bias != nullptr);
  if (!p->ok) return false;
  if (a.bias &&
      !lt_ok(hipblasLtMatmulDescSetAttribute(p->desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &a.bias, sizeof(a.bias)))) {
    *rc = sskd::fail(SSKD_ERR_HIP, "hipblasLt: cannot set the bias pointer");
    return true;
  }
  const float one = 1.0f, zero = 0.0f;
  const hipblasStatus_t s = hipblasLtMatmul(h, p->desc, &one, a.B, p->lw, a.A, p->la, &zero, a.C, p->lc, a.C, p->lc, &p->algo,
                                            nullptr, 0, st);
  if (!lt_ok(s)) *rc = sskd::fail(SSKD_ERR_HIP, "hipblasLtMatmul failed (status %d) on %d x %d x %d", (int)s, a.M, a.N, a.K);
  return true;
}

}  // namespace sskd_generic
