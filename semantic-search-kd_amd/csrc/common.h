// Shared host-side helpers for the C-ABI translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "sskd_amd.h"

namespace sskd {

// Thread-local message behind sskd_last_error().
char* last_error_buf();
int fail(int code, const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Checks the launch that was just enqueued (no synchronisation).
inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(SSKD_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  return SSKD_OK;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// compute units of the current device (256 on MI355X)
int cu_count();

// Side streams of the multi-branch forwards, per device, created on first use (process lifetime): index 0 .. 2.
constexpr int MAX_STREAM_PARTS = 4;
hipStream_t side_stream(int i);
// Parts a batch is cut into by default (SSKD_FORWARD_STREAMS = 1 .. 4 for A/B runs; default 2).
int forward_stream_parts();

// Runs part(i, stream) for i = 0 .. parts - 1: part 0 on the caller's stream `st`, the others on side streams forked from
// `st` and joined back into it with events (under stream capture the side streams join the capture and the graph gets
// branches).  The joins are enqueued whatever a part returns, so a capturing caller always gets every branch back.
// Returns the first error.  Why: every kernel of a part fills the chip by itself; what the branches buy is that the parts
// drift apart, so that the un-overlapped memory phases of one (one workgroup per CU, every CU in the same phase) meet
// the other's compute - measured + 4.8 % on the student encoder, + 2.7 % on the teacher.
template <class F>
int run_parts_on_streams(int parts, hipStream_t st, F&& part) {
  if (parts > MAX_STREAM_PARTS) parts = MAX_STREAM_PARTS;
  hipStream_t side[MAX_STREAM_PARTS - 1] = {};
  for (int i = 0; i + 1 < parts; ++i)
    if (!(side[i] = side_stream(i))) return fail(SSKD_ERR_HIP, "cannot create a side stream");
  if (parts <= 1) return part(0, st);
  hipEvent_t fork = nullptr, join[MAX_STREAM_PARTS - 1] = {};
  bool ok = hipEventCreateWithFlags(&fork, hipEventDisableTiming) == hipSuccess;
  for (int i = 0; ok && i + 1 < parts; ++i) ok = hipEventCreateWithFlags(&join[i], hipEventDisableTiming) == hipSuccess;
  int rc = ok && hipEventRecord(fork, st) == hipSuccess ? SSKD_OK : fail(SSKD_ERR_HIP, "cannot fork the side streams");
  int forked = 0;   // side streams that wait on the caller's stream (and, under capture, belong to its capture)
  for (int i = 1; rc == SSKD_OK && i < parts; ++i) {
    if (hipStreamWaitEvent(side[i - 1], fork, 0) != hipSuccess) {
      rc = fail(SSKD_ERR_HIP, "cannot fork side stream %d", i);
      break;
    }
    forked = i;
    rc = part(i, side[i - 1]);
  }
  const int rc0 = rc == SSKD_OK ? part(0, st) : rc;
  for (int i = 1; i <= forked; ++i)
    if (!(hipEventRecord(join[i - 1], side[i - 1]) == hipSuccess && hipStreamWaitEvent(st, join[i - 1], 0) == hipSuccess) &&
        rc == SSKD_OK)
      rc = fail(SSKD_ERR_HIP, "cannot join side stream %d", i);
  if (fork) (void)hipEventDestroy(fork);   // destruction is deferred until the recorded work has completed
  for (hipEvent_t e : join)
    if (e) (void)hipEventDestroy(e);
  return rc != SSKD_OK ? rc : rc0;
}

// erf-GELU (HF "gelu") = x Phi(x), evaluated as x * sigmoid(x (a + b x^2 + c x^4)) with (a, b, c)
// fitted (minimax on [-8, 8]) to the exact erf form: max |error| 2.6e-5 - the same bound as the
// Abramowitz-Stegun 7.1.25 erf it replaces, 80x below the bf16 half-ulp of the value produced -
// in 7 VALU + 2 transcendental instructions instead of 10 + 2 (the MLP is bound by the SIMD's
// instruction issue, not by the matrix pipe: every VALU instruction per element is 16 x 4 issue
// cycles per chunk).  The polynomial is evaluated on clamp(x, -8, 8) (beyond, sigmoid is 0 or 1 to
// fp32 precision and the quartic term would turn over at |x| > 11); coefficients carry the
// -log2(e) of exp(-t) = exp2(-t log2 e).
__device__ inline float gelu_erf(float x) {
  constexpr float A = -2.3011212f, B = -0.10677574f, C = 0.0010142655f;
  const float xc = __builtin_amdgcn_fmed3f(x, -8.0f, 8.0f);
  const float x2 = xc * xc;
  float q = fmaf(C, x2, B);
  q = fmaf(q, x2, A);
  const float e = __builtin_amdgcn_exp2f(q * xc);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

}  // namespace sskd

#define SSKD_REQUIRE(cond, ...) \
  do {                          \
    if (!(cond)) return sskd::fail(SSKD_ERR_INVALID, __VA_ARGS__); \
  } while (0)
