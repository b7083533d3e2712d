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

}  // namespace sskd

#define SSKD_REQUIRE(cond, ...) \
  do {                          \
    if (!(cond)) return sskd::fail(SSKD_ERR_INVALID, __VA_ARGS__); \
  } while (0)
