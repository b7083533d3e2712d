// Error plumbing and library-level queries of the C-ABI (include/sskd_amd.h).
#include "common.h"

#include <cstdlib>
#include <cstring>
#include <mutex>

namespace sskd {

char* last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

int cu_count() {
  static int cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    cached[dev] = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0 ? n : 256;
  }
  return cached[dev];
}

hipStream_t side_stream(int i) {
  static std::mutex mu;
  static hipStream_t streams[64][MAX_STREAM_PARTS - 1] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || i < 0 || i >= MAX_STREAM_PARTS - 1) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!streams[dev][i] && hipStreamCreateWithFlags(&streams[dev][i], hipStreamNonBlocking) != hipSuccess)
    streams[dev][i] = nullptr;
  return streams[dev][i];
}

int forward_stream_parts() {
  static const int n = [] {
    const char* e = std::getenv("SSKD_FORWARD_STREAMS");
    const int v = e ? std::atoi(e) : 2;
    return v < 1 ? 1 : (v > MAX_STREAM_PARTS ? MAX_STREAM_PARTS : v);
  }();
  return n;
}

}  // namespace sskd

extern "C" {

int sskd_abi_version(void) { return SSKD_ABI_VERSION; }

const char* sskd_last_error(void) { return sskd::last_error_buf(); }

int sskd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return n;
}

}  // extern "C"
