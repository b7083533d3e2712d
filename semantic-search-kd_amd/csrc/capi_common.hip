// Error plumbing and library-level queries of the C-ABI (include/sskd_amd.h).
#include "common.h"

#include <cstring>

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
