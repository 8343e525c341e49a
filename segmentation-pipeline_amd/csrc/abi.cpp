// Version / error reporting for libm355seg.
#include "common.hpp"

namespace m355 {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int num_cus() {
  static int cache[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    (void)hipGetLastError();  // no device (build host): not an error for a host-side query
    return 256;
  }
  if (cache[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
      (void)hipGetLastError();
      n = 256;
    }
    cache[dev] = n;
  }
  return cache[dev];
}
}  // namespace m355

extern "C" int m355_version(void) { return M355_ABI_VERSION; }
extern "C" const char* m355_last_error(void) { return m355::g_err; }
