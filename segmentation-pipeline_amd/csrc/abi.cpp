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
}  // namespace m355

extern "C" int m355_version(void) { return M355_ABI_VERSION; }
extern "C" const char* m355_last_error(void) { return m355::g_err; }
