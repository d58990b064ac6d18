// Library plumbing: version, arch string, thread-local error text.
#include <stdarg.h>

#include "ubv_common.h"

namespace ubv {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace ubv

extern "C" int ubv_version(void) { return 100; }   // 0.1.0
extern "C" const char* ubv_last_error(void) { return ubv::g_err; }
extern "C" const char* ubv_arch(void) { return "gfx950"; }
