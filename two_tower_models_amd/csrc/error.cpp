// Thread-local last-error string + ABI version (host-only translation unit).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/tt_hotpath.h"

namespace tt {
static thread_local char g_err[512] = "no error";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace tt

extern "C" int tt_abi_version(void) { return TT_ABI_VERSION; }
extern "C" const char* tt_last_error_string(void) { return tt::g_err; }
