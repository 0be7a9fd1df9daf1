#include <stdarg.h>
#include "tp_common.h"

namespace tp {
static thread_local char g_err[512] = "ok";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace tp

extern "C" int tp_abi_version(void) { return TP_ABI_VERSION; }
extern "C" const char* tp_last_error(void) { return tp::g_err; }
