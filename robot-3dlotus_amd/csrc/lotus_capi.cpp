// lotus-hip: C-ABI plumbing shared by every op family (error string, version).
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void lotus_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {
const char* lotus_last_error(void) { return g_err; }
int lotus_abi_version(void) { return 1; }
}
