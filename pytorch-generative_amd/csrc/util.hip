// util.hip — error reporting + ABI version for libpg_hip.so.
#include <stdarg.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void pg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

PG_EXPORT int pg_abi_version(void) { return PG_ABI_VERSION; }
PG_EXPORT const char* pg_last_error(void) { return g_err; }
