// thread-local last-error string for the C ABI (cfgpp_last_error).
#include <cstdarg>
#include <cstdio>
#include "common.h"
static thread_local char g_err[1024] = "";
void cfgpp_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* cfgpp_last_error(void) { return g_err; }
