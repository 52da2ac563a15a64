// Error text + version for the C ABI (include/movae.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/movae.h"

static thread_local char g_err[512] = "";

void movae_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

int movae_version(void) { return 100; }  // 0.1.0

const char* movae_last_error(void) { return g_err; }

}
