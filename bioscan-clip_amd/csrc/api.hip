// Host-side support of the C ABI: per-thread error string, ABI version.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

static thread_local char g_err[512] = "";

void bsclip_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* bsclip_last_error(void) { return g_err; }
extern "C" int bsclip_abi_version(void) { return 3; }  // 3: bsclip_epi_args.struct_size (leading), bsclip_epi_args_size; diag builds moved out
