// api.hip — library-level plumbing of libneosr_amd.so: error string, build info, ABI version.
#include "common.h"
#include "../../include/neosr_amd.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void neosr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* neosr_last_error(void) { return g_err; }

extern "C" const char* neosr_build_info(void) {
  return "libneosr_amd gfx950 (CDNA4) fp32-MFMA " __DATE__ " " __TIME__;
}

extern "C" int neosr_abi_version(void) { return 1; }
