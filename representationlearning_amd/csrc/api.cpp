// Host-side glue shared by every entry point of librssf: version / arch / thread-local error string.
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/rssf.h"

namespace rssf {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return RSSF_ERR_LAUNCH;
  }
  return RSSF_OK;
}
}  // namespace rssf

extern "C" {
const char* rssf_version(void) { return "0.1.0"; }
const char* rssf_arch(void) { return "gfx950"; }
const char* rssf_last_error(void) { return rssf::g_err; }
}
