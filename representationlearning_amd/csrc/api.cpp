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
// The code objects in this library exist for gfx950 only.  With a device present the answer is that device's own
// gcnArchName (so a caller on anything else sees the mismatch instead of a launch failure later); without one (build /
// symbol checks on a CPU host) it is the architecture the library was compiled for.
const char* rssf_arch(void) {
  static thread_local char arch[64];
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.gcnArchName[0]) {
    size_t n = 0;
    while (prop.gcnArchName[n] && prop.gcnArchName[n] != ':' && n + 1 < sizeof(arch)) { arch[n] = prop.gcnArchName[n]; ++n; }
    arch[n] = 0;
    return arch;
  }
  (void)hipGetLastError();
  return "gfx950";
}
const char* rssf_last_error(void) { return rssf::g_err; }
}
