// Error slot + version for libisd_hip.so.
#include "common.h"
#include <string.h>

namespace isd {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace isd

extern "C" int isd_abi_version(void) { return ISD_ABI_VERSION; }
extern "C" const char* isd_last_error(void) { return isd::g_err; }
extern "C" int isd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
