// Error string, version and device info for libfaceoff_hip.so.
#include <stdarg.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";

void fo_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fo_cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cus = n;
    else
      cus = 256;
  }
  return cus;
}

extern "C" {
int fo_version(void) { return 100; }
const char* fo_last_error(void) { return g_err; }
int fo_device_info(int32_t* out3) {
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
    fo_set_error("hipGetDeviceProperties failed");
    return FO_E_HIP;
  }
  out3[0] = p.multiProcessorCount;
  out3[1] = p.clockRate;
  out3[2] = strncmp(p.gcnArchName, "gfx950", 6) == 0;
  return FO_OK;
}
}
