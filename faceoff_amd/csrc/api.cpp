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

std::atomic<int> faceoff_notes_on{0};
static thread_local char g_kernel[160] = "";

void fo_note_kernel(const char* base, const char* pretty) {
  if (!pretty) {
    snprintf(g_kernel, sizeof g_kernel, "%s", base);
    return;
  }
  // pretty = "const char *fo_tname() [T = fo_vals<256, 256, 2, 4>]": keep what stands between "fo_vals" and the closing bracket
  const char* a = strstr(pretty, "fo_vals<");
  const char* z = a ? strrchr(a, ']') : nullptr;
  if (!a || !z) {
    snprintf(g_kernel, sizeof g_kernel, "%s<?>", base);
    return;
  }
  snprintf(g_kernel, sizeof g_kernel, "%s%.*s", base, (int)(z - (a + 7)), a + 7);
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
int fo_version(void) { return FO_ABI_VERSION; }
const char* fo_last_error(void) { return g_err; }
int fo_kernel_notes(int enable) { return faceoff_notes_on.exchange(enable ? 1 : 0); }
const char* fo_last_kernel(void) {
  static thread_local char out[160];
  memcpy(out, g_kernel, sizeof out);
  g_kernel[0] = 0;
  return out;
}
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
