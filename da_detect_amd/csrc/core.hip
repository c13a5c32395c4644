// Error reporting + device probe for libdadet_hip.so.
#include "common.h"
#include <string.h>

namespace dadet {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace dadet

extern "C" const char* dadet_last_error(void) { return dadet::g_err; }

extern "C" int dadet_version(void) { return 100; }

extern "C" int dadet_device_info(int* cu_count, int* clock_khz, size_t* hbm_bytes, char* arch_name,
                                 int arch_name_len) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) {
    dadet::set_error("hipGetDevice: %s", hipGetErrorString(e));
    return DADET_ELAUNCH;
  }
  hipDeviceProp_t p;
  e = hipGetDeviceProperties(&p, dev);
  if (e != hipSuccess) {
    dadet::set_error("hipGetDeviceProperties: %s", hipGetErrorString(e));
    return DADET_ELAUNCH;
  }
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (clock_khz) *clock_khz = p.clockRate;
  if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
  if (arch_name && arch_name_len > 0) {
    strncpy(arch_name, p.gcnArchName, arch_name_len - 1);
    arch_name[arch_name_len - 1] = 0;
  }
  return DADET_OK;
}
