// core.hip -- error reporting, ABI version and the gfx950 device check of libovla_hip.
#include "common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void ovla_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ovla_last_error(void) { return g_err; }
extern "C" int ovla_abi_version(void) { return OVLA_ABI_VERSION; }
#ifndef OVLA_SRC_HASH
#define OVLA_SRC_HASH "unhashed"
#endif
extern "C" const char* ovla_build_hash(void) { return OVLA_SRC_HASH; }

extern "C" int ovla_check_device(int device) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    ovla_set_error("ovla_check_device: hipGetDeviceProperties(%d): %s", device, hipGetErrorString(e));
    return OVLA_ELAUNCH;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    ovla_set_error("ovla_check_device: device %d is %s, this library is built for gfx950 only", device, prop.gcnArchName);
    return OVLA_EARCH;
  }
  return OVLA_OK;
}
