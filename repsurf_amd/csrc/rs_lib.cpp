// rs_lib.cpp — library-level entry points of librepsurf_hip (error text, ABI version, device info).
#include "rs_common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void rs_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *rs_last_error(void) { return g_err; }
extern "C" int rs_abi_version(void) { return 37; }

// bytes of LDS one workgroup may ask for on the current device (cached per process: this library targets one device model)
int rs_lds_limit(void) {
  static int cached = 0;
  if (cached > 0) return cached;
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 65536;      // (not cached: ask again)
  cached = (int)p.sharedMemPerBlock;
  return cached;
}

extern "C" int rs_device_info(int *cu_count, int *wave_size, int *lds_bytes, char *arch, int arch_len) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) { rs_set_error("rs_device_info: %s", hipGetErrorString(e)); return RS_ERR_HIP_BASE + (int)e; }
  hipDeviceProp_t p;
  e = hipGetDeviceProperties(&p, dev);
  if (e != hipSuccess) { rs_set_error("rs_device_info: %s", hipGetErrorString(e)); return RS_ERR_HIP_BASE + (int)e; }
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (wave_size) *wave_size = p.warpSize;
  if (lds_bytes) *lds_bytes = (int)p.sharedMemPerBlock;
  if (arch && arch_len > 0) { strncpy(arch, p.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
  return RS_OK;
}
