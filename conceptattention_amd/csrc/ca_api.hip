// Host-side plumbing of libconceptattn: version, thread-local error text, device check.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "ca_common.h"

static thread_local char g_err[512] = "";

void ca_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// CU count of the current device (cached per device; a race only repeats the query); -1 if the query fails
int ca_cu_count() {
  static std::atomic<int> cus[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  int n = cus[dev & 63].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = -1;
    cus[dev & 63].store(n, std::memory_order_relaxed);
  }
  return n;
}

extern "C" int ca_version(void) { return CA_VERSION; }

extern "C" const char *ca_last_error(void) { return g_err; }

extern "C" int ca_check_device(void) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) {
    ca_set_error("ca_check_device: hipGetDevice: %s", hipGetErrorString(e));
    return CA_ERR_ARCH;
  }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) {
    ca_set_error("ca_check_device: hipGetDeviceProperties: %s", hipGetErrorString(e));
    return CA_ERR_ARCH;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    ca_set_error("ca_check_device: device %d is %s; libconceptattn is built for gfx950 only", dev, prop.gcnArchName);
    return CA_ERR_ARCH;
  }
  return CA_OK;
}
