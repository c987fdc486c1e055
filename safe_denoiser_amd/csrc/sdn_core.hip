// Library identity + device probe.
#include "sdn_common.h"
#include <string.h>

extern "C" {

int sdn_abi_version(void) { return 3; }

const char* sdn_device_arch_host(void) {
  static char name[64];
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return nullptr;
  strncpy(name, prop.gcnArchName, sizeof(name) - 1);
  name[sizeof(name) - 1] = 0;
  char* colon = strchr(name, ':');
  if (colon) *colon = 0;
  return name;
}

}  // extern "C"
