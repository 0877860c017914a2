// Library-level entry points: version, last-error string.
#include <stdarg.h>
#include <string.h>

#include "common.h"

static thread_local char g_last_error[512] = "";

void cradle_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}

extern "C" {

const char* cradle_last_error(void) { return g_last_error; }

int cradle_abi_version(void) { return 1; }

// Number of visible HIP devices, or -1 if the runtime cannot be initialised.  Used by the
// Python loader to fail loudly instead of silently computing elsewhere.
int cradle_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return -1;
  return n;
}

}  // extern "C"
