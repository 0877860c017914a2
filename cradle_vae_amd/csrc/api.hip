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

// HIP-event helpers for callers without a HIP binding of their own (bench.py times single
// kernels of the fused step with them; see cradle_step_desc.ev_k1_begin/end).
int cradle_event_create(void** ev) {
  hipEvent_t e;
  if (!ev || hipEventCreate(&e) != hipSuccess) {
    cradle_set_error("event_create failed");
    return CRADLE_ERR_HIP;
  }
  *ev = (void*)e;
  return CRADLE_OK;
}
int cradle_event_destroy(void* ev) { return hipEventDestroy((hipEvent_t)ev) == hipSuccess ? CRADLE_OK : CRADLE_ERR_HIP; }
// elapsed milliseconds between two recorded events; both must have completed
int cradle_event_elapsed_ms(void* begin, void* end, float* ms) {
  hipError_t e = hipEventElapsedTime(ms, (hipEvent_t)begin, (hipEvent_t)end);
  if (e != hipSuccess) {
    cradle_set_error("event_elapsed_ms: %s", hipGetErrorString(e));
    return CRADLE_ERR_HIP;
  }
  return CRADLE_OK;
}

// makes everything enqueued on `stream` from now on wait for `ev` (no-op if it was never recorded)
int cradle_stream_wait_event(hipStream_t stream, void* ev) {
  if (!ev || hipStreamWaitEvent(stream, (hipEvent_t)ev, 0) != hipSuccess) {
    cradle_set_error("stream_wait_event failed");
    return CRADLE_ERR_HIP;
  }
  return CRADLE_OK;
}

}  // extern "C"
