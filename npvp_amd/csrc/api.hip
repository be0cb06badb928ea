// Library-wide entry points: version, last-error string.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void npvp_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* npvp_last_error(void) { return g_err; }
extern "C" int npvp_version(void) { return 100; }

// kernels launched by this library in this process so far (every launch site is an NPVP_LAUNCH, common.h)
namespace npvp { long long g_launches = 0; }
extern "C" long long npvp_launch_count(void) { return __atomic_load_n(&npvp::g_launches, __ATOMIC_RELAXED); }

// A HIP stream of the LOWEST priority the device offers (PyTorch only hands out normal / high).  The gradient stream
// (npvp_amd.ops.WgradStream) is created with it so that, whenever CUs free up, the kernels of the critical
// forward/backward chain are dispatched before the queued weight-gradient workgroups.  *least / *greatest receive the
// device's priority range (nullable).  The stream lives until npvp_stream_destroy.
extern "C" void* npvp_stream_create_low_priority(int* least, int* greatest) {
  int lo = 0, hi = 0;
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = hi = 0; }
  if (least) *least = lo;
  if (greatest) *greatest = hi;
  hipStream_t s = nullptr;
  if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo) != hipSuccess) {
    npvp_set_error("stream_create_low_priority: hipStreamCreateWithPriority failed");
    return nullptr;
  }
  return (void*)s;
}
extern "C" int npvp_stream_destroy(void* s) {
  return hipStreamDestroy((hipStream_t)s) == hipSuccess ? 0 : -3;
}
