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
