// Error plumbing + version for the C ABI.
#include "aph_device.h"
#include "aph_host.h"

static thread_local char g_err[512] = "";

int aph_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int aph_check_launch(const char* where) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return aph_fail(APH_ERR_HIP, "%s: %s", where, hipGetErrorString(e));
  return APH_OK;
}

extern "C" {
int aph_version(void) { return 100; }
const char* aph_last_error(void) { return g_err; }
}
