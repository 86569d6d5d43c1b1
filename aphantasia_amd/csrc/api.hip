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

#ifndef APH_EMU
#include <mutex>
#include <unordered_map>
void aph_allow_smem(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::unordered_map<unsigned long long, int> granted;     // (kernel, device) -> largest size granted
  int dev = 0;
  APH_HIP(hipGetDevice(&dev));
  const unsigned long long key = (unsigned long long)(uintptr_t)kernel * 64ull + (unsigned long long)(dev & 63);
  std::lock_guard<std::mutex> lk(mu);
  auto it = granted.find(key);
  if (it != granted.end() && it->second >= bytes) return;
  APH_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  granted[key] = bytes;
}
#endif

extern "C" {
int aph_version(void) { return 200; }
const char* aph_last_error(void) { return g_err; }
}
