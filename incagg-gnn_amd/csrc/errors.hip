// Error reporting: per-thread host message + a per-device "bad index" latch.
#include "common.h"

#include <cstring>
#include <mutex>

namespace incagg {

static thread_local char t_msg[512] = "";

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(t_msg, sizeof(t_msg), fmt, ap);
  va_end(ap);
  return code;
}

// Kernels set this word (plain vector store) when they skip an out-of-range id.
__device__ int g_bad_index = 0;

static constexpr int kMaxDevices = 64;
static int *s_flag_addr[kMaxDevices] = {};
static int *s_flag_host[kMaxDevices] = {};   // one pinned word per device: the flag is read into it
static std::mutex s_flag_mutex;

int error_flag_address(int **out) {
  int dev = 0;
  INCAGG_HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return fail(INCAGG_E_INVALID, "device id %d unsupported", dev);
  std::lock_guard<std::mutex> lock(s_flag_mutex);
  if (!s_flag_addr[dev]) {
    void *p = nullptr;
    INCAGG_HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(g_bad_index)));
    s_flag_addr[dev] = static_cast<int *>(p);
  }
  *out = s_flag_addr[dev];
  return INCAGG_OK;
}

// Pinned landing word for the flag of the current device (allocated once, kept for the life of
// the process): the read never goes through pageable host memory.
static int error_flag_host(int **out) {
  int dev = 0;
  INCAGG_HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return fail(INCAGG_E_INVALID, "device id %d unsupported", dev);
  std::lock_guard<std::mutex> lock(s_flag_mutex);
  if (!s_flag_host[dev]) {
    void *p = nullptr;
    INCAGG_HIP_TRY(hipHostMalloc(&p, sizeof(int), hipHostMallocDefault));
    s_flag_host[dev] = static_cast<int *>(p);
  }
  *out = s_flag_host[dev];
  return INCAGG_OK;
}

}  // namespace incagg

extern "C" {

const char *incagg_last_error(void) { return incagg::t_msg; }

int incagg_abi_version(void) { return INCAGG_ABI_VERSION; }

int incagg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int incagg_check_device_errors(void) {
  int *flag = nullptr;
  int rc = incagg::error_flag_address(&flag);
  if (rc) return rc;
  int *landing = nullptr;
  rc = incagg::error_flag_host(&landing);
  if (rc) return rc;
  INCAGG_HIP_TRY(hipDeviceSynchronize());
  *landing = 0;
  INCAGG_HIP_TRY(hipMemcpyAsync(landing, flag, sizeof(int), hipMemcpyDeviceToHost, nullptr));
  INCAGG_HIP_TRY(hipStreamSynchronize(nullptr));
  if (*landing) {
    INCAGG_HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int), nullptr));
    INCAGG_HIP_TRY(hipStreamSynchronize(nullptr));
    return incagg::fail(INCAGG_E_INDEX, "a kernel skipped an out-of-range row or column id");
  }
  return INCAGG_OK;
}

}  // extern "C"
