// runtime.hip -- error reporting, version and device queries of the C ABI.
#include "common.h"

#include <mutex>
#include <string.h>

namespace {
thread_local char g_err[512] = "";
std::once_flag g_dev_once;
int g_cus = 256;
int g_lds = 160 * 1024;
char g_arch[64] = "";

void init_dev() {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) {
    g_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    g_lds = (int)prop.sharedMemPerBlock;
    strncpy(g_arch, prop.gcnArchName, sizeof(g_arch) - 1);
  }
}
}  // namespace

int dvt_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int dvt_fail_hip(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: HIP error %d (%s)", where, (int)e, hipGetErrorString(e));
  return DVT_ERR_HIP;
}

int dvt_num_cus() {
  std::call_once(g_dev_once, init_dev);
  return g_cus;
}

namespace {
// one lane spins on the 100 MHz wall clock for `ticks`; bounded (the loop also ends after a fixed number of polls)
__global__ void delay_kernel(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  for (int i = 0; i < (1 << 24); ++i) {
    if (wall_clock64() - t0 >= ticks) break;
    __builtin_amdgcn_s_sleep(64);
  }
}
}  // namespace

extern "C" {

int dvt_device_delay(uint64_t microseconds, dvt_stream_t stream) {
  DVT_REQUIRE(microseconds <= 1000000, "dvt_device_delay: at most one second");
  hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long)microseconds * 100ull);
  DVT_LAUNCH_CHECK("dvt_device_delay");
  return DVT_OK;
}

int dvt_zero(void* dst, size_t nbytes, dvt_stream_t stream) {
  DVT_REQUIRE(dst || nbytes == 0, "dvt_zero: null pointer");
  if (nbytes == 0) return DVT_OK;
  const hipError_t e = hipMemsetAsync(dst, 0, nbytes, (hipStream_t)stream);
  if (e != hipSuccess) return dvt_fail_hip(e, "dvt_zero");
  return DVT_OK;
}

int dvt_version(void) { return DVT_ABI_VERSION; }

const char* dvt_last_error(void) { return g_err; }

int dvt_device_info(int* cu_count, int* lds_bytes, char* arch, int arch_len) {
  std::call_once(g_dev_once, init_dev);
  if (cu_count) *cu_count = g_cus;
  if (lds_bytes) *lds_bytes = g_lds;
  if (arch && arch_len > 0) {
    strncpy(arch, g_arch, (size_t)arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return DVT_OK;
}

}  // extern "C"
