// Dev probe: what one CU's vector-memory pipeline sustains (bytes per shader clock), one workgroup per CU.
//   read : every lane loads 16 B (global_load_dwordx4), coalesced 1 KiB per wave-instruction, from a buffer that fits L2
//          (per-XCD working set 2 MiB) or streams from HBM (1 GiB)
//   write: every lane stores 16 B, coalesced, streaming
// build: hipcc --offload-arch=gfx950 -O3 tools/cu_bw_probe.hip -o tools/_bin/cu_bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(1024) void read_kernel(const v4i* __restrict__ src, size_t n_per_wg, size_t wrap, int* sink,
                                                    long long* ticks) {
  const size_t base = ((size_t)blockIdx.x * n_per_wg) % wrap;
  v4i acc = {0, 0, 0, 0};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (size_t i = threadIdx.x; i + (UNROLL - 1) * blockDim.x < n_per_wg; i += UNROLL * blockDim.x) {
    v4i v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = src[(base + i + u * blockDim.x) % wrap];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (acc.x == 0x12345678) *sink = acc.y;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int UNROLL>
__global__ __launch_bounds__(1024) void write_kernel(v4i* __restrict__ dst, size_t n_per_wg, long long* ticks) {
  const size_t base = (size_t)blockIdx.x * n_per_wg;
  const v4i val = {(int)threadIdx.x, 1, 2, 3};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (size_t i = threadIdx.x; i + (UNROLL - 1) * blockDim.x < n_per_wg; i += UNROLL * blockDim.x) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) dst[base + i + u * blockDim.x] = val;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
  const int cus = 256;
  const size_t bytes = (size_t)1 << 30;
  v4i* buf; int* sink; long long* ticks;
  hipMalloc(&buf, bytes); hipMalloc(&sink, 4); hipMalloc(&ticks, cus * 8);
  hipMemset(buf, 1, bytes);
  std::vector<long long> h(cus);
  auto report = [&](const char* what, size_t bytes_per_wg) {
    hipDeviceSynchronize();
    hipMemcpy(h.data(), ticks, cus * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto t : h) s += (double)t;
    printf("%-58s %6.1f B/tick per CU\n", what, (double)bytes_per_wg / (s / cus));
  };
  const size_t per_wg = (size_t)4 << 20;                 // 4 MiB per workgroup
#define SWEEP(U)                                                                                                        \
  for (int threads : {256, 512, 1024}) {                                                                                \
    char name[128];                                                                                                      \
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((read_kernel<U>), dim3(cus), dim3(threads), 0, 0, buf, per_wg / 16, ((size_t)2 << 20) / 16, sink, ticks); \
    snprintf(name, sizeof name, "read  16 B/lane, %4d threads, %2d in flight, L2-resident:", threads, U); report(name, per_wg); \
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((read_kernel<U>), dim3(cus), dim3(threads), 0, 0, buf, per_wg / 16, bytes / 16, sink, ticks); \
    snprintf(name, sizeof name, "read  16 B/lane, %4d threads, %2d in flight, HBM stream :", threads, U); report(name, per_wg); \
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((write_kernel<U>), dim3(cus), dim3(threads), 0, 0, buf, per_wg / 16, ticks); \
    snprintf(name, sizeof name, "write 16 B/lane, %4d threads, %2d per loop trip, streaming:", threads, U); report(name, per_wg); \
  }
  SWEEP(2) SWEEP(4) SWEEP(8) SWEEP(16)
  // one CU alone (no contention from the other 255)
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((read_kernel<8>), dim3(1), dim3(512), 0, 0, buf, per_wg / 16, ((size_t)2 << 20) / 16, sink, ticks);
  hipDeviceSynchronize(); hipMemcpy(h.data(), ticks, 8, hipMemcpyDeviceToHost);
  printf("%-58s %6.1f B/tick\n", "read  16 B/lane, 512 threads, ONE workgroup, L2-resident:", (double)per_wg / h[0]);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((write_kernel<8>), dim3(1), dim3(512), 0, 0, buf, per_wg / 16, ticks);
  hipDeviceSynchronize(); hipMemcpy(h.data(), ticks, 8, hipMemcpyDeviceToHost);
  printf("%-58s %6.1f B/tick\n", "write 16 B/lane, 512 threads, ONE workgroup:", (double)per_wg / h[0]);
  return 0;
}
