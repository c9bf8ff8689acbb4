// Dev experiment (not shipped): what bounds the BatchNorm-backward column-statistics pass (conv.hip: bn_colstats_kernel,
// MODE 1, mask recomputed from z) at R(2+1)D layer 1's shapes -- 3.9 TB/s at C = 144, 4.1 - 4.6 at C = 64 against the
// ~6 TB/s a streaming read reaches (the apply pass of the same layer: 5.5 - 5.9).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I data-efficient-video-transformers_amd/csrc tools/dev/bn_colstats_price.hip -o tools/_bin/bn_cs_price
// Variants: rows in flight per thread (U), software-pipelined loop (next round's loads requested before this round's
// arithmetic), row partition (workgroups per launch), and a plain two-tensor streaming sum as the ceiling.
#include "common.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

namespace {

struct Aff {
  float mu[8], is[8], s[8], t[8];
  __device__ void init(const float* mean, const float* invstd, const float* gamma, const float* beta, int c) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      mu[k] = mean[c + k]; is[k] = invstd[c + k];
      s[k] = is[k] * gamma[c + k];
      t[k] = fmaf(-mu[k], s[k], beta[c + k]);
    }
  }
};

template <int U, bool PIPE>
__global__ __launch_bounds__(256) void cs_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, const float* mean,
                                                 const float* invstd, const float* gamma, const float* beta, int64_t rows, int C,
                                                 int rows_per_block, int vc, float* __restrict__ partial) {
  __shared__ float red[2][256][8];
  const int nrl = 256 / vc;
  const int cl = threadIdx.x % vc, rl = threadIdx.x / vc;
  const int c = cl * 8;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = min(rows, r0 + rows_per_block);
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (r0 < r1 && rl < nrl) {
    Aff af;
    af.init(mean, invstd, gamma, beta, c);
    bf16x8 xv[2][U], dv[2][U];
    auto request = [&](int64_t rb, int s) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = min(rb + (int64_t)u * nrl, r1 - 1);
        xv[s][u] = *reinterpret_cast<const bf16x8*>(x + r * C + c);
        dv[s][u] = *reinterpret_cast<const bf16x8*>(dy + r * C + c);
      }
    };
    auto consume = [&](int64_t rb, int s) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (rb + (int64_t)u * nrl >= r1) break;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float xf = (float)xv[s][u][k];
          const float xh = (xf - af.mu[k]) * af.is[k];
          const float dz = fmaf(xf, af.s[k], af.t[k]) > 0.f ? (float)dv[s][u][k] : 0.f;
          a[k] += dz;
          b[k] = fmaf(dz, xh, b[k]);
        }
      }
    };
    const int64_t step = (int64_t)U * nrl;
    if (PIPE) {
      int64_t rb = r0 + rl;
      if (rb < r1) request(rb, 0);
      int s = 0;
      for (; rb < r1; rb += step, s ^= 1) {
        if (rb + step < r1) {
          if (s == 0) request(rb + step, 1); else request(rb + step, 0);
        }
        if (s == 0) consume(rb, 0); else consume(rb, 1);
      }
    } else {
      for (int64_t rb = r0 + rl; rb < r1; rb += step) {
        request(rb, 0);
        consume(rb, 0);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[0][threadIdx.x][k] = a[k]; red[1][threadIdx.x][k] = b[k]; }
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * vc * 8; t += 256) {
    const int st = t / (vc * 8), col = t % (vc * 8), cvi = col >> 3, k = col & 7;
    float acc = 0.f;
    for (int r = 0; r < nrl; ++r) acc += red[st][r * vc + cvi][k];
    partial[((int64_t)blockIdx.y * 2 + st) * C + cvi * 8 + k] = acc;
  }
}

// ceiling: both tensors read once, linearly, 4 x 16 bytes of each in flight per thread, grid-stride
__global__ __launch_bounds__(256) void read2_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, int64_t n8, float* out) {
  float acc = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += 4 * stride) {
    bf16x8 xv[4], dv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t j = min(i + u * stride, n8 - 1);
      xv[u] = *reinterpret_cast<const bf16x8*>(x + j * 8);
      dv[u] = *reinterpret_cast<const bf16x8*>(dy + j * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc = fmaf((float)xv[u][k], (float)dv[u][k], acc);
  }
  if (acc == 123.456f) out[0] = acc;
}

template <typename F> float time_us(F f, hipStream_t st) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ts;
  for (int rnd = 0; rnd < 5; ++rnd) {
    f();
    hipEventRecord(e0, st);
    for (int i = 0; i < 6; ++i) f();
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ts.push_back(ms * 1000.f / 6);
  }
  std::sort(ts.begin(), ts.end());
  return ts[2];
}

template <int U, bool PIPE>
float run(const bf16* x, const bf16* dy, const float* st, int64_t rows, int C, int parts, float* part, hipStream_t s) {
  const int vc = C / 8, nrl = 256 / vc;
  int64_t p = parts, cap = rows / (4 * nrl);
  if (p > cap) p = cap;
  const int rpb = (int)((rows + p - 1) / p);
  const int np = (int)((rows + rpb - 1) / rpb);
  return time_us([&] { hipLaunchKernelGGL((cs_kernel<U, PIPE>), dim3(1, np), dim3(256), 0, s, x, dy, st, st + C, st + 2 * C, st + 3 * C, rows, C, rpb, vc, part); }, s);
}

}  // namespace

int main() {
  const int64_t rows = 28LL * 12 * 56 * 56;
  hipStream_t s; hipStreamCreate(&s);
  for (int C : {144, 64}) {
    // three rotating buffer pairs (the step never re-reads a map from the Infinity Cache)
    const size_t n = (size_t)rows * C;
    bf16 *x[3], *dy[3];
    for (int i = 0; i < 3; ++i) { hipMalloc(&x[i], n * 2); hipMalloc(&dy[i], n * 2); hipMemset(x[i], 0x3c, n * 2); hipMemset(dy[i], 0x3b, n * 2); }
    float *st, *part;
    hipMalloc(&st, 4 * C * 4); hipMalloc(&part, (size_t)16384 * 2 * C * 4);
    std::vector<float> h(4 * C, 1.0f);
    hipMemcpy(st, h.data(), 4 * C * 4, hipMemcpyHostToDevice);
    const double mb = 2.0 * n * 2 / 1e6;
    printf("== C = %d, %lld rows, %.0f MB read per pass\n", C, (long long)rows, mb);
    int rot = 0;
    auto nx = [&]() { rot = (rot + 1) % 3; return rot; };
    {
      const float t = time_us([&] { int r = nx(); hipLaunchKernelGGL(read2_kernel, dim3(2048), dim3(256), 0, s, x[r], dy[r], (int64_t)(n / 8), part); }, s);
      printf("  plain streaming read of both tensors (2048 x 256 threads)   %7.1f us  %.2f TB/s\n", t, mb / t);
      const float t2 = time_us([&] { int r = nx(); hipLaunchKernelGGL(read2_kernel, dim3(8192), dim3(256), 0, s, x[r], dy[r], (int64_t)(n / 8), part); }, s);
      printf("  plain streaming read of both tensors (8192 x 256 threads)   %7.1f us  %.2f TB/s\n", t2, mb / t2);
    }
    for (int parts : {1024, 2048, 4096, 8192}) {
#define CELL(U, P) { int r = nx(); const float t = run<U, P>(x[r], dy[r], st, rows, C, parts, part, s); \
      printf("  parts %5d  U %d  %s  %7.1f us  %.2f TB/s\n", parts, U, P ? "pipelined" : "plain    ", t, mb / t); }
      CELL(4, false) CELL(8, false) CELL(2, true) CELL(4, true)
#undef CELL
    }
    for (int i = 0; i < 3; ++i) { hipFree(x[i]); hipFree(dy[i]); }
    hipFree(st); hipFree(part);
  }
  return 0;
}
