// ln_reduce.h -- column sums of per-workgroup partial rows [nparts][2][d] (dgamma | dbeta), shared by the LayerNorm
// backward (layernorm.hip) and the folded single-query attention backward (attention_cls.hip).
#pragma once
#include "common.h"

namespace {

// out[c] (+)= sum_p partial[p][c]   for c in [0, 2d): dgamma then dbeta.
// Block = 32 columns x 8 part-lanes; each thread sums nparts/8 partials (4 loads in
// flight), then an LDS tree over the 8 part-lanes.  Fixed order: reproducible.
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ partial,
                                                            int nparts, int d,
                                                            float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta,
                                                            int accumulate /* bit 0: dgamma, bit 1: dbeta */) {
  __shared__ float red[8][33];
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int64_t ld = 2 * (int64_t)d;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < 2 * d) {
    int p = pl;
    for (; p + 24 < nparts; p += 32) {
      a0 += partial[(int64_t)p * ld + c];
      a1 += partial[(int64_t)(p + 8) * ld + c];
      a2 += partial[(int64_t)(p + 16) * ld + c];
      a3 += partial[(int64_t)(p + 24) * ld + c];
    }
    for (; p < nparts; p += 8) a0 += partial[(int64_t)p * ld + c];
  }
  red[pl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (pl == 0 && c < 2 * d) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += red[i][cl];
    float* o = c < d ? dgamma + c : dbeta + (c - d);
    const bool acc = c < d ? (accumulate & 1) : (accumulate & 2);
    *o = acc ? *o + t : t;
  }
}

inline void dvt_ln_partials_reduce(const float* partial, int nparts, int d, float* dgamma, float* dbeta,
                                   int accumulate_gamma, int accumulate_beta, hipStream_t st) {
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((unsigned)dvt_cdiv(2 * d, 32)), dim3(256), 0, st, partial, nparts, d,
                     dgamma, dbeta, (accumulate_gamma ? 1 : 0) | (accumulate_beta ? 2 : 0));
}

}  // namespace
