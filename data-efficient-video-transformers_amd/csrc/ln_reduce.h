// ln_reduce.h -- column sums of per-workgroup partial rows [nparts][2][d] (dgamma | dbeta), shared by the LayerNorm
// backward (layernorm.hip) and the folded single-query attention backward (attention_cls.hip).
#pragma once
#include "common.h"

namespace {

// out[c] (+)= sum_p partial[p][c]   for c in [0, 2d): dgamma then dbeta.
// Block = 32 columns x PL part-lanes; each thread sums nparts / PL partials (4 loads in flight), then an LDS tree over
// the part-lanes.  PL = 32 for the 1,024 partial rows of a full-size launch: 8 part-lanes left every thread 32 dependent
// batches of loads (12 us for 4 MB); 8 for the short ones.  Fixed order: reproducible.
template <int PL>
__global__ __launch_bounds__(32 * PL) void ln_bwd_reduce_kernel(const float* __restrict__ partial, int nparts, int d,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                int accumulate /* bit 0: dgamma, bit 1: dbeta */) {
  __shared__ float red[PL][33];
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int64_t ld = 2 * (int64_t)d;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < 2 * d) {
    int p = pl;
    for (; p + 3 * PL < nparts; p += 4 * PL) {
      a0 += partial[(int64_t)p * ld + c];
      a1 += partial[(int64_t)(p + PL) * ld + c];
      a2 += partial[(int64_t)(p + 2 * PL) * ld + c];
      a3 += partial[(int64_t)(p + 3 * PL) * ld + c];
    }
    for (; p < nparts; p += PL) a0 += partial[(int64_t)p * ld + c];
  }
  red[pl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (pl < 4) {                                           // four lanes of the tree, then one
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < PL / 4; ++i) t += red[pl * (PL / 4) + i][cl];
    red[pl * (PL / 4)][cl] = t;
  }
  __syncthreads();
  if (pl == 0 && c < 2 * d) {
    const float t = (red[0][cl] + red[PL / 4][cl]) + (red[2 * (PL / 4)][cl] + red[3 * (PL / 4)][cl]);
    float* o = c < d ? dgamma + c : dbeta + (c - d);
    const bool acc = c < d ? (accumulate & 1) : (accumulate & 2);
    *o = acc ? *o + t : t;
  }
}

inline void dvt_ln_partials_reduce(const float* partial, int nparts, int d, float* dgamma, float* dbeta,
                                   int accumulate_gamma, int accumulate_beta, hipStream_t st) {
  const int acc = (accumulate_gamma ? 1 : 0) | (accumulate_beta ? 2 : 0);
  const dim3 grid((unsigned)dvt_cdiv(2 * d, 32));
  if (nparts > 128)
    hipLaunchKernelGGL(ln_bwd_reduce_kernel<32>, grid, dim3(1024), 0, st, partial, nparts, d, dgamma, dbeta, acc);
  else
    hipLaunchKernelGGL(ln_bwd_reduce_kernel<8>, grid, dim3(256), 0, st, partial, nparts, d, dgamma, dbeta, acc);
}

}  // namespace
