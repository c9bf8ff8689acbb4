// layernorm.hip -- row LayerNorm forward / backward (HBM-bound).
//
// One wave64 per row: the row lives in registers (VPL vectors of 8 elements per
// lane, 16-byte loads), statistics by wave shuffles (two-pass: mean, then the
// variance of the centred values), fp32 math.  Algorithmic traffic per row:
// fwd reads d and writes d elements (+8 B of statistics); bwd reads 2d, writes d.
#include "common.h"
#include "ln_reduce.h"

namespace {

constexpr int kLnBlock = 256;          // forward: 4 waves, one row each per iteration
constexpr int kLnBwdBlock = 256;       // backward: 4 waves per workgroup share one dgamma / dbeta partial row.  (16 waves and
constexpr int kLnBwdMaxBlocks = 1024;  // 256 partial rows: the reduce 7.3 -> 4.7 us, this kernel 37.5 -> 47.7 / 6.9 -> 15.5 us)

template <typename T, int VPL>
__global__ __launch_bounds__(kLnBlock) void ln_fwd_kernel(
    const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int64_t rows, int64_t n1,
    int d, int64_t xs0, int64_t xs1, int64_t ys0, int64_t ys1, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (kLnBlock / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (kLnBlock / 64);
  const float inv_d = 1.0f / (float)d;

  float g[VPL][8], b[VPL][8];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < d) {
      load8<float>(gamma + c, g[i]);
      load8<float>(beta + c, b[i]);
    }
  }

  for (int64_t r = wave; r < rows; r += nwaves) {
    const int64_t i0 = r / n1, i1 = r - i0 * n1;
    const T* xr = x + i0 * xs0 + i1 * xs1;
    T* yr = y + i0 * ys0 + i1 * ys1;
    float v[VPL][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (lane + 64 * i) * 8;
      if (c < d) {
        load8<T>(xr + c, v[i]);
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[i][k];
      }
    }
    const float mu = wave_sum_dpp(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (lane + 64 * i) * 8;
      if (c < d) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          v[i][k] -= mu;
          q = fmaf(v[i][k], v[i][k], q);
        }
      }
    }
    const float rs = rsqrtf(wave_sum_dpp(q) * inv_d + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (lane + 64 * i) * 8;
      if (c < d) {
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = fmaf(v[i][k] * rs, g[i][k], b[i][k]);
        store8<T>(yr + c, o);
      }
    }
    if (lane == 0) {
      mean[r] = mu;
      rstd[r] = rs;
    }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
// per-wave partial sums of dy*xhat (dgamma) and dy (dbeta) are combined across the
// block's 4 waves in LDS and written as one partial row per block.
template <typename T, int VPL>
__global__ __launch_bounds__(kLnBwdBlock) void ln_bwd_kernel(
    const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd, const T* dx_add,
    T* dx, float* __restrict__ partial /* [gridDim.x][2][d] */, int64_t rows, int64_t n1, int d,
    int64_t xs0, int64_t xs1, int64_t ys0, int64_t ys1,
    const T* __restrict__ dy_first, int64_t dyf_s, const T* __restrict__ dx_first, int64_t dxf_s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [2][d]
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * (kLnBwdBlock / 64) + wid;
  const int64_t nwaves = (int64_t)gridDim.x * (kLnBwdBlock / 64);
  const float inv_d = 1.0f / (float)d;

  float g[VPL][8], dg[VPL][8], db[VPL][8];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < d) load8<float>(gamma + c, g[i]);
#pragma unroll
    for (int k = 0; k < 8; ++k) { dg[i][k] = 0.f; db[i][k] = 0.f; }
  }

  for (int64_t r = wave; r < rows; r += nwaves) {
    const int64_t i0 = r / n1, i1 = r - i0 * n1;
    const T* xr = x + i0 * xs0 + i1 * xs1;
    const T* dyr = dy + i0 * ys0 + i1 * ys1;
    T* dxr = dx + i0 * xs0 + i1 * xs1;
    const float mu = mean[r], rs = rstd[r];
    float xh[VPL][8], gg[VPL][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (lane + 64 * i) * 8;
      if (c < d) {
        float xv[8], dv[8];
        load8<T>(xr + c, xv);
        load8<T>(dyr + c, dv);
        if (dy_first && i1 == 0) {          // a second gradient path into the first row of each group
          float fv[8];
          load8<T>(dy_first + i0 * dyf_s + c, fv);
#pragma unroll
          for (int k = 0; k < 8; ++k) dv[k] += fv[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          xh[i][k] = (xv[k] - mu) * rs;
          gg[i][k] = dv[k] * g[i][k];
          s1 += gg[i][k];
          s2 = fmaf(gg[i][k], xh[i][k], s2);
          dg[i][k] = fmaf(dv[k], xh[i][k], dg[i][k]);
          db[i][k] += dv[k];
        }
      }
    }
    const float c1 = wave_sum_dpp(s1) * inv_d;
    const float c2 = wave_sum_dpp(s2) * inv_d;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (lane + 64 * i) * 8;
      if (c < d) {
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = rs * (gg[i][k] - c1 - xh[i][k] * c2);
        if (dx_add) {
          float a[8];
          load8<T>(dx_add + (dxr - dx) + c, a);
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] += a[k];
        }
        if (dx_first && i1 == 0) {
          float a[8];
          load8<T>(dx_first + i0 * dxf_s + c, a);
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] += a[k];
        }
        store8<T>(dxr + c, o);
      }
    }
  }

  // combine the waves' partials: wave w adds in turn (fixed order: reproducible)
  for (int w = 0; w < kLnBwdBlock / 64; ++w) {
    if (wid == w) {
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < d) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            if (w == 0) { lds[c + k] = dg[i][k]; lds[d + c + k] = db[i][k]; }
            else { lds[c + k] += dg[i][k]; lds[d + c + k] += db[i][k]; }
          }
        }
      }
    }
    __syncthreads();
  }
  float* out = partial + (int64_t)blockIdx.x * 2 * d;
  for (int c = threadIdx.x; c < 2 * d; c += kLnBwdBlock) out[c] = lds[c];
}

int ln_check(const char* name, const void* a, const void* b, int64_t n0, int64_t n1, int64_t d,
             int64_t xs0, int64_t xs1, int64_t ys0, int64_t ys1) {
  DVT_REQUIRE(a && b, "%s: null pointer", name);
  DVT_REQUIRE(n0 >= 0 && n1 >= 1 && d >= 8, "%s: bad sizes", name);
  DVT_REQUIRE(d % 8 == 0, "%s: d = %lld must be a multiple of 8", name, (long long)d);
  if (d > 8 * 64 * 8) DVT_UNSUPPORTED("%s: d = %lld > 4096 not supported", name, (long long)d);
  DVT_REQUIRE(xs0 % 8 == 0 && xs1 % 8 == 0 && ys0 % 8 == 0 && ys1 % 8 == 0,
              "%s: row strides must be multiples of 8 elements", name);
  DVT_REQUIRE(dvt_aligned16(a) && dvt_aligned16(b), "%s: buffers must be 16-byte aligned", name);
  return DVT_OK;
}

}  // namespace

#define DVT_LN_VPL_SWITCH(vpl, ...)                    \
  switch (vpl) {                                       \
    case 1: { constexpr int VPL = 1; __VA_ARGS__; break; } \
    case 2: { constexpr int VPL = 2; __VA_ARGS__; break; } \
    case 3: { constexpr int VPL = 3; __VA_ARGS__; break; } \
    case 4: { constexpr int VPL = 4; __VA_ARGS__; break; } \
    default: { constexpr int VPL = 8; __VA_ARGS__; break; } \
  }

extern "C" {

int dvt_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                      float* rstd, int64_t n0, int64_t n1, int64_t d, int64_t xs0, int64_t xs1,
                      int64_t ys0, int64_t ys1, float eps, int dtype, dvt_stream_t stream) {
  int rc = ln_check("dvt_layernorm_fwd", x, y, n0, n1, d, xs0, xs1, ys0, ys1);
  if (rc) return rc;
  DVT_REQUIRE(gamma && beta && mean && rstd, "dvt_layernorm_fwd: null parameter/statistics pointer");
  DVT_REQUIRE(dvt_aligned16(gamma) && dvt_aligned16(beta), "dvt_layernorm_fwd: gamma/beta misaligned");
  const int64_t rows = n0 * n1;
  if (rows == 0) return DVT_OK;
  hipStream_t st = (hipStream_t)stream;
  int64_t blocks = dvt_cdiv(rows, kLnBlock / 64);
  const int64_t cap = (int64_t)dvt_num_cus() * 8;
  if (blocks > cap) blocks = cap;
  const int vpl = (int)dvt_cdiv(d, 8 * 64);
  DVT_DISPATCH_DTYPE(dtype, T, DVT_LN_VPL_SWITCH(vpl, hipLaunchKernelGGL(
      (ln_fwd_kernel<T, VPL>), dim3((unsigned)blocks), dim3(kLnBlock), 0, st, (const T*)x, gamma,
      beta, (T*)y, mean, rstd, rows, n1, (int)d, xs0, xs1, ys0, ys1, eps)));
  DVT_LAUNCH_CHECK("dvt_layernorm_fwd");
  return DVT_OK;
}

size_t dvt_layernorm_bwd_workspace_bytes(int64_t d) {
  return (size_t)kLnBwdMaxBlocks * 2 * (size_t)(d > 0 ? d : 0) * sizeof(float);
}

int dvt_layernorm_bwd_first(const void* dy, const void* x, const float* gamma, const float* mean,
                            const float* rstd, const void* dx_add, void* dx, float* dgamma, float* dbeta,
                            void* workspace,
                            int64_t n0, int64_t n1, int64_t d, int64_t xs0, int64_t xs1, int64_t ys0,
                            int64_t ys1, const void* dy_first, int64_t dy_first_stride,
                            const void* dx_first, int64_t dx_first_stride, int dtype,
                            int accumulate_gamma, int accumulate_beta, dvt_stream_t stream) {
  int rc = ln_check("dvt_layernorm_bwd", x, dy, n0, n1, d, xs0, xs1, ys0, ys1);
  if (rc) return rc;
  DVT_REQUIRE(gamma && mean && rstd && dx && dgamma && dbeta && workspace,
              "dvt_layernorm_bwd: null pointer");
  DVT_REQUIRE(dvt_aligned16(dx) && dvt_aligned16(gamma) && dvt_aligned16(workspace) && dvt_aligned16(dx_add) &&
                  dvt_aligned16(dy_first) && dvt_aligned16(dx_first),
              "dvt_layernorm_bwd: buffers must be 16-byte aligned");
  DVT_REQUIRE(dy_first_stride % 8 == 0 && dx_first_stride % 8 == 0,
              "dvt_layernorm_bwd: first-row strides must be multiples of 8 elements");
  const int64_t rows = n0 * n1;
  DVT_REQUIRE(rows > 0, "dvt_layernorm_bwd: no rows");
  hipStream_t st = (hipStream_t)stream;
  int64_t blocks = dvt_cdiv(rows, kLnBwdBlock / 64);
  if (blocks > kLnBwdMaxBlocks) blocks = kLnBwdMaxBlocks;
  const int vpl = (int)dvt_cdiv(d, 8 * 64);
  const size_t lds = 2 * (size_t)d * sizeof(float);
  float* partial = (float*)workspace;
  DVT_DISPATCH_DTYPE(dtype, T, DVT_LN_VPL_SWITCH(vpl, hipLaunchKernelGGL(
      (ln_bwd_kernel<T, VPL>), dim3((unsigned)blocks), dim3(kLnBwdBlock), lds, st, (const T*)dy,
      (const T*)x, gamma, mean, rstd, (const T*)dx_add, (T*)dx, partial, rows, n1, (int)d, xs0, xs1, ys0, ys1,
      (const T*)dy_first, dy_first_stride, (const T*)dx_first, dx_first_stride)));
  DVT_LAUNCH_CHECK("dvt_layernorm_bwd");
  dvt_ln_partials_reduce(partial, (int)blocks, (int)d, dgamma, dbeta, accumulate_gamma, accumulate_beta, st);
  DVT_LAUNCH_CHECK("dvt_layernorm_bwd(reduce)");
  return DVT_OK;
}

int dvt_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                      const float* rstd, const void* dx_add, void* dx, float* dgamma, float* dbeta,
                      void* workspace,
                      int64_t n0, int64_t n1, int64_t d, int64_t xs0, int64_t xs1, int64_t ys0,
                      int64_t ys1, int dtype, int accumulate, dvt_stream_t stream) {
  return dvt_layernorm_bwd_first(dy, x, gamma, mean, rstd, dx_add, dx, dgamma, dbeta, workspace, n0, n1, d, xs0,
                                 xs1, ys0, ys1, nullptr, 0, nullptr, 0, dtype, accumulate, accumulate, stream);
}

}  // extern "C"
