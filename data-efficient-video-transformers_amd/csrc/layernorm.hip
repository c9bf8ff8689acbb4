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

// TX / TY: element types of x and y.  They differ for the fp32 residual stream of the launch-bound zone (the 33-token
// temporal encoder): x fp32 -> y 16-bit (the operand of the GEMM behind the norm), or x 16-bit -> y fp32 (the CLS rows of
// the space stack entering that stream).
template <typename TX, typename TY, int VPL>
__global__ __launch_bounds__(kLnBlock) void ln_fwd_kernel(
    const TX* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    TY* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int64_t rows, int64_t n1,
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
    const TX* xr = x + i0 * xs0 + i1 * xs1;
    TY* yr = y + i0 * ys0 + i1 * ys1;
    float v[VPL][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (lane + 64 * i) * 8;
      if (c < d) {
        load8<TX>(xr + c, v[i]);
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
        store8<TY>(yr + c, o);
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
// TDY: element type of dy / dy_first; TX: of x; TDX: of dx / dx_add / dx_first; TL: of the optional second copy dx_lp of dx
// (the 16-bit operand of the GEMMs behind an fp32 gradient stream; TL = TDX when unused).
template <typename TDY, typename TX, typename TDX, typename TL, int VPL>
__global__ __launch_bounds__(kLnBwdBlock) void ln_bwd_kernel(
    const TDY* __restrict__ dy, const TX* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd, const TDX* dx_add,
    TDX* dx, TL* __restrict__ dx_lp, float* __restrict__ partial /* [gridDim.x][2][d] */, int64_t rows, int64_t n1, int d,
    int64_t xs0, int64_t xs1, int64_t ys0, int64_t ys1,
    const TDY* __restrict__ dy_first, int64_t dyf_s, const TDX* __restrict__ dx_first, int64_t dxf_s) {
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
    const TX* xr = x + i0 * xs0 + i1 * xs1;
    const TDY* dyr = dy + i0 * ys0 + i1 * ys1;
    const int64_t xoff = i0 * xs0 + i1 * xs1;
    TDX* dxr = dx + xoff;
    const float mu = mean[r], rs = rstd[r];
    float xh[VPL][8], gg[VPL][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (lane + 64 * i) * 8;
      if (c < d) {
        float xv[8], dv[8];
        load8<TX>(xr + c, xv);
        load8<TDY>(dyr + c, dv);
        if (dy_first && i1 == 0) {          // a second gradient path into the first row of each group
          float fv[8];
          load8<TDY>(dy_first + i0 * dyf_s + c, fv);
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
          load8<TDX>(dx_add + xoff + c, a);
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] += a[k];
        }
        if (dx_first && i1 == 0) {
          float a[8];
          load8<TDX>(dx_first + i0 * dxf_s + c, a);
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] += a[k];
        }
        store8<TDX>(dxr + c, o);
        if (dx_lp) store8<TL>(dx_lp + xoff + c, o);
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

// dispatch on (x dtype, y dtype): equal types, or fp32 on exactly one side with a 16-bit type on the other
#define DVT_LN_FWD_LAUNCH(TX, TY)                                                                                          \
  DVT_LN_VPL_SWITCH(vpl, hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, VPL>), dim3((unsigned)blocks), dim3(kLnBlock), 0, st,    \
                                            (const TX*)x, gamma, beta, (TY*)y, mean, rstd, rows, n1, (int)d, xs0, xs1, ys0, \
                                            ys1, eps))

#define DVT_LN_BWD_LAUNCH(TDY, TX, TDX, TL)                                                                                \
  DVT_LN_VPL_SWITCH(vpl, hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, TL, VPL>), dim3((unsigned)blocks),                 \
                                            dim3(kLnBwdBlock), lds, st, (const TDY*)q->dy, (const TX*)q->x, q->gamma,       \
                                            q->mean, q->rstd, (const TDX*)q->dx_add, (TDX*)q->dx, (TL*)q->dx_lp, partial,   \
                                            rows, q->n1, (int)q->d, q->xs0, q->xs1, q->ys0, q->ys1, (const TDY*)q->dy_first, \
                                            q->dy_first_stride, (const TDX*)q->dx_first, q->dx_first_stride))

namespace {

// up to kLnGroup deferred dgamma / dbeta reduces in ONE launch: blockIdx.y names the entry
constexpr int kLnGroup = 32;
struct LnGroup { dvt_ln_pending e[kLnGroup]; };

__global__ __launch_bounds__(1024) void ln_bwd_reduce_group_kernel(const LnGroup grp) {
  const dvt_ln_pending& q = grp.e[blockIdx.y];
  constexpr int PL = 32;
  __shared__ float red[PL][33];
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int d = q.d;
  if (blockIdx.x * 32 >= 2 * d) return;              // (entries may differ in d: the grid is sized for the widest)
  const int64_t ld = 2 * (int64_t)d;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < 2 * d) {
    int p = pl;
    for (; p + 3 * PL < q.nparts; p += 4 * PL) {
      a0 += q.partial[(int64_t)p * ld + c];
      a1 += q.partial[(int64_t)(p + PL) * ld + c];
      a2 += q.partial[(int64_t)(p + 2 * PL) * ld + c];
      a3 += q.partial[(int64_t)(p + 3 * PL) * ld + c];
    }
    for (; p < q.nparts; p += PL) a0 += q.partial[(int64_t)p * ld + c];
  }
  red[pl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (pl < 4) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < PL / 4; ++i) t += red[pl * (PL / 4) + i][cl];
    red[pl * (PL / 4)][cl] = t;
  }
  __syncthreads();
  if (pl == 0 && c < 2 * d) {
    const float t = (red[0][cl] + red[PL / 4][cl]) + (red[2 * (PL / 4)][cl] + red[3 * (PL / 4)][cl]);
    float* o = c < d ? q.dgamma + c : q.dbeta + (c - d);
    const bool acc = c < d ? (q.accumulate & 1) : (q.accumulate & 2);
    *o = acc ? *o + t : t;
  }
}

}  // namespace

extern "C" {

int dvt_layernorm_fwd_mixed(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype, float* mean,
                            float* rstd, int64_t n0, int64_t n1, int64_t d, int64_t xs0, int64_t xs1,
                            int64_t ys0, int64_t ys1, float eps, dvt_stream_t stream) {
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
  if (x_dtype == y_dtype) {
    DVT_DISPATCH_DTYPE(x_dtype, T, DVT_LN_FWD_LAUNCH(T, T));
  } else if (x_dtype == DVT_F32 && dvt_is_16bit(y_dtype)) {
    DVT_DISPATCH_16BIT(y_dtype, E, DVT_LN_FWD_LAUNCH(float, E));
  } else if (y_dtype == DVT_F32 && dvt_is_16bit(x_dtype)) {
    DVT_DISPATCH_16BIT(x_dtype, E, DVT_LN_FWD_LAUNCH(E, float));
  } else {
    DVT_UNSUPPORTED("dvt_layernorm_fwd: x dtype %d -> y dtype %d (equal types, or fp32 on one side)", x_dtype, y_dtype);
  }
  DVT_LAUNCH_CHECK("dvt_layernorm_fwd");
  return DVT_OK;
}

int dvt_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                      float* rstd, int64_t n0, int64_t n1, int64_t d, int64_t xs0, int64_t xs1,
                      int64_t ys0, int64_t ys1, float eps, int dtype, dvt_stream_t stream) {
  return dvt_layernorm_fwd_mixed(x, dtype, gamma, beta, y, dtype, mean, rstd, n0, n1, d, xs0, xs1, ys0, ys1, eps, stream);
}

size_t dvt_layernorm_bwd_workspace_bytes(int64_t d) {
  return (size_t)kLnBwdMaxBlocks * 2 * (size_t)(d > 0 ? d : 0) * sizeof(float);
}

size_t dvt_layernorm_bwd_partial_bytes(int64_t rows, int64_t d) {
  int64_t blocks = dvt_cdiv(rows > 0 ? rows : 1, kLnBwdBlock / 64);
  if (blocks > kLnBwdMaxBlocks) blocks = kLnBwdMaxBlocks;
  return (size_t)blocks * 2 * (size_t)(d > 0 ? d : 0) * sizeof(float);
}

int dvt_layernorm_bwd_ex(const dvt_ln_bwd_desc* q, dvt_stream_t stream) {
  DVT_REQUIRE(q, "dvt_layernorm_bwd: null descriptor");
  int rc = ln_check("dvt_layernorm_bwd", q->x, q->dy, q->n0, q->n1, q->d, q->xs0, q->xs1, q->ys0, q->ys1);
  if (rc) return rc;
  DVT_REQUIRE(q->gamma && q->mean && q->rstd && q->dx && q->dgamma && q->dbeta && q->workspace, "dvt_layernorm_bwd: null pointer");
  DVT_REQUIRE(dvt_aligned16(q->dx) && dvt_aligned16(q->gamma) && dvt_aligned16(q->workspace) && dvt_aligned16(q->dx_add) &&
                  dvt_aligned16(q->dy_first) && dvt_aligned16(q->dx_first) && dvt_aligned16(q->dx_lp),
              "dvt_layernorm_bwd: buffers must be 16-byte aligned");
  DVT_REQUIRE(q->dy_first_stride % 8 == 0 && q->dx_first_stride % 8 == 0,
              "dvt_layernorm_bwd: first-row strides must be multiples of 8 elements");
  DVT_REQUIRE(!q->defer_reduce || q->pending, "dvt_layernorm_bwd: defer_reduce needs a pending descriptor to fill");
  const int64_t rows = q->n0 * q->n1;
  DVT_REQUIRE(rows > 0, "dvt_layernorm_bwd: no rows");
  hipStream_t st = (hipStream_t)stream;
  int64_t blocks = dvt_cdiv(rows, kLnBwdBlock / 64);
  if (blocks > kLnBwdMaxBlocks) blocks = kLnBwdMaxBlocks;
  const int64_t d = q->d;
  const int vpl = (int)dvt_cdiv(d, 8 * 64);
  const size_t lds = 2 * (size_t)d * sizeof(float);
  float* partial = (float*)q->workspace;
  const int dyt = q->dy_dtype, xt = q->x_dtype, dxt = q->dx_dtype;
  if (dyt == xt && xt == dxt && !q->dx_lp) {
    DVT_DISPATCH_DTYPE(xt, T, DVT_LN_BWD_LAUNCH(T, T, T, T));
  } else if (dvt_is_16bit(dyt) && xt == DVT_F32 && dxt == DVT_F32 && (!q->dx_lp || q->dx_lp_dtype == dyt)) {
    // the fp32 residual stream of the launch-bound zone: 16-bit gradient in, fp32 gradient stream out (+ its 16-bit copy)
    DVT_DISPATCH_16BIT(dyt, E, DVT_LN_BWD_LAUNCH(E, float, float, E));
  } else if (dyt == DVT_F32 && dvt_is_16bit(xt) && dxt == xt && !q->dx_lp) {
    // the seam where that stream ends: fp32 gradient in, 16-bit map (the space stack's CLS rows) and gradient out
    DVT_DISPATCH_16BIT(xt, E, DVT_LN_BWD_LAUNCH(float, E, E, E));
  } else {
    DVT_UNSUPPORTED("dvt_layernorm_bwd: dtype combination (dy %d, x %d, dx %d, dx_lp %d) not instantiated", dyt, xt, dxt,
                    q->dx_lp ? q->dx_lp_dtype : -1);
  }
  DVT_LAUNCH_CHECK("dvt_layernorm_bwd");
  if (q->defer_reduce) {                          // dgamma / dbeta: left to dvt_layernorm_reduce_group (many layers, one launch)
    dvt_ln_pending* pn = q->pending;
    pn->partial = partial; pn->nparts = (int)blocks; pn->d = (int)d; pn->dgamma = q->dgamma; pn->dbeta = q->dbeta;
    pn->accumulate = (q->accumulate_gamma ? 1 : 0) | (q->accumulate_beta ? 2 : 0); pn->valid = 1;
    return DVT_OK;
  }
  dvt_ln_partials_reduce(partial, (int)blocks, (int)d, q->dgamma, q->dbeta, q->accumulate_gamma, q->accumulate_beta, st);
  DVT_LAUNCH_CHECK("dvt_layernorm_bwd(reduce)");
  return DVT_OK;
}

int dvt_layernorm_reduce_group(const dvt_ln_pending* list, int count, dvt_stream_t stream) {
  DVT_REQUIRE(count >= 0 && (count == 0 || list), "dvt_layernorm_reduce_group: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  for (int base = 0; base < count; base += kLnGroup) {
    LnGroup grp{};
    int n = 0, dmax = 0;
    for (int i = base; i < count && n < kLnGroup; ++i) {
      if (!list[i].valid) continue;
      DVT_REQUIRE(list[i].partial && list[i].dgamma && list[i].dbeta && list[i].nparts > 0 && list[i].d > 0,
                  "dvt_layernorm_reduce_group: bad entry");
      grp.e[n++] = list[i];
      if (list[i].d > dmax) dmax = list[i].d;
    }
    if (n == 0) continue;
    hipLaunchKernelGGL(ln_bwd_reduce_group_kernel, dim3((unsigned)dvt_cdiv(2 * dmax, 32), (unsigned)n), dim3(1024), 0, st, grp);
    DVT_LAUNCH_CHECK("dvt_layernorm_reduce_group");
  }
  return DVT_OK;
}

int dvt_layernorm_bwd_first(const void* dy, const void* x, const float* gamma, const float* mean,
                            const float* rstd, const void* dx_add, void* dx, float* dgamma, float* dbeta,
                            void* workspace,
                            int64_t n0, int64_t n1, int64_t d, int64_t xs0, int64_t xs1, int64_t ys0,
                            int64_t ys1, const void* dy_first, int64_t dy_first_stride,
                            const void* dx_first, int64_t dx_first_stride, int dtype,
                            int accumulate_gamma, int accumulate_beta, dvt_stream_t stream) {
  dvt_ln_bwd_desc q{};
  q.dy = dy; q.dy_dtype = dtype; q.x = x; q.x_dtype = dtype; q.gamma = gamma; q.mean = mean; q.rstd = rstd;
  q.dx_add = dx_add; q.dx = dx; q.dx_dtype = dtype; q.dgamma = dgamma; q.dbeta = dbeta; q.workspace = workspace;
  q.n0 = n0; q.n1 = n1; q.d = d; q.xs0 = xs0; q.xs1 = xs1; q.ys0 = ys0; q.ys1 = ys1;
  q.dy_first = dy_first; q.dy_first_stride = dy_first_stride; q.dx_first = dx_first; q.dx_first_stride = dx_first_stride;
  q.accumulate_gamma = accumulate_gamma; q.accumulate_beta = accumulate_beta;
  return dvt_layernorm_bwd_ex(&q, stream);
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
