// fusion_ssl.hip -- multi-modal gating and the contrastive (NT-Xent) objective.
//
//   * row L2 normalisation  F.normalize(x) / the cosine-similarity normaliser
//       (src/models/collabgating.py:70 GatedEmbeddingUnit; src/models/losses/ntxent.py:63)
//   * context gating        F.glu(cat(x, x + x1), -1) = x * sigmoid(x + x1)   (collabgating.py:83-86)
//   * contrastive row loss  -log( exp(pos/T) / sum_{j != k} exp(sim_kj/T) )      (ntxent.py:66-74)
// Small HBM-bound kernels; the similarity matrix itself is a GEMM (dvt_gemm).
#include "common.h"

namespace {

// one wave per row: y = x / max(||x||, eps); inv[r] = 1 / max(||x||, eps)
template <typename T>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                         float* __restrict__ inv, int64_t rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) { const float v = to_f32<T>(x[r * D + c]); s = fmaf(v, v, s); }
  s = wave_sum(s);
  const float iv = 1.0f / fmaxf(sqrtf(s), eps);
  for (int c = lane; c < D; c += 64) y[r * D + c] = from_f32<T>(to_f32<T>(x[r * D + c]) * iv);
  if (lane == 0) inv[r] = iv;
}

// dx = inv * (dy - y * <y, dy>)      (rows clamped by eps have y = x*inv with ||y|| < 1: same formula as torch
// only away from the clamp; the clamp region is the zero vector, whose gradient is inv * dy)
template <typename T>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                         const float* __restrict__ inv, T* __restrict__ dx,
                                                         int64_t rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float dot = 0.f;
  for (int c = lane; c < D; c += 64) dot = fmaf(to_f32<T>(y[r * D + c]), to_f32<T>(dy[r * D + c]), dot);
  dot = wave_sum(dot);
  const float iv = inv[r];
  const bool clamped = iv >= 1.0f / eps;
  for (int c = lane; c < D; c += 64) {
    const float g = to_f32<T>(dy[r * D + c]);
    dx[r * D + c] = from_f32<T>(clamped ? iv * g : iv * (g - to_f32<T>(y[r * D + c]) * dot));
  }
}

// out[r] = <a[r], b[r]> / max(||a[r]|| * ||b[r]||, eps)     (nn.CosineSimilarity(dim=1), frame_transformer.py:121,257)
template <typename T>
__global__ __launch_bounds__(256) void cosine_rows_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                          float* __restrict__ out, int64_t rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float ab = 0.f, aa = 0.f, bb = 0.f;
  for (int c = lane; c < D; c += 64) {
    const float x = to_f32<T>(a[r * D + c]), y = to_f32<T>(b[r * D + c]);
    ab = fmaf(x, y, ab); aa = fmaf(x, x, aa); bb = fmaf(y, y, bb);
  }
  ab = wave_sum(ab); aa = wave_sum(aa); bb = wave_sum(bb);
  if (lane == 0) out[r] = ab / fmaxf(sqrtf(aa) * sqrtf(bb), eps);
}

// y = a * sigmoid(b)
template <typename T>
__global__ void gate_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float s = 1.0f / (1.0f + expf(-to_f32<T>(b[i])));
    y[i] = from_f32<T>(to_f32<T>(a[i]) * s);
  }
}

// da = dy * s;  db = dy * a * s * (1 - s)
template <typename T>
__global__ void gate_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ a, const T* __restrict__ b,
                                T* __restrict__ da, T* __restrict__ db, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float s = 1.0f / (1.0f + expf(-to_f32<T>(b[i])));
    const float g = to_f32<T>(dy[i]);
    da[i] = from_f32<T>(g * s);
    db[i] = from_f32<T>(g * to_f32<T>(a[i]) * s * (1.0f - s));
  }
}

// sim [M, M] f32 (M = 2B): row k's positive is column (k + B) mod M.  One wave per row.
// row_loss[k] = log(sum_{j != k} exp(sim_kj / T)) - sim_k,pos / T
__global__ __launch_bounds__(256) void contrastive_rows_fwd_kernel(const float* __restrict__ sim, int M, int B,
                                                                   float inv_t, float* __restrict__ row_loss,
                                                                   float* __restrict__ row_lse) {
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= M) return;
  const float* s = sim + (int64_t)k * M;
  float mx = -INFINITY;
  for (int j = lane; j < M; j += 64) if (j != k) mx = fmaxf(mx, s[j] * inv_t);
  mx = wave_max(mx);
  float acc = 0.f;
  for (int j = lane; j < M; j += 64) if (j != k) acc += expf(s[j] * inv_t - mx);
  acc = wave_sum(acc);
  if (lane == 0) {
    const float lse = mx + logf(acc);
    row_lse[k] = lse;
    row_loss[k] = lse - s[(k + B) % M] * inv_t;
  }
}

// dsim_kj = gscale * inv_t * (softmax_kj [j != k] - 1[j == pos(k)]),  gscale = dloss / M
__global__ void contrastive_rows_bwd_kernel(const float* __restrict__ sim, const float* __restrict__ row_lse, int M,
                                            int B, float inv_t, const float* __restrict__ gloss, float inv_m,
                                            float* __restrict__ dsim) {
  const int64_t total = (int64_t)M * M;
  const float g = gloss[0] * inv_m * inv_t;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i / M), j = (int)(i % M);
    float p = (j == k) ? 0.f : expf(sim[i] * inv_t - row_lse[k]);
    if (j == (k + B) % M) p -= 1.0f;
    dsim[i] = g * p;
  }
}

// out[0] = mean of x[0..n)   (fixed order, single block)
__global__ void mean_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)x[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(red[0] / n);
}

inline int egrid(int64_t n) {
  int64_t b = dvt_cdiv(n, 256);
  const int64_t cap = (int64_t)dvt_num_cus() * 8;
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int dvt_l2norm_rows_fwd(const void* x, void* y, float* inv_norm, int64_t rows, int D, float eps, int dtype,
                        dvt_stream_t stream) {
  DVT_REQUIRE(x && y && inv_norm && rows >= 0 && D > 0 && eps > 0.f, "dvt_l2norm_rows_fwd: bad arguments");
  if (rows == 0) return DVT_OK;
  DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((l2norm_fwd_kernel<T>), dim3((unsigned)dvt_cdiv(rows, 4)), dim3(256), 0,
                                                  (hipStream_t)stream, (const T*)x, (T*)y, inv_norm, rows, D, eps));
  DVT_LAUNCH_CHECK("dvt_l2norm_rows_fwd");
  return DVT_OK;
}

int dvt_l2norm_rows_bwd(const void* dy, const void* y, const float* inv_norm, void* dx, int64_t rows, int D, float eps,
                        int dtype, dvt_stream_t stream) {
  DVT_REQUIRE(dy && y && inv_norm && dx && rows >= 0 && D > 0 && eps > 0.f, "dvt_l2norm_rows_bwd: bad arguments");
  if (rows == 0) return DVT_OK;
  DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((l2norm_bwd_kernel<T>), dim3((unsigned)dvt_cdiv(rows, 4)), dim3(256), 0,
                                                  (hipStream_t)stream, (const T*)dy, (const T*)y, inv_norm, (T*)dx, rows,
                                                  D, eps));
  DVT_LAUNCH_CHECK("dvt_l2norm_rows_bwd");
  return DVT_OK;
}

int dvt_cosine_rows(const void* a, const void* b, float* out, int64_t rows, int D, float eps, int dtype,
                    dvt_stream_t stream) {
  DVT_REQUIRE(rows >= 0 && D > 0 && eps > 0.f && (rows == 0 || (a && b && out)), "dvt_cosine_rows: bad arguments");
  if (rows == 0) return DVT_OK;
  DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((cosine_rows_kernel<T>), dim3((unsigned)dvt_cdiv(rows, 4)), dim3(256), 0,
                                                  (hipStream_t)stream, (const T*)a, (const T*)b, out, rows, D, eps));
  DVT_LAUNCH_CHECK("dvt_cosine_rows");
  return DVT_OK;
}

int dvt_gate_fwd(const void* a, const void* b, void* y, int64_t n, int dtype, dvt_stream_t stream) {
  DVT_REQUIRE(a && b && y && n >= 0, "dvt_gate_fwd: bad arguments");
  if (n == 0) return DVT_OK;
  DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((gate_fwd_kernel<T>), dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream,
                                                  (const T*)a, (const T*)b, (T*)y, n));
  DVT_LAUNCH_CHECK("dvt_gate_fwd");
  return DVT_OK;
}

int dvt_gate_bwd(const void* dy, const void* a, const void* b, void* da, void* db, int64_t n, int dtype,
                 dvt_stream_t stream) {
  DVT_REQUIRE(dy && a && b && da && db && n >= 0, "dvt_gate_bwd: bad arguments");
  if (n == 0) return DVT_OK;
  DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((gate_bwd_kernel<T>), dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream,
                                                  (const T*)dy, (const T*)a, (const T*)b, (T*)da, (T*)db, n));
  DVT_LAUNCH_CHECK("dvt_gate_bwd");
  return DVT_OK;
}

int dvt_contrastive_fwd(const float* sim, int M, float temperature, float* loss, float* row_lse, float* row_loss,
                        dvt_stream_t stream) {
  DVT_REQUIRE(sim && loss && row_lse && row_loss && M >= 2 && M % 2 == 0 && temperature > 0.f,
              "dvt_contrastive_fwd: bad arguments (M = 2 * batch, temperature > 0)");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(contrastive_rows_fwd_kernel, dim3((unsigned)dvt_cdiv(M, 4)), dim3(256), 0, st, sim, M, M / 2,
                     1.0f / temperature, row_loss, row_lse);
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, (const float*)row_loss, M, loss);
  DVT_LAUNCH_CHECK("dvt_contrastive_fwd");
  return DVT_OK;
}

int dvt_contrastive_bwd(const float* sim, const float* row_lse, int M, float temperature, const float* gloss,
                        float* dsim, dvt_stream_t stream) {
  DVT_REQUIRE(sim && row_lse && gloss && dsim && M >= 2 && M % 2 == 0 && temperature > 0.f,
              "dvt_contrastive_bwd: bad arguments");
  hipLaunchKernelGGL(contrastive_rows_bwd_kernel, dim3(egrid((int64_t)M * M)), dim3(256), 0, (hipStream_t)stream, sim,
                     row_lse, M, M / 2, 1.0f / temperature, gloss, 1.0f / (float)M, dsim);
  DVT_LAUNCH_CHECK("dvt_contrastive_bwd");
  return DVT_OK;
}

}  // extern "C"
