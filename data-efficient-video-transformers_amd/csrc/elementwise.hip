// elementwise.hip -- HBM-bound data-movement kernels of the clip path:
// casts, residual add, patchify gather, token assembly (CLS + positional add),
// CLS row gather, losses, AdamW.  All are streaming kernels: 16-byte accesses
// per lane, grid-stride over at most 256 CUs x 8 blocks.
#include "common.h"
#include <type_traits>

namespace {

constexpr int kBlock = 256;

inline int grid_for(int64_t work_items) {
  int64_t blocks = dvt_cdiv(work_items, kBlock);
  int64_t cap = (int64_t)dvt_num_cus() * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

__device__ __forceinline__ float block_sum(float v, float* smem /* >= 4 floats */) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smem[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += smem[i];
  return t;
}

// ------------------------------------------------------------------ cast / add / axpby
template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, int64_t n) {
  const int64_t nvec = n >> 3;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    float v[8];
    load8<S>(src + i * 8, v);
    store8<D>(dst + i * 8, v);
  }
  if (blockIdx.x == 0) {  // tail (n % 8)
    const int64_t t = (nvec << 3) + threadIdx.x;
    if (t < n) dst[t] = from_f32<D>(to_f32<S>(src[t]));
  }
}

template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out,
                           int64_t n) {
  const int64_t nvec = n >> 3;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    float x[8], y[8];
    load8<T>(a + i * 8, x);
    load8<T>(b + i * 8, y);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] += y[j];
    store8<T>(out + i * 8, x);
  }
  if (blockIdx.x == 0) {
    const int64_t t = (nvec << 3) + threadIdx.x;
    if (t < n) out[t] = from_f32<T>(to_f32<T>(a[t]) + to_f32<T>(b[t]));
  }
}

template <typename T>
__global__ void add_rowtable_kernel(const T* __restrict__ x, const float* __restrict__ table,
                                    T* __restrict__ out, int64_t rows, int64_t d, int64_t rpe) {
  const int64_t dv = d >> 3;
  const int64_t items = rows * dv;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const int64_t c = (it % dv) << 3, r = it / dv;
    float v[8], t[8];
    load8<T>(x + r * d + c, v);
    load8<float>(table + (r / rpe) * d + c, t);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] += t[k];
    store8<T>(out + r * d + c, v);
  }
}

template <typename T>
__global__ void copy2d_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t rows, int64_t cols,
                              int64_t src_ld, int64_t dst_ld, int vec) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (vec) {
    const int64_t cv = cols >> 3;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < rows * cv; it += stride) {
      const int64_t c = (it % cv) << 3, r = it / cv;
      float v[8];
      load8<T>(src + r * src_ld + c, v);
      store8<T>(dst + r * dst_ld + c, v);
    }
  } else {
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < rows * cols; it += stride) {
      const int64_t c = it % cols, r = it / cols;
      dst[r * dst_ld + c] = src[r * src_ld + c];
    }
  }
}

template <typename T>
__global__ void permute021_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t A, int64_t B,
                                  int64_t Cc) {
  const int64_t cv = Cc >> 3;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < A * B * cv; it += stride) {
    const int64_t c = (it % cv) << 3;
    const int64_t ab = it / cv;           // destination row index (b, a)
    const int64_t a = ab % A, b = ab / A;
    float v[8];
    load8<T>(src + (a * B + b) * Cc + c, v);
    store8<T>(dst + ab * Cc + c, v);
  }
}

template <typename S>
__global__ void axpby_kernel(const S* __restrict__ src, float alpha, float* __restrict__ dst,
                             float beta, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float s = alpha * to_f32<S>(src[i]);
    dst[i] = beta == 0.0f ? s : fmaf(beta, dst[i], s);
  }
}

template <typename T, bool FWD>
__global__ void act_kernel(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ out,
                           int64_t n, int act) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = to_f32<T>(x[i]);
    float r;
    if (act == 3) {                       // sigmoid (TPN.py:99)
      const float sg = 1.0f / (1.0f + __expf(-v));
      r = FWD ? sg : to_f32<T>(dy[i]) * sg * (1.0f - sg);
    } else if (FWD) r = act == 1 ? gelu_erf_f(v) : fmaxf(v, 0.f);
    else r = to_f32<T>(dy[i]) * (act == 1 ? gelu_erf_grad_f(v) : (v > 0.f ? 1.f : 0.f));
    out[i] = from_f32<T>(r);
  }
}

// ------------------------------------------------------------------ patchify
// out[(f*nh + ph)*nw + pw, (p1*P + p2)*C + c] = x[f, c, ph*P + p1, pw*P + p2]
// LDS operations of one wave complete in order; this only keeps the compiler from moving them across the point
__device__ __forceinline__ void wave_lds_fence_ew() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// Work item = one patch row (frame, ph, p1, pw): C runs of P contiguous pixels in,
// one run of P*C contiguous patch-vector elements out.  pw is the fastest index so
// a wave reads 64 adjacent pixel runs (a contiguous span of the image row per channel).
template <int P, int C, typename S, typename D, bool FWD>
__global__ void patchify_vec_kernel(const S* __restrict__ x, D* __restrict__ out, S* __restrict__ dx,
                                    const D* __restrict__ dout, int64_t frames, int H, int W) {
  const int nh = H / P, nw = W / P;
  const int64_t items = frames * nh * P * nw;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  constexpr int64_t pd = (int64_t)P * P * C;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const int pw = (int)(it % nw);
    int64_t r = it / nw;
    const int p1 = (int)(r % P);
    r /= P;
    const int ph = (int)(r % nh);
    const int64_t f = r / nh;
    const int64_t orow = (f * nh + ph) * nw + pw;
    const int64_t ooff = orow * pd + (int64_t)p1 * P * C;
    const int64_t xoff = ((f * C) * H + (int64_t)ph * P + p1) * W + (int64_t)pw * P;
    float pix[C][P];
    float vec[P * C];
    if (FWD) {
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int q = 0; q < P / 8; ++q) {
          float t[8];
          load8<S>(x + xoff + (int64_t)c * H * W + q * 8, t);
#pragma unroll
          for (int k = 0; k < 8; ++k) pix[c][q * 8 + k] = t[k];
        }
#pragma unroll
      for (int p2 = 0; p2 < P; ++p2)
#pragma unroll
        for (int c = 0; c < C; ++c) vec[p2 * C + c] = pix[c][p2];
#pragma unroll
      for (int q = 0; q < P * C / 8; ++q) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = vec[q * 8 + k];
        store8<D>(out + ooff + q * 8, t);
      }
    } else {
#pragma unroll
      for (int q = 0; q < P * C / 8; ++q) {
        float t[8];
        load8<D>(dout + ooff + q * 8, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) vec[q * 8 + k] = t[k];
      }
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int q = 0; q < P / 8; ++q) {
          float t[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) t[k] = vec[(q * 8 + k) * C + c];
          store8<S>(dx + xoff + (int64_t)c * H * W + q * 8, t);
        }
    }
  }
}

// Forward, 16 x 16 patches of 3 channels, 16-bit patch vectors (the metric shape): lane = (patch & 3) * 16 + p1, so a wave
// owns FOUR consecutive patch vectors = 6 KiB of contiguous output.  The lane's 96 output bytes (one patch row: 16 pixels
// x 3 channels) go through a wave-private LDS patch and leave as six fully contiguous 1-KiB store instructions; the
// per-lane form above stores 16-byte pieces at a 1,536-byte stride (one 64-byte segment per piece: 3.3 TB/s).
template <typename S, typename D>
__global__ __launch_bounds__(256) void patchify16_fwd_kernel(const S* __restrict__ x, D* __restrict__ out, int64_t npatches,
                                                             int H, int W, int nh, int nw) {
  constexpr int P = 16, C = 3, ROWB = P * C * 2;              // 96 bytes per patch row
  __shared__ __attribute__((aligned(16))) char lds[4][64 * ROWB];
  typedef D v8 __attribute__((ext_vector_type(8)));
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t ngroups = (npatches + 3) >> 2, nwaves = (int64_t)gridDim.x * 4;
  char* mine = lds[wid];
  for (int64_t g = (int64_t)blockIdx.x * 4 + wid; g < ngroups; g += nwaves) {
    const int64_t patch = min(g * 4 + (lane >> 4), npatches - 1);
    const int p1 = lane & 15;
    const int pw = (int)(patch % nw);
    const int64_t r = patch / nw;
    const int ph = (int)(r % nh);
    const int64_t f = r / nh;
    const int64_t xoff = ((f * C) * H + (int64_t)ph * P + p1) * W + (int64_t)pw * P;
    float pix[C][P];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int q = 0; q < P / 8; ++q) {
        float t[8];
        load8<S>(x + xoff + (int64_t)c * H * W + q * 8, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) pix[c][q * 8 + k] = t[k];
      }
#pragma unroll
    for (int q = 0; q < P * C / 8; ++q) {
      v8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int e = q * 8 + k;                                // e = p2 * C + c
        o[k] = (D)pix[e % C][e / C];
      }
      *reinterpret_cast<v8*>(mine + lane * ROWB + q * 16) = o;
    }
    wave_lds_fence_ew();
    char* dst = reinterpret_cast<char*>(out) + g * 4 * (int64_t)(P * ROWB);
    const int64_t lim = (npatches - g * 4) * (int64_t)(P * ROWB);  // bytes of this group that exist
#pragma unroll
    for (int q = 0; q < P * C / 8; ++q) {
      const int off = (q * 64 + lane) * 16;
      const v8 o = *reinterpret_cast<const v8*>(mine + off);
      if (off < lim) *reinterpret_cast<v8*>(dst + off) = o;
    }
    wave_lds_fence_ew();
  }
}

// Any P, C, alignment: scalar accesses.
template <typename S, typename D, bool FWD>
__global__ void patchify_generic_kernel(const S* __restrict__ x, D* __restrict__ out,
                                        S* __restrict__ dx, const D* __restrict__ dout,
                                        int64_t frames, int C, int H, int W, int P) {
  const int nh = H / P, nw = W / P;
  const int64_t pd = (int64_t)P * P * C;
  const int64_t total = frames * nh * nw * pd;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t e = i % pd;
    const int64_t orow = i / pd;
    const int c = (int)(e % C);
    const int p2 = (int)((e / C) % P);
    const int p1 = (int)(e / ((int64_t)C * P));
    const int pw = (int)(orow % nw);
    const int ph = (int)((orow / nw) % nh);
    const int64_t f = orow / ((int64_t)nw * nh);
    const int64_t xi = ((f * C + c) * H + (int64_t)ph * P + p1) * W + (int64_t)pw * P + p2;
    if (FWD) out[i] = from_f32<D>(to_f32<S>(x[xi]));
    else dx[xi] = from_f32<S>(to_f32<D>(dout[i]));
  }
}

// ------------------------------------------------------------------ token assembly
template <typename T>
__global__ void tokens_fwd_kernel(const T* __restrict__ emb, const float* __restrict__ cls,
                                  const float* __restrict__ pos, T* __restrict__ out, int64_t S,
                                  int64_t Tn, int64_t n, int64_t d, int64_t pos_rows) {
  const int64_t dv = d >> 3;
  const int64_t items = S * (n + 1) * dv;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const int64_t c = (it % dv) << 3;
    const int64_t row = it / dv;
    const int64_t j = row % (n + 1);
    const int64_t s = row / (n + 1);
    float v[8], p[8];
    load8<float>(pos + ((s % Tn) * pos_rows + j) * d + c, p);
    if (j == 0) load8<float>(cls + c, v);
    else load8<T>(emb + (s * n + (j - 1)) * d + c, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] += p[k];
    store8<T>(out + row * d + c, v);
  }
}

template <typename T>
__global__ void tokens_bwd_emb_kernel(const T* __restrict__ dout, T* __restrict__ demb, int64_t S,
                                      int64_t n, int64_t d) {
  const int64_t dv = d >> 3;
  const int64_t items = S * n * dv;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const int64_t c = (it % dv) << 3;
    const int64_t row = it / dv;  // = s*n + j
    const int64_t j = row % n;
    const int64_t s = row / n;
    float v[8];
    load8<T>(dout + (s * (n + 1) + j + 1) * d + c, v);
    store8<T>(demb + row * d + c, v);
  }
}

// dpos[t, j, :] (+)= sum_b dout[b*T + t, j, :]   (rows j > n get 0 when overwriting)
// demb != nullptr: the same pass also writes the patch-row gradients demb[(b*T + t)*n + j - 1, :] = dout[b*T + t, j, :]
// (j >= 1): one read of dout serves both (tokens_bwd_emb_kernel alone re-read it).
template <typename T>
__global__ void tokens_bwd_pos_kernel(const T* __restrict__ dout, float* __restrict__ dpos,
                                      int64_t S, int64_t Tn, int64_t n, int64_t d, int64_t pos_rows,
                                      int accumulate, T* __restrict__ demb = nullptr) {
  const int64_t dv = d >> 3;
  const int64_t items = Tn * pos_rows * dv;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t B = S / Tn;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const int64_t c = (it % dv) << 3;
    const int64_t row = it / dv;
    const int64_t j = row % pos_rows;
    const int64_t t = row / pos_rows;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (j <= n) {
      for (int64_t b = 0; b < B; ++b) {
        float v[8];
        load8<T>(dout + ((b * Tn + t) * (n + 1) + j) * d + c, v);
        if (demb && j >= 1) store8<T>(demb + ((b * Tn + t) * n + j - 1) * d + c, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += v[k];
      }
    }
    float* o = dpos + row * d + c;
    if (accumulate) {
      float p[8];
      load8<float>(o, p);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += p[k];
    }
    store8<float>(o, acc);
  }
}

// out[c] (+)= sum_{r < R} src[r * row_stride + c]; one block per 8-column chunk.
template <typename T>
__device__ __forceinline__ void strided_rows_sum_block(const T* __restrict__ src, int64_t row_stride, int64_t R,
                                                       float* __restrict__ out, int accumulate, int64_t blk) {
  __shared__ float red[8][kBlock / 64];
  const int64_t c = blk << 3;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t r = threadIdx.x; r < R; r += blockDim.x) {
    float v[8];
    load8<T>(src + r * row_stride + c, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += v[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float s = wave_sum(acc[k]);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    float t = 0.f;
    for (int i = 0; i < kBlock / 64; ++i) t += red[threadIdx.x][i];
    float* o = out + c + threadIdx.x;
    *o = accumulate ? *o + t : t;
  }
}

template <typename T>
__global__ void strided_rows_sum_kernel(const T* __restrict__ src, int64_t row_stride, int64_t R,
                                        float* __restrict__ out, int accumulate) {
  strided_rows_sum_block<T>(src, row_stride, R, out, accumulate, blockIdx.x);
}

// The same sum for FEW rows of MANY columns (the learnable pixel-space CLS chunk of FrameTransformer: B = 2 rows of 451 k
// values, frame_transformer.py:105,195): a thread per 8-column chunk walks the rows -- the block-per-chunk form above spent a
// 256-thread workgroup and two barriers on each 32 bytes (56,448 workgroups, 119 us).
template <typename T>
__global__ void few_rows_sum_kernel(const T* __restrict__ src, int64_t row_stride, int64_t R, int64_t chunks,
                                    float* __restrict__ out, int accumulate) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t ch = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ch < chunks; ch += stride) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t r = 0; r < R; ++r) {
      float v[8];
      load8<T>(src + r * row_stride + (ch << 3), v);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += v[k];
    }
    if (accumulate) {
      float o[8];
      load8<float>(out + (ch << 3), o);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += o[k];
    }
    store8<float>(out + (ch << 3), acc);
  }
}

// ------------------------------------------------------------------ CLS row gather
template <typename T>
__global__ void rows_gather_fwd_kernel(const T* __restrict__ src, int64_t src_row_stride,
                                       const float* __restrict__ tok, T* __restrict__ out, int64_t B,
                                       int64_t Tn, int64_t d, int lead) {
  const int64_t dv = d >> 3;
  const int64_t items = B * (Tn + lead) * dv;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const int64_t c = (it % dv) << 3;
    const int64_t row = it / dv;
    const int64_t j = row % (Tn + lead);
    const int64_t b = row / (Tn + lead);
    float v[8];
    if (lead && j == 0) load8<float>(tok + c, v);
    else load8<T>(src + (b * Tn + (j - lead)) * src_row_stride + c, v);
    store8<T>(out + row * d + c, v);
  }
}

// The last `tok_blocks` workgroups of the grid sum the token rows (row 0 of every sequence) into dtok instead -- the token
// parameter's gradient in the same launch as the scatter of the other rows.
template <typename T>
__global__ void rows_gather_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dsrc,
                                       int64_t src_row_stride, int64_t B, int64_t Tn, int64_t d,
                                       int lead, float* __restrict__ dtok, int accumulate, int tok_blocks) {
  const int64_t dv = d >> 3;
  const int64_t items = B * Tn * dv;
  const int gather_blocks = (int)gridDim.x - tok_blocks;
  if ((int)blockIdx.x >= gather_blocks) {
    strided_rows_sum_block<T>(dout, (Tn + 1) * d, B, dtok, accumulate, (int)blockIdx.x - gather_blocks);
    return;
  }
  const int64_t stride = (int64_t)gather_blocks * blockDim.x;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const int64_t c = (it % dv) << 3;
    const int64_t row = it / dv;  // b*T + t
    const int64_t t = row % Tn;
    const int64_t b = row / Tn;
    float v[8];
    load8<T>(dout + (b * (Tn + lead) + t + lead) * d + c, v);
    store8<T>(dsrc + row * src_row_stride + c, v);
  }
}

// ------------------------------------------------------------------ mean over rows
template <typename T, bool FWD>
__global__ void mean_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t B, int64_t Ln,
                                 int64_t d, float inv) {
  const int64_t dv = d >> 3;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (FWD) {
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < B * dv; it += stride) {
      const int64_t c = (it % dv) << 3, b = it / dv;
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      // sixteen rows requested before the first is added (same order of additions): a rolled loop waited for every row's
      // load in turn -- 98 round trips = 31.6 us for the 2.8 MB pooling in front of the frametransformer's encoder
      int64_t j = 0;
      for (; j + 16 <= Ln; j += 16) {
        float v[16][8];
#pragma unroll
        for (int u = 0; u < 16; ++u) load8<T>(src + (b * Ln + j + u) * d + c, v[u]);
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[k] += v[u][k];
      }
      for (; j < Ln; ++j) {
        float v[8];
        load8<T>(src + (b * Ln + j) * d + c, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += v[k];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] *= inv;
      store8<T>(dst + b * d + c, acc);
    }
  } else {
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < B * Ln * dv; it += stride) {
      const int64_t c = (it % dv) << 3, row = it / dv, b = row / Ln;
      float v[8];
      load8<T>(src + b * d + c, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] *= inv;
      store8<T>(dst + row * d + c, v);
    }
  }
}

// ------------------------------------------------------------------ losses
template <typename T>
__global__ void bce_fwd_kernel(const T* __restrict__ z, const float* __restrict__ y,
                               float* __restrict__ loss, int64_t n) {
  __shared__ float red[kBlock / 64];
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    const float x = to_f32<T>(z[i]);
    acc += fmaxf(x, 0.f) - x * y[i] + log1pf(expf(-fabsf(x)));
  }
  const float t = block_sum(acc, red);
  if (threadIdx.x == 0) loss[0] = t / (float)n;
}

template <typename T>
__global__ void bce_bwd_kernel(const T* __restrict__ z, const float* __restrict__ y,
                               const float* __restrict__ gloss, T* __restrict__ dz, int64_t n) {
  const float g = gloss[0] / (float)n;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float x = to_f32<T>(z[i]);
    const float s = 1.0f / (1.0f + expf(-x));
    dz[i] = from_f32<T>(g * (s - y[i]));
  }
}

// One wave per row: lse(student) - student[argmax teacher]; first max wins (torch.argmax).
template <typename T, bool FWD>
__global__ void ce_argmax_kernel(const T* __restrict__ st, const T* __restrict__ te,
                                 const float* __restrict__ gloss, float* __restrict__ loss,
                                 T* __restrict__ dst, int64_t rows, int64_t C) {
  __shared__ float red[kBlock / 64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  float total = 0.f;
  for (int64_t r = wave; r < rows; r += nw) {
    const T* s = st + r * C;
    const T* t = te + r * C;
    float m = -INFINITY, tm = -INFINITY;
    int64_t ti = C;
    for (int64_t c = lane; c < C; c += 64) {
      m = fmaxf(m, to_f32<T>(s[c]));
      const float tv = to_f32<T>(t[c]);
      if (tv > tm) { tm = tv; ti = c; }
    }
    m = wave_max(m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(tm, o, 64);
      const int64_t oi = __shfl_xor((long long)ti, o, 64);
      if (om > tm || (om == tm && oi < ti)) { tm = om; ti = oi; }
    }
    float se = 0.f;
    for (int64_t c = lane; c < C; c += 64) se += expf(to_f32<T>(s[c]) - m);
    se = wave_sum(se);
    if (FWD) {
      if (lane == 0) total += m + logf(se) - to_f32<T>(s[ti]);
    } else {
      const float g = gloss[0] / (float)rows;
      for (int64_t c = lane; c < C; c += 64) {
        const float p = expf(to_f32<T>(s[c]) - m) / se;
        dst[r * C + c] = from_f32<T>(g * (p - (c == ti ? 1.f : 0.f)));
      }
    }
  }
  if (FWD) {
    const float t = block_sum(total, red);
    if (threadIdx.x == 0) loss[0] = t / (float)rows;
  }
}

// ------------------------------------------------------------------ AdamW
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                             float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                             float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float step_size = lr / bc1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gi = g[i];
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = fmaf(b1, m[i], (1.0f - b1) * gi);
    const float vi = fmaf(b2, v[i], (1.0f - b2) * gi * gi);
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

}  // namespace

extern "C" {

int dvt_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n,
             dvt_stream_t stream) {
  if (n == 0) return DVT_OK;   // empty tensors carry null pointers: nothing to validate, nothing to launch
  DVT_REQUIRE(src && dst && n >= 0, "dvt_cast: null pointer or negative size");
  DVT_REQUIRE(dvt_aligned16(src) && dvt_aligned16(dst), "dvt_cast: buffers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int g = grid_for((n >> 3) + 1);
#define DVT_CAST_CASE(SD, S, DD, D)                                                        \
  if (src_dtype == SD && dst_dtype == DD) {                                                \
    hipLaunchKernelGGL((cast_kernel<S, D>), dim3(g), dim3(kBlock), 0, st, (const S*)src, (D*)dst, n); \
    DVT_LAUNCH_CHECK("dvt_cast");                                                          \
    return DVT_OK;                                                                         \
  }
  DVT_CAST_CASE(DVT_F32, float, DVT_BF16, bf16)
  DVT_CAST_CASE(DVT_F32, float, DVT_F16, f16)
  DVT_CAST_CASE(DVT_BF16, bf16, DVT_F32, float)
  DVT_CAST_CASE(DVT_F16, f16, DVT_F32, float)
  DVT_CAST_CASE(DVT_F32, float, DVT_F32, float)
  DVT_CAST_CASE(DVT_BF16, bf16, DVT_BF16, bf16)
  DVT_CAST_CASE(DVT_F16, f16, DVT_F16, f16)
#undef DVT_CAST_CASE
  DVT_UNSUPPORTED("dvt_cast: dtype pair (%d -> %d) not supported", src_dtype, dst_dtype);
}

int dvt_add(const void* a, const void* b, void* out, int64_t n, int dtype, dvt_stream_t stream) {
  if (n == 0) return DVT_OK;   // empty tensors carry null pointers: nothing to validate, nothing to launch
  DVT_REQUIRE(a && b && out && n >= 0, "dvt_add: null pointer or negative size");
  DVT_REQUIRE(dvt_aligned16(a) && dvt_aligned16(b) && dvt_aligned16(out),
              "dvt_add: buffers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  DVT_DISPATCH_DTYPE(dtype, T,
                     hipLaunchKernelGGL((add_kernel<T>), dim3(grid_for((n >> 3) + 1)), dim3(kBlock),
                                        0, st, (const T*)a, (const T*)b, (T*)out, n));
  DVT_LAUNCH_CHECK("dvt_add");
  return DVT_OK;
}

int dvt_act_fwd(const void* x, void* y, int64_t n, int act, int dtype, dvt_stream_t stream) {
  if (n == 0) return DVT_OK;   // empty tensors carry null pointers: nothing to validate, nothing to launch
  DVT_REQUIRE(x && y && n >= 0 && act >= 1 && act <= 3, "dvt_act_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  DVT_DISPATCH_DTYPE(dtype, T,
                     hipLaunchKernelGGL((act_kernel<T, true>), dim3(grid_for(n)), dim3(kBlock), 0, st,
                                        (const T*)nullptr, (const T*)x, (T*)y, n, act));
  DVT_LAUNCH_CHECK("dvt_act_fwd");
  return DVT_OK;
}

int dvt_act_bwd(const void* dy, const void* x, void* dx, int64_t n, int act, int dtype,
                dvt_stream_t stream) {
  if (n == 0) return DVT_OK;   // empty tensors carry null pointers: nothing to validate, nothing to launch
  DVT_REQUIRE(dy && x && dx && n >= 0 && act >= 1 && act <= 3, "dvt_act_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  DVT_DISPATCH_DTYPE(dtype, T,
                     hipLaunchKernelGGL((act_kernel<T, false>), dim3(grid_for(n)), dim3(kBlock), 0, st,
                                        (const T*)dy, (const T*)x, (T*)dx, n, act));
  DVT_LAUNCH_CHECK("dvt_act_bwd");
  return DVT_OK;
}

int dvt_add_rowtable(const void* x, const float* table, void* out, int64_t rows, int64_t d,
                     int64_t rows_per_entry, int dtype, dvt_stream_t stream) {
  DVT_REQUIRE(x && table && out && rows >= 0 && d > 0 && rows_per_entry > 0, "dvt_add_rowtable: bad arguments");
  DVT_REQUIRE(d % 8 == 0, "dvt_add_rowtable: d must be a multiple of 8");
  DVT_REQUIRE(dvt_aligned16(x) && dvt_aligned16(table) && dvt_aligned16(out), "dvt_add_rowtable: misaligned buffer");
  if (rows == 0) return DVT_OK;
  hipStream_t st = (hipStream_t)stream;
  DVT_DISPATCH_DTYPE(dtype, T,
                     hipLaunchKernelGGL((add_rowtable_kernel<T>), dim3(grid_for(rows * (d >> 3))), dim3(kBlock),
                                        0, st, (const T*)x, table, (T*)out, rows, d, rows_per_entry));
  DVT_LAUNCH_CHECK("dvt_add_rowtable");
  return DVT_OK;
}

int dvt_copy2d(const void* src, void* dst, int64_t rows, int64_t cols, int64_t src_ld, int64_t dst_ld,
               int dtype, dvt_stream_t stream) {
  DVT_REQUIRE(src && dst && rows >= 0 && cols >= 0 && src_ld >= 0 && dst_ld >= cols, "dvt_copy2d: bad arguments");
  if (rows == 0 || cols == 0) return DVT_OK;
  hipStream_t st = (hipStream_t)stream;
  const int vec = (cols % 8 == 0) && (src_ld % 8 == 0) && (dst_ld % 8 == 0) && dvt_aligned16(src) && dvt_aligned16(dst);
  DVT_DISPATCH_DTYPE(dtype, T,
                     hipLaunchKernelGGL((copy2d_kernel<T>), dim3(grid_for(rows * (vec ? cols >> 3 : cols))),
                                        dim3(kBlock), 0, st, (const T*)src, (T*)dst, rows, cols, src_ld, dst_ld, vec));
  DVT_LAUNCH_CHECK("dvt_copy2d");
  return DVT_OK;
}

int dvt_rows_sum(const void* src, int64_t row_stride, int64_t rows, int64_t cols, float* out, int dtype,
                 int accumulate, dvt_stream_t stream) {
  DVT_REQUIRE(src && out && rows >= 0 && cols > 0, "dvt_rows_sum: bad arguments");
  DVT_REQUIRE(cols % 8 == 0 && row_stride % 8 == 0 && dvt_aligned16(src) && dvt_aligned16(out),
              "dvt_rows_sum: cols / row stride must be multiples of 8 and buffers 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (rows <= 16 && cols >= 8192) {
    DVT_DISPATCH_DTYPE(dtype, T,
                       hipLaunchKernelGGL((few_rows_sum_kernel<T>), dim3(grid_for(cols >> 3)), dim3(kBlock), 0, st,
                                          (const T*)src, row_stride, rows, cols >> 3, out, accumulate));
  } else {
    DVT_DISPATCH_DTYPE(dtype, T,
                       hipLaunchKernelGGL((strided_rows_sum_kernel<T>), dim3((unsigned)(cols >> 3)), dim3(kBlock), 0,
                                          st, (const T*)src, row_stride, rows, out, accumulate));
  }
  DVT_LAUNCH_CHECK("dvt_rows_sum");
  return DVT_OK;
}

int dvt_permute_021(const void* src, void* dst, int64_t A, int64_t B, int64_t C, int dtype,
                    dvt_stream_t stream) {
  DVT_REQUIRE(src && dst && A >= 0 && B >= 0 && C > 0 && C % 8 == 0, "dvt_permute_021: bad arguments (C % 8 == 0)");
  DVT_REQUIRE(dvt_aligned16(src) && dvt_aligned16(dst), "dvt_permute_021: misaligned buffer");
  if (A == 0 || B == 0) return DVT_OK;
  hipStream_t st = (hipStream_t)stream;
  DVT_DISPATCH_DTYPE(dtype, T,
                     hipLaunchKernelGGL((permute021_kernel<T>), dim3(grid_for(A * B * (C >> 3))), dim3(kBlock), 0,
                                        st, (const T*)src, (T*)dst, A, B, C));
  DVT_LAUNCH_CHECK("dvt_permute_021");
  return DVT_OK;
}

int dvt_axpby_f32(const void* src, int src_dtype, float alpha, float* dst, float beta, int64_t n,
                  dvt_stream_t stream) {
  if (n == 0) return DVT_OK;   // empty tensors carry null pointers: nothing to validate, nothing to launch
  DVT_REQUIRE(src && dst && n >= 0, "dvt_axpby_f32: null pointer or negative size");
  hipStream_t st = (hipStream_t)stream;
  DVT_DISPATCH_DTYPE(src_dtype, T,
                     hipLaunchKernelGGL((axpby_kernel<T>), dim3(grid_for(n)), dim3(kBlock), 0, st,
                                        (const T*)src, alpha, dst, beta, n));
  DVT_LAUNCH_CHECK("dvt_axpby_f32");
  return DVT_OK;
}

}  // extern "C"

namespace {
__global__ void adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                 float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                 float b1, float b2, float eps, float wd,
                                 const int64_t* __restrict__ step_dev, const uint8_t* __restrict__ skip) {
  const float t = (float)(step_dev[0] + 1);
  const float bc1 = 1.0f - powf(b1, t);
  const float bc2_sqrt = sqrtf(1.0f - powf(b2, t));
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float step_size = lr / bc1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (skip && skip[i >> 6]) continue;               // parameter without a gradient this step: untouched (torch: grad None)
    const float gi = g[i];
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = fmaf(b1, m[i], (1.0f - b1) * gi);
    const float vi = fmaf(b2, v[i], (1.0f - b2) * gi * gi);
    pi -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

__global__ void inc_step_kernel(int64_t* step_dev) { step_dev[0] += 1; }

// The flat-buffer step of the training loop in ONE launch: AdamW on four elements per thread (16-byte accesses), the
// 16-bit mirror of the updated weights that the next step's GEMMs read (M = bf16 / f16; float: no mirror), and the step
// counter: every block reads step_dev[0] before it takes a ticket in step_dev[1], the last ticket stores the increment.
template <typename M>
__global__ __launch_bounds__(256) void adamw_fused_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                          float b1, float b2, float eps, float wd, int64_t* step_dev,
                                                          const uint8_t* __restrict__ skip, M* __restrict__ mirror) {
  const int64_t steps = step_dev[0];
  const float t = (float)(steps + 1);
  const float bc1 = 1.0f - powf(b1, t);
  const float bc2_sqrt = sqrtf(1.0f - powf(b2, t));
  const float step_size = lr / bc1, decay = 1.0f - lr * wd;
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int64_t e = i << 2;
    f32x4 pv = *reinterpret_cast<const f32x4*>(p + e);
    if (!(skip && skip[e >> 6])) {                     // parameter without a gradient this step: untouched (torch: grad None)
      const f32x4 gv = *reinterpret_cast<const f32x4*>(g + e);
      f32x4 mv = *reinterpret_cast<const f32x4*>(m + e), vv = *reinterpret_cast<const f32x4*>(v + e);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float pi = pv[k] * decay;
        mv[k] = fmaf(b1, mv[k], (1.0f - b1) * gv[k]);
        vv[k] = fmaf(b2, vv[k], (1.0f - b2) * gv[k] * gv[k]);
        pi -= step_size * (mv[k] / (sqrtf(vv[k]) / bc2_sqrt + eps));
        pv[k] = pi;
      }
      *reinterpret_cast<f32x4*>(p + e) = pv;
      *reinterpret_cast<f32x4*>(m + e) = mv;
      *reinterpret_cast<f32x4*>(v + e) = vv;
    }
    if (!std::is_same<M, float>::value) {
      typedef M m4 __attribute__((ext_vector_type(4)));
      m4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = (M)pv[k];
      *reinterpret_cast<m4*>(mirror + e) = o;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {       // tail of a buffer whose length is not a multiple of 4
    const int64_t i = (n4 << 2) + threadIdx.x;
    float pi = p[i];
    if (!(skip && skip[i >> 6])) {
      const float gi = g[i];
      pi *= decay;
      const float mi = fmaf(b1, m[i], (1.0f - b1) * gi);
      const float vi = fmaf(b2, v[i], (1.0f - b2) * gi * gi);
      pi -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
      p[i] = pi; m[i] = mi; v[i] = vi;
    }
    if (!std::is_same<M, float>::value) mirror[i] = (M)pi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // every thread of this block has read step_dev[0]; relaxed device-scope ticket: the last block publishes the increment
    const unsigned long long ticket = __hip_atomic_fetch_add((unsigned long long*)(step_dev + 1), 1ull, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT);
    if (ticket == (unsigned long long)gridDim.x - 1) {
      step_dev[0] = steps + 1;
      step_dev[1] = 0;
    }
  }
}

// ---- dropout (nn.Dropout in training mode: frame_transformer.py:22,41-44; TPN.py:92,95; vit.py:23,25,43,104)
// Counter-based Philox4x32-10: element i draws word (i & 3) of block (offset + i / 4) under the key (seed).  No mask
// is stored: backward re-draws the same words.  state[0] = seed, state[1] = per-step base offset live on the device
// (advanced by dvt_rng_advance once per step), so a captured hipGraph draws fresh masks on every replay.
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, uint32_t threshold, float scale,
                               const uint64_t* __restrict__ state, uint64_t call_offset) {
  const uint64_t seed = state[0], base = state[1] + call_offset;
  const int64_t nblk = (n + 3) >> 2;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t ctr = base + (uint64_t)b;
    uint32_t r[4];
    philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t i = b * 4 + e;
      if (i < n) y[i] = from_f32<T>(r[e] >= threshold ? to_f32<T>(x[i]) * scale : 0.f);
    }
  }
}

// The same mask around its neighbours (dvt_dropout_fused): y = residual? + keep * scale * relu?(x), or -- state == nullptr,
// the backward of the ReLU form -- y = gate != 0 ? scale * x : 0 with the forward's OUTPUT as gate (an element the mask
// dropped and one the ReLU zeroed both pass no gradient).
template <typename T>
__global__ void dropout_fused_kernel(const T* __restrict__ x, const T* __restrict__ res, const T* __restrict__ gate,
                                     T* __restrict__ y, int64_t n, uint32_t threshold, float scale,
                                     const uint64_t* __restrict__ state, uint64_t call_offset, int relu) {
  const uint64_t seed = state ? state[0] : 0, base = state ? state[1] + call_offset : 0;
  const int64_t nblk = (n + 3) >> 2;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += (int64_t)gridDim.x * blockDim.x) {
    uint32_t r[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (state) {
      const uint64_t ctr = base + (uint64_t)b;
      philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t i = b * 4 + e;
      if (i < n) {
        float v = to_f32<T>(x[i]);
        if (relu) v = fmaxf(v, 0.f);
        bool keep = r[e] >= threshold;
        if (gate) keep = to_f32<T>(gate[i]) != 0.f;
        v = keep ? v * scale : 0.f;
        if (res) v += to_f32<T>(res[i]);
        y[i] = from_f32<T>(v);
      }
    }
  }
}

__global__ void rng_advance_kernel(uint64_t* state, uint64_t delta) { state[1] += delta; }

// ---- fp16 loss scaling (BASELINE configs[4]: "fp16 + loss scaling"), all state on the device so the step stays
// hipGraph-capturable: scale[0], found_inf[0] (int32), good_steps[0] (int32), loss_grad[0] = scale * base.
__global__ void check_finite_kernel(const float* __restrict__ g, int64_t n, int* __restrict__ found_inf) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = g[i];
    bad |= !(fabsf(v) <= 3.402823466e38f);          // inf or nan
  }
  if (bad) atomicOr(found_inf, 1);
}

__global__ void adamw_scaled_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                    float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                                    const int64_t* __restrict__ step_dev, const float* __restrict__ scale,
                                    const int* __restrict__ found_inf, const uint8_t* __restrict__ skip) {
  if (found_inf[0]) return;                           // overflow: skip the whole update
  const float inv_scale = 1.0f / scale[0];
  const float t = (float)(step_dev[0] + 1);
  const float bc1 = 1.0f - powf(b1, t);
  const float bc2_sqrt = sqrtf(1.0f - powf(b2, t));
  const float step_size = lr / bc1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (skip && skip[i >> 6]) continue;
    const float gi = g[i] * inv_scale;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = fmaf(b1, m[i], (1.0f - b1) * gi);
    const float vi = fmaf(b2, v[i], (1.0f - b2) * gi * gi);
    pi -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

// after the step: step counter, dynamic scale (torch.cuda.amp.GradScaler rule), flag reset, next loss gradient
__global__ void loss_scale_update_kernel(int64_t* step_dev, float* scale, int* found_inf, int* good_steps,
                                         int growth_interval, float growth, float backoff, float* loss_grad,
                                         float base) {
  if (found_inf[0]) {
    scale[0] *= backoff;
    good_steps[0] = 0;
  } else {
    step_dev[0] += 1;
    if (++good_steps[0] >= growth_interval) { scale[0] *= growth; good_steps[0] = 0; }
  }
  found_inf[0] = 0;
  loss_grad[0] = scale[0] * base;
}

// torch.optim.SGD (dampening 0, no nesterov): d = g + wd p; buf = mu buf + d; p -= lr buf   (buf starts at 0,
// which reproduces torch's "first step: buf = d" rule).  mu == 0: plain p -= lr d, buf untouched.
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                           int64_t n, float lr, float mu, float wd, const uint8_t* __restrict__ skip) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (skip && skip[i >> 6]) continue;
    const float pi = p[i];
    float d = fmaf(wd, pi, g[i]);
    if (mu != 0.f) {
      d = fmaf(mu, buf[i], d);
      buf[i] = d;
    }
    p[i] = pi - lr * d;
  }
}

// torch.optim.Adagrad: d = g + wd p; sum += d^2; p -= clr d / (sqrt(sum) + eps)
__global__ void adagrad_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sum,
                               int64_t n, float clr, float eps, float wd, const uint8_t* __restrict__ skip) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (skip && skip[i >> 6]) continue;
    const float pi = p[i];
    const float d = fmaf(wd, pi, g[i]);
    const float si = fmaf(d, d, sum[i]);
    sum[i] = si;
    p[i] = pi - clr * (d / (sqrtf(si) + eps));
  }
}

template <typename S, typename D, bool FWD>
int patchify_dispatch(const void* x, void* out, void* dx, const void* dout, int64_t frames, int C,
                      int H, int W, int P, hipStream_t st, const char* name) {
  const int nh = H / P, nw = W / P;
  const int64_t items = frames * nh * P * nw;
  const void* pix = FWD ? x : (const void*)dx;
  const void* vecp = FWD ? (const void*)out : dout;
  const bool vec_ok = (W % 8 == 0) && (P % 8 == 0) && ((P * C) % 8 == 0) && dvt_aligned16(pix) &&
                      dvt_aligned16(vecp) && ((int64_t)H * W % 8 == 0);
  if constexpr (FWD && sizeof(D) == 2) {
    if (vec_ok && P == 16 && C == 3) {
      const int64_t npatches = frames * nh * nw;
      int64_t blocks = dvt_cdiv(dvt_cdiv(npatches, 4), 4);
      const int64_t cap = (int64_t)dvt_num_cus() * 16;
      if (blocks > cap) blocks = cap;
      hipLaunchKernelGGL((patchify16_fwd_kernel<S, D>), dim3((unsigned)blocks), dim3(256), 0, st, (const S*)x, (D*)out,
                         npatches, H, W, nh, nw);
      DVT_LAUNCH_CHECK(name);
      return DVT_OK;
    }
  }
  if (vec_ok && P == 16 && C == 3) {
    hipLaunchKernelGGL((patchify_vec_kernel<16, 3, S, D, FWD>), dim3(grid_for(items)), dim3(kBlock),
                       0, st, (const S*)x, (D*)out, (S*)dx, (const D*)dout, frames, H, W);
  } else if (vec_ok && P == 8 && C == 3) {
    hipLaunchKernelGGL((patchify_vec_kernel<8, 3, S, D, FWD>), dim3(grid_for(items)), dim3(kBlock),
                       0, st, (const S*)x, (D*)out, (S*)dx, (const D*)dout, frames, H, W);
  } else {
    const int64_t total = frames * nh * nw * (int64_t)P * P * C;
    hipLaunchKernelGGL((patchify_generic_kernel<S, D, FWD>), dim3(grid_for(total)), dim3(kBlock), 0,
                       st, (const S*)x, (D*)out, (S*)dx, (const D*)dout, frames, C, H, W, P);
  }
  DVT_LAUNCH_CHECK(name);
  return DVT_OK;
}

template <bool FWD>
int patchify_entry(const void* pix, int pix_dtype, const void* vec, int vec_dtype, int64_t frames,
                   int C, int H, int W, int P, dvt_stream_t stream, const char* name) {
  DVT_REQUIRE(pix && vec, "%s: null pointer", name);
  DVT_REQUIRE(frames >= 0 && C > 0 && H > 0 && W > 0 && P > 0, "%s: bad sizes", name);
  DVT_REQUIRE(H % P == 0 && W % P == 0, "%s: image %dx%d not divisible by patch %d", name, H, W, P);
  if (frames == 0) return DVT_OK;
  hipStream_t st = (hipStream_t)stream;
  // pixel tensor: x (fwd, const) / dx (bwd, written); vector tensor: out (fwd) / dout (bwd)
#define DVT_PATCH_CASE(PD, S, VD, D)                                                            \
  if (pix_dtype == PD && vec_dtype == VD)                                                       \
    return patchify_dispatch<S, D, FWD>(FWD ? pix : nullptr, FWD ? (void*)vec : nullptr,        \
                                        FWD ? nullptr : (void*)pix, FWD ? nullptr : vec, frames, \
                                        C, H, W, P, st, name);
  DVT_PATCH_CASE(DVT_F32, float, DVT_F32, float)
  DVT_PATCH_CASE(DVT_F32, float, DVT_BF16, bf16)
  DVT_PATCH_CASE(DVT_F32, float, DVT_F16, f16)
  DVT_PATCH_CASE(DVT_BF16, bf16, DVT_BF16, bf16)
  DVT_PATCH_CASE(DVT_F16, f16, DVT_F16, f16)
  DVT_PATCH_CASE(DVT_BF16, bf16, DVT_F32, float)
  DVT_PATCH_CASE(DVT_F16, f16, DVT_F32, float)
#undef DVT_PATCH_CASE
  DVT_UNSUPPORTED("%s: dtype pair (%d, %d) not supported", name, pix_dtype, vec_dtype);
}
}  // namespace

extern "C" {

int dvt_patchify(const void* x, int x_dtype, void* out, int out_dtype, int64_t frames, int C, int H,
                 int W, int P, dvt_stream_t stream) {
  return patchify_entry<true>(x, x_dtype, out, out_dtype, frames, C, H, W, P, stream, "dvt_patchify");
}

int dvt_patchify_bwd(const void* dout, int dout_dtype, void* dx, int dx_dtype, int64_t frames, int C,
                     int H, int W, int P, dvt_stream_t stream) {
  return patchify_entry<false>(dx, dx_dtype, dout, dout_dtype, frames, C, H, W, P, stream,
                               "dvt_patchify_bwd");
}

int dvt_tokens_assemble_fwd(const void* emb, const float* cls, const float* pos, void* out,
                            int64_t S, int64_t T, int64_t n, int64_t d, int64_t pos_rows, int dtype,
                            dvt_stream_t stream) {
  DVT_REQUIRE(emb && cls && pos && out, "dvt_tokens_assemble_fwd: null pointer");
  DVT_REQUIRE(S >= 0 && T > 0 && n >= 0 && d > 0 && S % T == 0, "dvt_tokens_assemble_fwd: bad sizes");
  DVT_REQUIRE(pos_rows >= n + 1, "dvt_tokens_assemble_fwd: positional table has %lld rows < n+1 = %lld",
              (long long)pos_rows, (long long)(n + 1));
  DVT_REQUIRE(d % 8 == 0, "dvt_tokens_assemble_fwd: d = %lld must be a multiple of 8", (long long)d);
  DVT_REQUIRE(dvt_aligned16(emb) && dvt_aligned16(cls) && dvt_aligned16(pos) && dvt_aligned16(out),
              "dvt_tokens_assemble_fwd: buffers must be 16-byte aligned");
  if (S == 0) return DVT_OK;
  hipStream_t st = (hipStream_t)stream;
  const int64_t items = S * (n + 1) * (d >> 3);
  DVT_DISPATCH_DTYPE(dtype, Tt,
                     hipLaunchKernelGGL((tokens_fwd_kernel<Tt>), dim3(grid_for(items)), dim3(kBlock),
                                        0, st, (const Tt*)emb, cls, pos, (Tt*)out, S, T, n, d,
                                        pos_rows));
  DVT_LAUNCH_CHECK("dvt_tokens_assemble_fwd");
  return DVT_OK;
}

int dvt_tokens_assemble_bwd(const void* dout, void* demb, float* dcls, float* dpos, int64_t S,
                            int64_t T, int64_t n, int64_t d, int64_t pos_rows, int dtype,
                            int accumulate, dvt_stream_t stream) {
  DVT_REQUIRE(dout, "dvt_tokens_assemble_bwd: null dout");
  DVT_REQUIRE(S > 0 && T > 0 && n >= 0 && d > 0 && S % T == 0, "dvt_tokens_assemble_bwd: bad sizes");
  DVT_REQUIRE(pos_rows >= n + 1 && d % 8 == 0, "dvt_tokens_assemble_bwd: bad pos_rows / d");
  DVT_REQUIRE(dvt_aligned16(dout) && dvt_aligned16(demb) && dvt_aligned16(dcls) && dvt_aligned16(dpos),
              "dvt_tokens_assemble_bwd: buffers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t dv = d >> 3;
  DVT_DISPATCH_DTYPE(dtype, Tt, {
    const bool one_pass = demb && n > 0 && dpos;     // the positional-gradient pass reads all of dout: it writes demb as well
    if (demb && n > 0 && !one_pass)
      hipLaunchKernelGGL((tokens_bwd_emb_kernel<Tt>), dim3(grid_for(S * n * dv)), dim3(kBlock), 0, st,
                         (const Tt*)dout, (Tt*)demb, S, n, d);
    if (dpos)
      hipLaunchKernelGGL((tokens_bwd_pos_kernel<Tt>), dim3(grid_for(T * pos_rows * dv)), dim3(kBlock),
                         0, st, (const Tt*)dout, dpos, S, T, n, d, pos_rows, accumulate, one_pass ? (Tt*)demb : (Tt*)nullptr);
    if (dcls)
      hipLaunchKernelGGL((strided_rows_sum_kernel<Tt>), dim3((unsigned)dv), dim3(kBlock), 0, st,
                         (const Tt*)dout, (n + 1) * d, S, dcls, accumulate);
  });
  DVT_LAUNCH_CHECK("dvt_tokens_assemble_bwd");
  return DVT_OK;
}

int dvt_rows_gather_fwd(const void* src, int64_t src_row_stride, const float* tok, void* out,
                        int64_t B, int64_t T, int64_t d, int dtype, dvt_stream_t stream) {
  DVT_REQUIRE(src && out, "dvt_rows_gather_fwd: null pointer");
  DVT_REQUIRE(B >= 0 && T >= 0 && d > 0 && d % 8 == 0 && src_row_stride % 8 == 0,
              "dvt_rows_gather_fwd: bad sizes (d and row stride must be multiples of 8)");
  DVT_REQUIRE(dvt_aligned16(src) && dvt_aligned16(out) && dvt_aligned16(tok),
              "dvt_rows_gather_fwd: buffers must be 16-byte aligned");
  if (B == 0) return DVT_OK;
  hipStream_t st = (hipStream_t)stream;
  const int lead = tok ? 1 : 0;
  const int64_t items = B * (T + lead) * (d >> 3);
  DVT_DISPATCH_DTYPE(dtype, Tt,
                     hipLaunchKernelGGL((rows_gather_fwd_kernel<Tt>), dim3(grid_for(items)),
                                        dim3(kBlock), 0, st, (const Tt*)src, src_row_stride, tok,
                                        (Tt*)out, B, T, d, lead));
  DVT_LAUNCH_CHECK("dvt_rows_gather_fwd");
  return DVT_OK;
}

int dvt_rows_gather_bwd(const void* dout, void* dsrc, int64_t src_row_stride, float* dtok, int64_t B,
                        int64_t T, int64_t d, int dtype, int accumulate, dvt_stream_t stream) {
  DVT_REQUIRE(dout && dsrc, "dvt_rows_gather_bwd: null pointer");
  DVT_REQUIRE(B > 0 && T >= 0 && d > 0 && d % 8 == 0 && src_row_stride % 8 == 0,
              "dvt_rows_gather_bwd: bad sizes");
  DVT_REQUIRE(dvt_aligned16(dout) && dvt_aligned16(dsrc) && dvt_aligned16(dtok),
              "dvt_rows_gather_bwd: buffers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int lead = dtok ? 1 : 0;
  const int64_t dv = d >> 3;
  DVT_DISPATCH_DTYPE(dtype, Tt, {
    if (T > 0)
      hipLaunchKernelGGL((rows_gather_bwd_kernel<Tt>), dim3(grid_for(B * T * dv) + (dtok ? (unsigned)dv : 0u)), dim3(kBlock),
                         0, st, (const Tt*)dout, (Tt*)dsrc, src_row_stride, B, T, d, lead, dtok, accumulate,
                         dtok ? (int)dv : 0);
    else if (dtok)
      hipLaunchKernelGGL((strided_rows_sum_kernel<Tt>), dim3((unsigned)dv), dim3(kBlock), 0, st,
                         (const Tt*)dout, (T + 1) * d, B, dtok, accumulate);
  });
  DVT_LAUNCH_CHECK("dvt_rows_gather_bwd");
  return DVT_OK;
}

int dvt_mean_rows_fwd(const void* x, void* out, int64_t B, int64_t L, int64_t d, float scale, int dtype,
                      dvt_stream_t stream) {
  DVT_REQUIRE(x && out && B >= 0 && L > 0 && d > 0 && d % 8 == 0, "dvt_mean_rows_fwd: bad arguments");
  DVT_REQUIRE(dvt_aligned16(x) && dvt_aligned16(out), "dvt_mean_rows_fwd: misaligned buffer");
  if (B == 0) return DVT_OK;
  hipStream_t st = (hipStream_t)stream;
  DVT_DISPATCH_DTYPE(dtype, T,
                     hipLaunchKernelGGL((mean_rows_kernel<T, true>), dim3(grid_for(B * (d >> 3))), dim3(kBlock), 0,
                                        st, (const T*)x, (T*)out, B, L, d, scale));
  DVT_LAUNCH_CHECK("dvt_mean_rows_fwd");
  return DVT_OK;
}

int dvt_mean_rows_bwd(const void* dout, void* dx, int64_t B, int64_t L, int64_t d, float scale, int dtype,
                      dvt_stream_t stream) {
  DVT_REQUIRE(dout && dx && B >= 0 && L > 0 && d > 0 && d % 8 == 0, "dvt_mean_rows_bwd: bad arguments");
  DVT_REQUIRE(dvt_aligned16(dout) && dvt_aligned16(dx), "dvt_mean_rows_bwd: misaligned buffer");
  if (B == 0) return DVT_OK;
  hipStream_t st = (hipStream_t)stream;
  DVT_DISPATCH_DTYPE(dtype, T,
                     hipLaunchKernelGGL((mean_rows_kernel<T, false>), dim3(grid_for(B * L * (d >> 3))),
                                        dim3(kBlock), 0, st, (const T*)dout, (T*)dx, B, L, d, scale));
  DVT_LAUNCH_CHECK("dvt_mean_rows_bwd");
  return DVT_OK;
}

int dvt_bce_logits_fwd(const void* z, const float* target, float* loss, int64_t n, int dtype,
                       dvt_stream_t stream) {
  DVT_REQUIRE(z && target && loss && n > 0, "dvt_bce_logits_fwd: null pointer or n <= 0");
  hipStream_t st = (hipStream_t)stream;
  DVT_DISPATCH_DTYPE(dtype, T,
                     hipLaunchKernelGGL((bce_fwd_kernel<T>), dim3(1), dim3(kBlock), 0, st,
                                        (const T*)z, target, loss, n));
  DVT_LAUNCH_CHECK("dvt_bce_logits_fwd");
  return DVT_OK;
}

int dvt_bce_logits_bwd(const void* z, const float* target, const float* gloss, void* dz, int64_t n,
                       int dtype, dvt_stream_t stream) {
  DVT_REQUIRE(z && target && gloss && dz && n > 0, "dvt_bce_logits_bwd: null pointer or n <= 0");
  hipStream_t st = (hipStream_t)stream;
  DVT_DISPATCH_DTYPE(dtype, T,
                     hipLaunchKernelGGL((bce_bwd_kernel<T>), dim3(grid_for(n)), dim3(kBlock), 0, st,
                                        (const T*)z, target, gloss, (T*)dz, n));
  DVT_LAUNCH_CHECK("dvt_bce_logits_bwd");
  return DVT_OK;
}

int dvt_ce_argmax_fwd(const void* student, const void* teacher, float* loss, int64_t rows, int64_t C,
                      int dtype, dvt_stream_t stream) {
  DVT_REQUIRE(student && teacher && loss && rows > 0 && C > 0, "dvt_ce_argmax_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  DVT_DISPATCH_DTYPE(dtype, T,
                     hipLaunchKernelGGL((ce_argmax_kernel<T, true>), dim3(1), dim3(kBlock), 0, st,
                                        (const T*)student, (const T*)teacher, (const float*)nullptr,
                                        loss, (T*)nullptr, rows, C));
  DVT_LAUNCH_CHECK("dvt_ce_argmax_fwd");
  return DVT_OK;
}

int dvt_ce_argmax_bwd(const void* student, const void* teacher, const float* gloss, void* dstudent,
                      int64_t rows, int64_t C, int dtype, dvt_stream_t stream) {
  DVT_REQUIRE(student && teacher && gloss && dstudent && rows > 0 && C > 0,
              "dvt_ce_argmax_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  DVT_DISPATCH_DTYPE(dtype, T,
                     hipLaunchKernelGGL((ce_argmax_kernel<T, false>), dim3(1), dim3(kBlock), 0, st,
                                        (const T*)student, (const T*)teacher, gloss, (float*)nullptr,
                                        (T*)dstudent, rows, C));
  DVT_LAUNCH_CHECK("dvt_ce_argmax_bwd");
  return DVT_OK;
}

int dvt_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                   dvt_stream_t stream) {
  if (n == 0) return DVT_OK;   // empty tensors carry null pointers: nothing to validate, nothing to launch
  DVT_REQUIRE(param && grad && exp_avg && exp_avg_sq && n >= 0 && step >= 1,
              "dvt_adamw_step: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, param, grad, exp_avg,
                     exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, (float)bc1,
                     (float)sqrt(bc2));
  DVT_LAUNCH_CHECK("dvt_adamw_step");
  return DVT_OK;
}

int dvt_dropout(const void* x, void* y, int64_t n, float p, const uint64_t* rng_state, uint64_t call_offset, int dtype,
                dvt_stream_t stream) {
  if (n == 0) return DVT_OK;   // empty tensors carry null pointers: nothing to validate, nothing to launch
  DVT_REQUIRE(x && y && rng_state && n >= 0 && p >= 0.f && p < 1.f, "dvt_dropout: bad arguments (0 <= p < 1)");
  // keep iff word >= threshold, threshold = round(p * 2^32): P(drop) = p to 2^-32
  const double th = (double)p * 4294967296.0;
  const uint32_t threshold = th >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)(th + 0.5);
  const float scale = 1.0f / (1.0f - p);
  DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((dropout_kernel<T>), dim3(grid_for((n + 3) >> 2)), dim3(kBlock), 0,
                                                  (hipStream_t)stream, (const T*)x, (T*)y, n, threshold, scale, rng_state,
                                                  call_offset));
  DVT_LAUNCH_CHECK("dvt_dropout");
  return DVT_OK;
}

int dvt_dropout_fused(const void* x, const void* residual, const void* gate, void* y, int64_t n, float p,
                      const uint64_t* rng_state, uint64_t call_offset, int relu, int dtype, dvt_stream_t stream) {
  if (n == 0) return DVT_OK;
  DVT_REQUIRE(x && y && n >= 0 && p >= 0.f && p < 1.f && (rng_state != nullptr) != (gate != nullptr),
              "dvt_dropout_fused: bad arguments (0 <= p < 1; exactly one of rng_state and gate)");
  const double th = (double)p * 4294967296.0;
  const uint32_t threshold = th >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)(th + 0.5);
  const float scale = 1.0f / (1.0f - p);
  DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((dropout_fused_kernel<T>), dim3(grid_for((n + 3) >> 2)), dim3(kBlock), 0,
                                                  (hipStream_t)stream, (const T*)x, (const T*)residual, (const T*)gate, (T*)y, n,
                                                  threshold, scale, rng_state, call_offset, relu));
  DVT_LAUNCH_CHECK("dvt_dropout_fused");
  return DVT_OK;
}

int dvt_rng_advance(uint64_t* rng_state, uint64_t delta, dvt_stream_t stream) {
  DVT_REQUIRE(rng_state, "dvt_rng_advance: null state");
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rng_state, delta);
  DVT_LAUNCH_CHECK("dvt_rng_advance");
  return DVT_OK;
}

int dvt_adamw_step_scaled(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                          float beta1, float beta2, float eps, float weight_decay, int64_t* step_dev, float* scale,
                          int32_t* found_inf, int32_t* good_steps, int growth_interval, float growth, float backoff,
                          float* loss_grad, float loss_grad_base, const uint8_t* skip64, dvt_stream_t stream) {
  DVT_REQUIRE(param && grad && exp_avg && exp_avg_sq && step_dev && scale && found_inf && good_steps && loss_grad &&
                  n >= 0 && growth_interval > 0 && growth >= 1.f && backoff > 0.f && backoff <= 1.f,
              "dvt_adamw_step_scaled: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (n > 0) {
    hipLaunchKernelGGL(check_finite_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, grad, n, (int*)found_inf);
    hipLaunchKernelGGL(adamw_scaled_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, param, grad, exp_avg, exp_avg_sq, n,
                       lr, beta1, beta2, eps, weight_decay, (const int64_t*)step_dev, (const float*)scale,
                       (const int*)found_inf, skip64);
  }
  hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(1), 0, st, step_dev, scale, (int*)found_inf,
                     (int*)good_steps, growth_interval, growth, backoff, loss_grad, loss_grad_base);
  DVT_LAUNCH_CHECK("dvt_adamw_step_scaled");
  return DVT_OK;
}

int dvt_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                 float weight_decay, const uint8_t* skip64, dvt_stream_t stream) {
  if (n == 0) return DVT_OK;   // empty tensors carry null pointers: nothing to validate, nothing to launch
  DVT_REQUIRE(param && grad && n >= 0 && (momentum == 0.f || momentum_buf), "dvt_sgd_step: bad arguments");
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, param, grad, momentum_buf,
                     n, lr, momentum, weight_decay, skip64);
  DVT_LAUNCH_CHECK("dvt_sgd_step");
  return DVT_OK;
}

int dvt_adagrad_step(float* param, const float* grad, float* state_sum, int64_t n, float lr, float lr_decay,
                     float eps, float weight_decay, int64_t step, const uint8_t* skip64, dvt_stream_t stream) {
  if (n == 0) return DVT_OK;   // empty tensors carry null pointers: nothing to validate, nothing to launch
  DVT_REQUIRE(param && grad && state_sum && n >= 0 && step >= 1, "dvt_adagrad_step: bad arguments");
  const float clr = (float)((double)lr / (1.0 + (double)(step - 1) * (double)lr_decay));
  hipLaunchKernelGGL(adagrad_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, param, grad, state_sum,
                     n, clr, eps, weight_decay, skip64);
  DVT_LAUNCH_CHECK("dvt_adagrad_step");
  return DVT_OK;
}

int dvt_adamw_step_fused(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                         float beta1, float beta2, float eps, float weight_decay, int64_t* step_dev2,
                         const uint8_t* skip64, void* mirror, int mirror_dtype, dvt_stream_t stream) {
  if (n == 0) return DVT_OK;
  DVT_REQUIRE(param && grad && exp_avg && exp_avg_sq && step_dev2 && n >= 0, "dvt_adamw_step_fused: bad arguments");
  DVT_REQUIRE(dvt_aligned16(param) && dvt_aligned16(grad) && dvt_aligned16(exp_avg) && dvt_aligned16(exp_avg_sq),
              "dvt_adamw_step_fused: buffers must be 16-byte aligned");
  DVT_REQUIRE(!mirror || (dvt_is_16bit(mirror_dtype) && ((uintptr_t)mirror & 7u) == 0),
              "dvt_adamw_step_fused: mirror must be bf16 / f16 and 8-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(grid_for((n >> 2) + 1)), block(256);
  if (!mirror)
    hipLaunchKernelGGL((adamw_fused_kernel<float>), grid, block, 0, st, param, grad, exp_avg, exp_avg_sq, n, lr, beta1,
                       beta2, eps, weight_decay, step_dev2, skip64, (float*)nullptr);
  else if (mirror_dtype == DVT_BF16)
    hipLaunchKernelGGL((adamw_fused_kernel<bf16>), grid, block, 0, st, param, grad, exp_avg, exp_avg_sq, n, lr, beta1,
                       beta2, eps, weight_decay, step_dev2, skip64, (bf16*)mirror);
  else
    hipLaunchKernelGGL((adamw_fused_kernel<f16>), grid, block, 0, st, param, grad, exp_avg, exp_avg_sq, n, lr, beta1,
                       beta2, eps, weight_decay, step_dev2, skip64, (f16*)mirror);
  DVT_LAUNCH_CHECK("dvt_adamw_step_fused");
  return DVT_OK;
}

int dvt_adamw_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                       float lr, float beta1, float beta2, float eps, float weight_decay,
                       int64_t* step_dev, const uint8_t* skip64, dvt_stream_t stream) {
  if (n == 0) return DVT_OK;   // empty tensors carry null pointers: nothing to validate, nothing to launch
  DVT_REQUIRE(param && grad && exp_avg && exp_avg_sq && step_dev && n >= 0, "dvt_adamw_step_dev: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adamw_dev_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, param, grad, exp_avg,
                     exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, (const int64_t*)step_dev, skip64);
  hipLaunchKernelGGL(inc_step_kernel, dim3(1), dim3(1), 0, st, step_dev);
  DVT_LAUNCH_CHECK("dvt_adamw_step_dev");
  return DVT_OK;
}

}  // extern "C"
