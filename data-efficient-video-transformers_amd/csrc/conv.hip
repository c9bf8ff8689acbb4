// conv.hip -- per-frame CNN encoder pieces (src/models/custom_resnet.py:19-153).
//
// Feature maps live in HBM as NHWC, i.e. as [N*H*W, C] row-major matrices, so that
//   * a convolution is   im2col gather -> MFMA GEMM (gemm256.hip) -> [N*Ho*Wo, Cout]  (already NHWC),
//     1x1 convolutions need no gather at all,
//   * BatchNorm / ReLU / residual are column-statistics + row-streaming kernels,
//   * global average pooling is the mean over rows of a [N, H*W, C] view.
// Column order is (ki, kj, c): contiguous C-runs.  The explicit gather below serves the C = 3 stem, strided data
// gradients and the fp32 parity mode; elsewhere the gather is fused into the GEMM's operand DMA (gemm256.hip,
// dvt_conv2d_implicit*).  All kernels in this file are HBM-bound streaming kernels.
#include "common.h"

namespace {

constexpr int kB = 256;

inline int cgrid(int64_t items) {
  int64_t b = dvt_cdiv(items, kB);
  const int64_t cap = (int64_t)dvt_num_cus() * 8;
  if (b > cap) b = cap;
  return (int)(b < 1 ? 1 : b);
}

// ------------------------------------------------------------------ im2col / col2im
// out[(n, ho, wo), (ki*kw + kj)*C + c] = x[n, ho*s - p + ki, wo*s - p + kj, c]   (0 outside)
// columns [kh*kw*C, ld) are zero (K padding for the MFMA kernels).
template <typename S, typename D, bool NCHW>
__global__ void im2col_kernel(const S* __restrict__ x, D* __restrict__ out, int N, int C, int H, int W,
                              int kh, int kw, int sh, int sw, int ph, int pw, int Ho, int Wo, int64_t ld) {
  const int64_t rows = (int64_t)N * Ho * Wo;
  const int K = kh * kw * C;
  const int64_t total = rows * ld;
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {
    const int col = (int)(i % ld);
    const int64_t r = i / ld;
    float v = 0.f;
    if (col < K) {
      const int c = col % C, kk = col / C, kj = kk % kw, ki = kk / kw;
      const int wo = (int)(r % Wo), ho = (int)((r / Wo) % Ho);
      const int64_t n = r / ((int64_t)Wo * Ho);
      const int h = ho * sh - ph + ki, w = wo * sw - pw + kj;
      if (h >= 0 && h < H && w >= 0 && w < W)
        v = to_f32<S>(NCHW ? x[((n * C + c) * H + h) * W + w] : x[((n * H + h) * W + w) * C + c]);
    }
    out[i] = from_f32<D>(v);
  }
}

// NCHW stem (C = 3): one thread builds 8 consecutive columns of one row and stores them with one 16-byte
// write; the column -> (ki, kj, c) decomposition uses compile-time divisors.  The reads hit L2 (every input
// pixel is used kh*kw/(sh*sw) times), the kernel is bound by its output write.
template <typename S, typename D, int CC, int KH, int KW>
__global__ void im2col_nchw_stem_kernel(const S* __restrict__ x, D* __restrict__ out, int N, int H, int W, int sh,
                                        int sw, int ph, int pw, int Ho, int Wo, int ld) {
  constexpr int K = KH * KW * CC;
  const int chunks = ld >> 3;
  const int64_t items = (int64_t)N * Ho * Wo * chunks;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(it % chunks);
    const int64_t r = it / chunks;
    const int wo = (int)(r % Wo), ho = (int)((r / Wo) % Ho);
    const int64_t n = r / ((int64_t)Wo * Ho);
    const S* xn = x + n * CC * (int64_t)H * W;
    const int h0 = ho * sh - ph, w0 = wo * sw - pw;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int col = ch * 8 + e;
      const int c = col % CC, kk = col / CC, kj = kk % KW, ki = kk / KW;
      const int h = h0 + ki, w = w0 + kj;
      v[e] = (col < K && h >= 0 && h < H && w >= 0 && w < W) ? to_f32<S>(xn[((int64_t)c * H + h) * W + w]) : 0.f;
    }
    store8<D>(out + r * ld + ch * 8, v);
  }
}

// vectorised NHWC form: C % 8 == 0, one thread copies 8 channels of one (row, ki, kj)
template <typename T>
__global__ void im2col_nhwc_vec_kernel(const T* __restrict__ x, T* __restrict__ out, int N, int C, int H,
                                       int W, int kh, int kw, int sh, int sw, int ph, int pw, int Ho, int Wo,
                                       int64_t ld) {
  const int cv = C >> 3;
  const int64_t rows = (int64_t)N * Ho * Wo;
  const int64_t items = rows * kh * kw * cv;
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += gs) {
    const int c = (int)(it % cv) << 3;
    int64_t t = it / cv;
    const int kk = (int)(t % (kh * kw));
    const int64_t r = t / (kh * kw);
    const int kj = kk % kw, ki = kk / kw;
    const int wo = (int)(r % Wo), ho = (int)((r / Wo) % Ho);
    const int64_t n = r / ((int64_t)Wo * Ho);
    const int h = ho * sh - ph + ki, w = wo * sw - pw + kj;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (h >= 0 && h < H && w >= 0 && w < W) load8<T>(x + ((n * H + h) * W + w) * C + c, v);
    store8<T>(out + r * ld + (int64_t)kk * C + c, v);
  }
}

// dx[n,h,w,c] = sum over (ki,kj) with (h+p-ki) % s == 0, (w+p-kj) % s == 0 of
//               dcol[(n, (h+p-ki)/s, (w+p-kj)/s), (ki*kw+kj)*C + c]        (gather form: no atomics)
template <typename T>
__global__ void col2im_nhwc_kernel(const T* __restrict__ dcol, T* __restrict__ dx, int N, int C, int H,
                                   int W, int kh, int kw, int sh, int sw, int ph, int pw, int Ho, int Wo,
                                   int64_t ld, const T* __restrict__ add = nullptr, int add_stride = 0) {
  const int cv = C >> 3;
  const int64_t items = (int64_t)N * H * W * cv;
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += gs) {
    const int c = (int)(it % cv) << 3;
    const int64_t px = it / cv;
    const int w = (int)(px % W), h = (int)((px / W) % H);
    const int64_t n = px / ((int64_t)W * H);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (add) {                                    // a second gradient path into the same pixel (see dvt_col2im)
      if (add_stride == 0) {
        load8<T>(add + px * C + c, acc);
      } else if (h % add_stride == 0 && w % add_stride == 0) {
        const int Hs = (H + add_stride - 1) / add_stride, Ws = (W + add_stride - 1) / add_stride;
        load8<T>(add + ((n * Hs + h / add_stride) * Ws + w / add_stride) * C + c, acc);
      }
    }
    for (int ki = 0; ki < kh; ++ki) {
      const int hh = h + ph - ki;
      if (hh < 0 || hh % sh) continue;
      const int ho = hh / sh;
      if (ho >= Ho) continue;
      for (int kj = 0; kj < kw; ++kj) {
        const int ww = w + pw - kj;
        if (ww < 0 || ww % sw) continue;
        const int wo = ww / sw;
        if (wo >= Wo) continue;
        float v[8];
        load8<T>(dcol + ((n * Ho + ho) * Wo + wo) * ld + (int64_t)(ki * kw + kj) * C + c, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += v[k];
      }
    }
    store8<T>(dx + px * C + c, acc);
  }
}

// ------------------------------------------------------------------ conv weight pack / unpack
// pack:   dst[co, (ki*kw+kj)*Cin + ci] = (T) w[co, ci, ki, kj]; columns >= kh*kw*Cin are 0
template <typename D>
__global__ void weight_pack_kernel(const float* __restrict__ w, D* __restrict__ dst, int Cout, int Cin,
                                   int kh, int kw, int64_t ld) {
  const int64_t total = (int64_t)Cout * ld;
  const int K = kh * kw * Cin;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % ld), co = (int)(i / ld);
    float v = 0.f;
    if (col < K) {
      const int ci = col % Cin, kk = col / Cin;
      v = w[((int64_t)co * Cin + ci) * kh * kw + kk];
    }
    dst[i] = from_f32<D>(v);
  }
}
// unpack: dw[co, ci, ki, kj] (+)= g[co, (ki*kw+kj)*Cin + ci]
__global__ void weight_unpack_kernel(const float* __restrict__ g, float* __restrict__ dw, int Cout, int Cin,
                                     int kh, int kw, int64_t ld, int accumulate) {
  const int64_t total = (int64_t)Cout * Cin * kh * kw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int kk = (int)(i % (kh * kw));
    const int ci = (int)((i / (kh * kw)) % Cin);
    const int co = (int)(i / ((int64_t)kh * kw * Cin));
    const float v = g[(int64_t)co * ld + (int64_t)kk * Cin + ci];
    dw[i] = accumulate ? dw[i] + v : v;
  }
}

// zero-padding of a [A, B, K] f32 array to [Ap, Bp, K] and the adjoint slice (channel padding of convolution
// weights / BatchNorm vectors: R(2+1)D mid-plane counts 45 / 230 / 460 / 921 are padded to MFMA-friendly multiples)
__global__ void pad3_kernel(const float* __restrict__ src, float* __restrict__ dst, int A, int B, int K, int Ap, int Bp) {
  const int64_t total = (int64_t)Ap * Bp * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % K);
    const int b = (int)((i / K) % Bp);
    const int a = (int)(i / ((int64_t)K * Bp));
    dst[i] = (a < A && b < B) ? src[((int64_t)a * B + b) * K + k] : 0.f;
  }
}

__global__ void unpad3_kernel(const float* __restrict__ src, float* __restrict__ dst, int A, int B, int K, int Bp,
                              int accumulate) {
  const int64_t total = (int64_t)A * B * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % K);
    const int b = (int)((i / K) % B);
    const int a = (int)(i / ((int64_t)K * B));
    const float v = src[((int64_t)a * Bp + b) * K + k];
    dst[i] = accumulate ? dst[i] + v : v;
  }
}

// transposed scatter: dw[co, ci, ki, kj] (+)= gt[(ki*kw + kj)*Cin + ci, co]
__global__ void weight_unpack_t_kernel(const float* __restrict__ gt, float* __restrict__ dw, int Cout, int Cin, int kh,
                                       int kw, int accumulate) {
  const int64_t total = (int64_t)Cout * Cin * kh * kw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int kk = (int)(i % (kh * kw));
    const int ci = (int)((i / (kh * kw)) % Cin);
    const int co = (int)(i / ((int64_t)kh * kw * Cin));
    const float v = gt[((int64_t)kk * Cin + ci) * Cout + co];
    dw[i] = accumulate ? dw[i] + v : v;
  }
}

// data-gradient operand: dst[ci, ((kh-1-ki)*kw + (kw-1-kj))*Cout + co] = (T) w[co, ci, ki, kj]
// (180-degree rotated taps, channels transposed): dx = conv(dz, dst) with padding k-1-p for a stride-1 convolution
template <typename D>
__global__ void weight_pack_dgrad_kernel(const float* __restrict__ w, D* __restrict__ dst, int Cout, int Cin, int kh,
                                         int kw) {
  const int64_t K = (int64_t)kh * kw * Cout;
  const int64_t total = (int64_t)Cin * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i / K);
    const int col = (int)(i % K);
    const int co = col % Cout, tap = col / Cout;
    const int kir = tap / kw, kjr = tap % kw;
    const int ki = kh - 1 - kir, kj = kw - 1 - kjr;
    dst[i] = from_f32<D>(w[(((int64_t)co * Cin + ci) * kh + ki) * kw + kj]);
  }
}

// Both packed forms of MANY convolution weights in one launch (once per optimizer step: the weights only change there), with
// the zero extension of channel-padded layers folded in.  Both forms are transposes of the parameter's [cout][cin * taps]
// rows, so both go through LDS with coalesced accesses on either side (an element-per-thread gather read one 4-byte word per
// 64-byte line for the data-gradient form and paid two 64-bit divisions per element):
//   kind 0 (forward operand [cout_p][ld], column = tap * cin_p + ci): a workgroup stages R whole parameter rows (one
//           contiguous read) and writes R destination rows, reading LDS with stride `taps`;
//   kind 1 (data-gradient operand [cin_p][taps * cout_p], rotated taps): a workgroup transposes a 64 (cout) x 64 (cin * taps)
//           tile: 256-byte reads along the parameter row, 128-byte writes along cout.
// Rows longer than kPackRowMax words (none in the path's models) keep the element-per-thread form.
constexpr int kPackGroup = 40;
constexpr int kPackRowMax = 8192;
struct PackGroup { dvt_pack_entry e[kPackGroup]; int begin[kPackGroup + 1]; int n; };

// workgroups of an entry and, for kind 0, the rows each one packs (host and device agree through this one function)
__host__ __device__ inline int pack_plan(const dvt_pack_entry& q, int* rows_per_block) {
  const int taps = q.kh * q.kw;
  const int64_t rowlen = (int64_t)q.cin_l * taps;
  *rows_per_block = 0;
  if (q.kind == 0) {
    if (rowlen > kPackRowMax || q.ld > 4 * kPackRowMax)
      return (int)(((int64_t)q.cout_p * q.ld + 2047) / 2048);
    const int R = q.ld >= 2048 ? 1 : 2048 / q.ld;
    *rows_per_block = R;
    return (q.cout_p + R - 1) / R;
  }
  return ((q.cout_p + 63) / 64) * (int)(((int64_t)q.cin_p * taps + 63) / 64);
}

template <typename D>
__device__ __forceinline__ void pack_elems(const dvt_pack_entry& q, int64_t i0) {          // fallback for very long rows
  const int taps = q.kh * q.kw;
  const float* __restrict__ w = q.src;
  D* __restrict__ dst = (D*)q.dst;
  const int64_t total = (int64_t)q.cout_p * q.ld;
  for (int64_t i = i0 + threadIdx.x; i < min(total, i0 + 2048); i += 256) {
    const int col = (int)(i % q.ld), co = (int)(i / q.ld);
    const int ci = col % q.cin_p, tap = col / q.cin_p;
    float v = 0.f;
    if (co < q.cout_l && ci < q.cin_l && tap < taps) v = w[((int64_t)co * q.cin_l + ci) * taps + tap];
    dst[i] = from_f32<D>(v);
  }
}

template <typename D>
__device__ __forceinline__ void pack_rows(const dvt_pack_entry& q, int blk, int R, float* lds) {
  const int taps = q.kh * q.kw, rowlen = q.cin_l * taps;
  const int co0 = blk * R;
  const int64_t src0 = (int64_t)co0 * rowlen, src_end = (int64_t)q.cout_l * rowlen;
  const int total = R * rowlen;
  if ((rowlen & 3) == 0 && (((uintptr_t)q.src) & 15) == 0) {   // 16-byte loads (the rows are contiguous in the parameter)
    for (int i = threadIdx.x * 4; i < total; i += 1024) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (src0 + i < src_end) v = *reinterpret_cast<const f32x4*>(q.src + src0 + i);
      *reinterpret_cast<f32x4*>(lds + i) = v;
    }
  } else {
    for (int i = threadIdx.x; i < total; i += 256) lds[i] = src0 + i < src_end ? q.src[src0 + i] : 0.f;
  }
  __syncthreads();
  D* __restrict__ dst = (D*)q.dst;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int body = taps * q.cin_p;
  if (sizeof(D) == 2 && (q.cin_p & 7) == 0 && (q.ld & 7) == 0 && (((uintptr_t)q.dst) & 15) == 0) {
    // a thread writes 8 consecutive input channels of one (row, tap) pair: one 16-byte store from eight LDS reads of stride
    // `taps` (2-byte stores were 128 bytes per wave instruction)
    const int c8n = q.cin_p >> 3, chunks = R * taps * c8n;
    for (int ch = threadIdx.x; ch < chunks; ch += 256) {
      const int pr = ch / c8n, c8 = ch - pr * c8n;
      const int r = pr / taps, tap = pr - r * taps;
      if (co0 + r >= q.cout_p) break;
      const float* l = lds + r * rowlen + tap;
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int ci = c8 * 8 + k;
        v[k] = ci < q.cin_l ? l[ci * taps] : 0.f;
      }
      store8<D>(dst + (int64_t)(co0 + r) * q.ld + tap * q.cin_p + c8 * 8, v);
    }
  } else {
    for (int p = ty; p < R * taps; p += 4) {                     // (row, tap) pairs, one per wave at a time
      const int r = p / taps, tap = p - r * taps;
      if (co0 + r >= q.cout_p) break;
      D* __restrict__ d = dst + (int64_t)(co0 + r) * q.ld + tap * q.cin_p;
      const float* l = lds + r * rowlen + tap;
      for (int ci = tx; ci < q.cin_p; ci += 64) d[ci] = from_f32<D>(ci < q.cin_l ? l[ci * taps] : 0.f);
    }
  }
  if (q.ld > body)                                             // K padding of the row
    for (int r = 0; r < R && co0 + r < q.cout_p; ++r)
      for (int j = body + threadIdx.x; j < q.ld; j += 256) dst[(int64_t)(co0 + r) * q.ld + j] = from_f32<D>(0.f);
}

// destination tap block of source tap (ki, kj) in the data-gradient forms: kind 1 = all taps, rotated; kind 2 = the taps of one
// parity class of a strided convolution (ki = rh + j * sh, j < nth; kj likewise), in decreasing ki / kj order; -1: not in it
__host__ __device__ inline int pack_cls_nt(int k, int r, int s) { return r < k ? (k - r + s - 1) / s : 0; }
__device__ __forceinline__ int pack_dst_tap(const dvt_pack_entry& q, int tap, int* ntaps_dst) {
  if (q.kind != 2) { *ntaps_dst = q.kh * q.kw; return q.kh * q.kw - 1 - tap; }
  const int nth = pack_cls_nt(q.kh, q.cls_rh, q.cls_sh), ntw = pack_cls_nt(q.kw, q.cls_rw, q.cls_sw);
  *ntaps_dst = nth * ntw;
  const int ki = tap / q.kw, kj = tap - ki * q.kw;
  const int dh = ki - q.cls_rh, dw = kj - q.cls_rw;
  if (dh < 0 || dw < 0 || dh % q.cls_sh || dw % q.cls_sw) return -1;
  return (nth - 1 - dh / q.cls_sh) * ntw + (ntw - 1 - dw / q.cls_sw);
}

template <typename D>
__device__ __forceinline__ void pack_tile_t(const dvt_pack_entry& q, int blk, float* lds) {
  const int taps = q.kh * q.kw, rowlen = q.cin_l * taps, cols = q.cin_p * taps;
  const int tiles_co = (q.cout_p + 63) / 64;
  const int co0 = (blk % tiles_co) * 64, c0 = (blk / tiles_co) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  if ((rowlen & 3) == 0 && (((uintptr_t)q.src) & 15) == 0) {   // 16-byte loads: thread = (row, four columns), four rounds
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = it * 256 + threadIdx.x, r = idx >> 4, c4 = (idx & 15) << 2;
      const int co = co0 + r, c = c0 + c4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (co < q.cout_l && c < rowlen) v = *reinterpret_cast<const f32x4*>(q.src + (int64_t)co * rowlen + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) lds[r * 65 + c4 + k] = v[k];
    }
  } else {
    for (int r = ty; r < 64; r += 4) {
      const int co = co0 + r, c = c0 + tx;
      lds[r * 65 + tx] = (co < q.cout_l && c < rowlen) ? q.src[(int64_t)co * rowlen + c] : 0.f;
    }
  }
  __syncthreads();
  D* __restrict__ dst = (D*)q.dst;
  if (sizeof(D) == 2 && (q.cout_p & 7) == 0 && (((uintptr_t)q.dst) & 15) == 0) {
    // a thread writes 8 consecutive output channels of one (channel, tap) row: 16-byte stores, 128 bytes per row from 8 lanes
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int ch = it * 256 + threadIdx.x, cc = ch >> 3, o8 = (ch & 7) << 3;
      const int c = c0 + cc, co = co0 + o8;
      if (c >= cols || co >= q.cout_p) continue;
      const int ci = c / taps, tap = c - ci * taps;
      int ntd;
      const int td = pack_dst_tap(q, tap, &ntd);
      if (td < 0) continue;
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = lds[(o8 + k) * 65 + cc];
      store8<D>(dst + ((int64_t)ci * ntd + td) * q.cout_p + co, v);
    }
    return;
  }
  const int co = co0 + tx;
  if (co >= q.cout_p) return;
  for (int cc = ty; cc < 64; cc += 4) {
    const int c = c0 + cc;
    if (c >= cols) break;
    const int ci = c / taps, tap = c - ci * taps;
    int ntd;
    const int td = pack_dst_tap(q, tap, &ntd);
    if (td >= 0) dst[((int64_t)ci * ntd + td) * q.cout_p + co] = from_f32<D>(lds[tx * 65 + cc]);
  }
}

__global__ __launch_bounds__(256) void weight_pack_group_kernel(const PackGroup g) {
  __shared__ float lds[kPackRowMax];
  // entry of this workgroup: the last e with begin[e] <= blockIdx.x (bisection: a linear scan was up to ~100 dependent
  // scalar loads in front of ~1 us of work per workgroup)
  int e = 0, hi = g.n;
  while (hi - e > 1) {
    const int mid = (e + hi) >> 1;
    if ((int)blockIdx.x >= g.begin[mid]) e = mid; else hi = mid;
  }
  const dvt_pack_entry& q = g.e[e];
  const int blk = (int)blockIdx.x - g.begin[e];
  int R;
  pack_plan(q, &R);
  if (q.kind >= 1) {
    if (q.dtype == DVT_BF16) pack_tile_t<bf16>(q, blk, lds);
    else if (q.dtype == DVT_F16) pack_tile_t<f16>(q, blk, lds);
    else pack_tile_t<float>(q, blk, lds);
  } else if (R > 0) {
    if (q.dtype == DVT_BF16) pack_rows<bf16>(q, blk, R, lds);
    else if (q.dtype == DVT_F16) pack_rows<f16>(q, blk, R, lds);
    else pack_rows<float>(q, blk, R, lds);
  } else {
    if (q.dtype == DVT_BF16) pack_elems<bf16>(q, (int64_t)blk * 2048);
    else if (q.dtype == DVT_F16) pack_elems<f16>(q, (int64_t)blk * 2048);
    else pack_elems<float>(q, (int64_t)blk * 2048);
  }
}

// ---- row-streaming BatchNorm kernels.  A thread owns ONE 8-channel group for its whole life (cg = thread % (C/8)) and walks
// the rows lane, lane + lanes, ...: the per-channel constants are loaded and folded ONCE (reloading six parameter vectors
// per 16 bytes of payload kept the CU's L1 / address pipeline four times busier with parameters than with data), and
// kBnU rows are requested before the first is consumed.
constexpr int kBnU = 4;

struct RowMap { int cg; int64_t row, lanes; };
__device__ __forceinline__ RowMap bn_row_map(int cv) {
  const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nth = (int64_t)gridDim.x * blockDim.x;
  RowMap m;
  m.lanes = nth / cv;                    // whole rows per sweep; the last nth % cv threads idle
  m.cg = (int)(gtid % cv);
  m.row = gtid / cv;
  return m;
}

// The affine part y = (x - mean) * invstd * gamma + beta of 8 channels.  16-bit maps: folded to one fma (x * s + t); the fp32
// parity mode keeps the reference's operation order.  EVERY kernel that decides a ReLU mask (forward, the pooled stem
// forward, the backward passes that recompute the mask from z) goes through apply(): one formula, identical decisions.
template <typename T>
struct BnAffine {
  float mu[8], is[8], g[8], b[8], s[8], t[8];
  // cv: number of channels gamma / beta really have (channel-padded layers: the channels beyond are gamma = beta = 0, i.e.
  // exact zeros through BatchNorm, ReLU and back); mean / invstd are internal arrays of the padded width
  __device__ __forceinline__ void init(const float* mean, const float* invstd, const float* gamma, const float* beta, int c,
                                       int cv) {
    load8<float>(mean + c, mu); load8<float>(invstd + c, is);
    if (c + 8 <= cv) {
      load8<float>(gamma + c, g);
      if (beta) load8<float>(beta + c, b);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        g[k] = c + k < cv ? gamma[c + k] : 0.f;
        b[k] = (beta && c + k < cv) ? beta[c + k] : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (!beta) b[k] = 0.f;
      s[k] = is[k] * g[k];
      t[k] = fmaf(-mu[k], s[k], b[k]);
    }
  }
  __device__ __forceinline__ float apply(float x, int k) const {
    if (sizeof(T) == 4) return fmaf((x - mu[k]) * is[k], g[k], b[k]);
    return fmaf(x, s[k], t[k]);
  }
  __device__ __forceinline__ float xhat(float x, int k) const { return (x - mu[k]) * is[k]; }
};

// ------------------------------------------------------------------ BatchNorm (columns of [rows, C])
// partial[b][0][c] = sum x, partial[b][1][c] = sum x^2 over the block's rows  (MODE 0)
// partial[b][0][c] = sum dz, partial[b][1][c] = sum dz * xhat                 (MODE 1), dz = dy * (y > 0 if relu)
// POOL: dy is the gradient of a 3x3 / 2 / 1 max-pool of this map (rows = N*H*W pixels); the map's own gradient is
// gathered from it on the fly (pooled_dy8_k3s2p1) instead of being read from a materialised tensor.
struct PoolGeom { const unsigned char* idx; int H, W, Ho, Wo; };
template <typename T> __device__ __forceinline__ void pooled_dy8_k3s2p1(const T*, const unsigned char*, int64_t, int, int, int, int, int, int, float*);

template <typename T, int MODE, bool POOL = false, int MSRC = 0>
__global__ __launch_bounds__(256) void bn_colstats_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                          const T* __restrict__ y, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, int64_t rows, int C,
                                                          int rows_per_block, int relu, int vc,
                                                          float* __restrict__ partial,
                                                          const float* __restrict__ gamma = nullptr,
                                                          const float* __restrict__ beta = nullptr,
                                                          PoolGeom pg = PoolGeom{nullptr, 0, 0, 0, 0},
                                                          const unsigned char* __restrict__ mask = nullptr, int c_valid = 1 << 30) {
  // 256 threads = vc column-vectors (8 channels each) x nrl = 256 / vc row lanes (the 256 % vc last threads idle); vc is a
  // divisor of C / 8 up to 32 (bn_vc), so narrow maps (C = 64: vc = 8, 32 row lanes) AND widths that are not powers of two
  // (R(2+1)D's 144 / 240 / 464 / 928 planes) keep every lane busy -- with vc = 16 for C = 144 the second block column had 2
  // of its 16 column lanes in use and those few threads walked the whole row range alone (198 us for 607 MB).  Four rows are
  // requested before the first is consumed.  ReLU mask of MODE 1 (MSRC): 2 = the forward's mask bytes, 1 = the stored output
  // y, 0 = recomputed from x.
  __shared__ float red[2][256][8];
  constexpr int U = POOL ? 1 : 4;
  const int nrl = 256 / vc;
  const int cl = threadIdx.x % vc, rl = threadIdx.x / vc;
  const int c = (blockIdx.x * vc + cl) * 8;
  const int cv = C >> 3;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = min(rows, r0 + rows_per_block);
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c < C && r0 < r1 && rl < nrl) {
    BnAffine<T> af;
    if (MODE == 1) af.init(mean, invstd, gamma, (relu && MSRC == 0) ? beta : nullptr, c, c_valid);
    constexpr int msrc = MSRC;
    for (int64_t rb = r0 + rl; rb < r1; rb += (int64_t)U * nrl) {
      float xv[U][8], dv[U][8], yv[U][8];
      unsigned mb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = min(rb + (int64_t)u * nrl, r1 - 1);      // clamped; surplus rows are dropped below
        load8<T>(x + r * C + c, xv[u]);
        if (MODE == 1) {
          if (POOL) {
            const int w = (int)(r % pg.W), h = (int)((r / pg.W) % pg.H);
            pooled_dy8_k3s2p1<T>(dy, pg.idx, r / ((int64_t)pg.W * pg.H), h, w, c, C, pg.Ho, pg.Wo, dv[u]);
          } else {
            load8<T>(dy + r * C + c, dv[u]);
          }
          if (msrc == 1) load8<T>(y + r * C + c, yv[u]);
          if (msrc == 2) mb[u] = mask[r * cv + (c >> 3)];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (rb + (int64_t)u * nrl >= r1) break;
        if (MODE == 0) {
#pragma unroll
          for (int k = 0; k < 8; ++k) { a[k] += xv[u][k]; b[k] = fmaf(xv[u][k], xv[u][k], b[k]); }
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float xh = af.xhat(xv[u][k], k);
            bool on = true;
            if (relu) on = msrc == 2 ? ((mb[u] >> k) & 1u) != 0u : (msrc == 1 ? yv[u][k] > 0.f : af.apply(xv[u][k], k) > 0.f);
            const float dz = on ? dv[u][k] : 0.f;
            a[k] += dz;
            b[k] = fmaf(dz, xh, b[k]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[0][threadIdx.x][k] = a[k]; red[1][threadIdx.x][k] = b[k]; }
  __syncthreads();
  // thread t < 2*vc*8 sums one (stat, column) over the row lanes in a fixed order
  for (int t = threadIdx.x; t < 2 * vc * 8; t += 256) {
    const int st = t / (vc * 8), col = t % (vc * 8), cvi = col >> 3, k = col & 7;
    const int cc = (blockIdx.x * vc + cvi) * 8 + k;
    if (cc < C) {
      float acc = 0.f;
      for (int r = 0; r < nrl; ++r) acc += red[st][r * vc + cvi][k];
      partial[((int64_t)blockIdx.y * 2 + st) * C + cc] = acc;
    }
  }
}

// MODE 0: mean, invstd (biased var), running stats update.  MODE 1: dgamma = sum dz*xhat, dbeta = sum dz.
template <int MODE, int CL = 32>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ partial, int nparts, int C,
                                                           float inv_rows, float eps, float* __restrict__ o0,
                                                           float* __restrict__ o1, float* __restrict__ run_mean,
                                                           float* __restrict__ run_var, float momentum, float unbias,
                                                           int accumulate, float* __restrict__ pub0 = nullptr,
                                                           float* __restrict__ pub1 = nullptr, int c_valid = 1 << 30) {
  // c_valid: the channels the CALLER's arrays (running statistics; published dgamma / dbeta) have -- a channel-padded layer's
  // padded channels exist in the internal arrays (o0 / o1) only.
  // block = CL columns x PL = 1024 / CL part lanes; fixed summation order (lane-strided partial sums, then a lane tree).
  // CL = 8 for narrow maps: a 64-channel layer then runs on 8 CUs instead of 2 (the launch is bound by what ONE CU can pull
  // in -- up to 1 MB of partial rows -- not by arithmetic).
  // The partial sums are added in double and the variance is formed in double: the one-pass form E[x^2] - mu^2 in fp32
  // loses the variance of channels whose |mean| is large against their spread to the rounding of sums over 10^5..10^7
  // rows; with double accumulation of the (fp32, <= 128-row) partials what is left is the rounding inside one partial.
  constexpr int PL = 1024 / CL;
  __shared__ double red[2][PL][CL + 1];
  const int cl = threadIdx.x % CL, pl = threadIdx.x / CL;
  const int c = blockIdx.x * CL + cl;
  double s0 = 0.0, s1 = 0.0;
  if (c < C) {
    // sixteen, then eight, then four partial rows per round trip (a rolled loop waits for each before it requests the next; with
    // four per trip the 8 - 18 rows a thread owns at 1024 - 2304 partial rows were two to five dependent trips to another XCD's L2:
    // 6.4 -> 5.7 us for the 22 launches per frametransformer step that see that many); same order of additions
    int p = pl;
    for (; p + 15 * PL < nparts; p += 16 * PL) {
      float a[16], b[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        a[u] = partial[((int64_t)(p + PL * u) * 2 + 0) * C + c];
        b[u] = partial[((int64_t)(p + PL * u) * 2 + 1) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) { s0 += (double)a[u]; s1 += (double)b[u]; }
    }
    if (p + 7 * PL < nparts) {
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a[u] = partial[((int64_t)(p + PL * u) * 2 + 0) * C + c];
        b[u] = partial[((int64_t)(p + PL * u) * 2 + 1) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s0 += (double)a[u]; s1 += (double)b[u]; }
      p += 8 * PL;
    }
    for (; p + 3 * PL < nparts; p += 4 * PL) {
      float a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = partial[((int64_t)(p + PL * u) * 2 + 0) * C + c];
        b[u] = partial[((int64_t)(p + PL * u) * 2 + 1) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { s0 += (double)a[u]; s1 += (double)b[u]; }
    }
    for (; p < nparts; p += PL) {
      s0 += (double)partial[((int64_t)p * 2 + 0) * C + c];
      s1 += (double)partial[((int64_t)p * 2 + 1) * C + c];
    }
  }
  red[0][pl][cl] = s0;
  red[1][pl][cl] = s1;
  __syncthreads();
  // lane tree in two levels (fixed order): part lanes 0 .. 7 each add PL / 8 consecutive lanes, lane 0 adds those eight (one
  // thread adding all PL = 128 lanes was a chain of 128 dependent double additions)
  constexpr int PG = PL / 8;
  if (pl < 8) {
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int q = 0; q < PG; ++q) { t0 += red[0][pl * PG + q][cl]; t1 += red[1][pl * PG + q][cl]; }
    s0 = t0; s1 = t1;
  }
  __syncthreads();
  if (pl < 8) { red[0][pl][cl] = s0; red[1][pl][cl] = s1; }
  __syncthreads();
  if (pl != 0 || c >= C) return;
  s0 = 0.0; s1 = 0.0;
#pragma unroll
  for (int p = 0; p < 8; ++p) { s0 += red[0][p][cl]; s1 += red[1][p][cl]; }
  if (MODE == 0) {
    const double mud = s0 * (double)inv_rows;
    const double vard = s1 * (double)inv_rows - mud * mud;
    const float mu = (float)mud;
    const float var = vard > 0.0 ? (float)vard : 0.f;
    o0[c] = mu;
    o1[c] = rsqrtf(var + eps);
    if (run_mean && c < c_valid) {
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * unbias;
    }
  } else {
    const float d0 = (float)s0, d1 = (float)s1;
    if (pub0) {                             // this launch's own sums for the apply pass; the caller's (accumulated) result
      o0[c] = d1;
      o1[c] = d0;
      if (c < c_valid) {
        pub0[c] = accumulate ? pub0[c] + d1 : d1;   // dgamma
        pub1[c] = accumulate ? pub1[c] + d0 : d0;   // dbeta
      }
    } else {
      o0[c] = accumulate ? o0[c] + d1 : d1;   // dgamma
      o1[c] = accumulate ? o1[c] + d0 : d0;   // dbeta
    }
  }
}

// launch of the finalize kernel with the column-lane count that suits C
template <int MODE>
static void bn_finalize_launch(hipStream_t st, const float* partial, int nparts, int C, float inv_rows, float eps, float* o0,
                               float* o1, float* run_mean, float* run_var, float momentum, float unbias, int accumulate,
                               float* pub0 = nullptr, float* pub1 = nullptr, int c_valid = 1 << 30) {
  if (C <= 256)
    hipLaunchKernelGGL((bn_finalize_kernel<MODE, 8>), dim3((unsigned)dvt_cdiv(C, 8)), dim3(1024), 0, st, partial, nparts, C,
                       inv_rows, eps, o0, o1, run_mean, run_var, momentum, unbias, accumulate, pub0, pub1, c_valid);
  else
    hipLaunchKernelGGL((bn_finalize_kernel<MODE, 32>), dim3((unsigned)dvt_cdiv(C, 32)), dim3(1024), 0, st, partial, nparts, C,
                       inv_rows, eps, o0, o1, run_mean, run_var, momentum, unbias, accumulate, pub0, pub1, c_valid);
}

}  // namespace

// For the kernels of other translation units that produce the backward sums themselves (conv3x3_stream.hip's fused data
// gradient + BatchNorm backward): partial rows [nparts][2][C] = (sum dz, sum dz * xhat) -> loc[0][C] = sum dz * xhat,
// loc[1][C] = sum dz, and the caller's dgamma / dbeta overwritten or accumulated.
namespace dvt_internal {
void bn_bwd_finalize(hipStream_t st, const float* partial, int nparts, int C, float* loc, int accumulate, float* dgamma,
                     float* dbeta, int c_valid) {
  bn_finalize_launch<1>(st, partial, nparts, C, 0.f, 0.f, loc, loc + C, (float*)nullptr, (float*)nullptr, 0.f, 0.f, accumulate,
                        dgamma, dbeta, c_valid);
}
}  // namespace dvt_internal

namespace {

// Fold many partial rows (one per 128 output rows of a convolution: thousands) into gridDim.y rows that bn_finalize can
// sum: block = 32 columns x 32 part lanes over its contiguous range of parts, four loads in flight, fixed order.
__global__ __launch_bounds__(1024) void bn_partial_fold_kernel(const float* __restrict__ partial, int nparts, int C,
                                                               int per_block, float* __restrict__ out) {
  __shared__ double red[2][32][33];
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int p0 = blockIdx.y * per_block, p1 = min(nparts, p0 + per_block);
  double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};
  if (c < C) {
    int p = p0 + pl;
    for (; p + 96 < p1; p += 128) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s0[u] += partial[((int64_t)(p + 32 * u) * 2 + 0) * C + c];
        s1[u] += partial[((int64_t)(p + 32 * u) * 2 + 1) * C + c];
      }
    }
    for (; p < p1; p += 32) {
      s0[0] += partial[((int64_t)p * 2 + 0) * C + c];
      s1[0] += partial[((int64_t)p * 2 + 1) * C + c];
    }
  }
  red[0][pl][cl] = (s0[0] + s0[1]) + (s0[2] + s0[3]);
  red[1][pl][cl] = (s1[0] + s1[1]) + (s1[2] + s1[3]);
  __syncthreads();
  if (pl != 0 || c >= C) return;
  double t0 = 0.0, t1 = 0.0;
  for (int q = 0; q < 32; ++q) { t0 += red[0][q][cl]; t1 += red[1][q][cl]; }
  out[((int64_t)blockIdx.y * 2 + 0) * C + c] = (float)t0;
  out[((int64_t)blockIdx.y * 2 + 1) * C + c] = (float)t1;
}

__global__ void rsqrt_eps_kernel(const float* __restrict__ var, float* __restrict__ out, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) out[c] = rsqrtf(var[c] + eps);
}

// y = relu?( BN(x) (+ residual) ).  MASK: also leaves one byte per (row, channel group) whose bit k says y[.., c + k] > 0 --
// the ReLU mask the backward of a residual layer needs (its output cannot be recomputed from z alone), 1/16 of re-reading y.
template <typename T, bool RES, bool MASK>
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const T* __restrict__ residual,
                                                           T* __restrict__ y, unsigned char* __restrict__ mask, int64_t rows,
                                                           int C, int relu, int c_valid) {
  const int cv = C >> 3;
  const RowMap m = bn_row_map(cv);
  if (m.row >= m.lanes) return;
  const int c = m.cg << 3;
  BnAffine<T> af;
  af.init(mean, invstd, gamma, beta, c, c_valid);
  for (int64_t r0 = m.row; r0 < rows; r0 += kBnU * m.lanes) {
    float xv[kBnU][8], rv[kBnU][8];
#pragma unroll
    for (int u = 0; u < kBnU; ++u) {
      const int64_t r = min(r0 + u * m.lanes, rows - 1);       // clamped: loaded unconditionally, stored only when live
      load8<T>(x + r * C + c, xv[u]);
      if (RES) load8<T>(residual + r * C + c, rv[u]);
    }
#pragma unroll
    for (int u = 0; u < kBnU; ++u) {
      const int64_t r = r0 + u * m.lanes;
      if (r >= rows) break;
      unsigned bits = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float v = af.apply(xv[u][k], k);
        if (RES) v += rv[u][k];
        if (relu) v = fmaxf(v, 0.f);
        xv[u][k] = v;
        if (MASK) bits |= (to_f32<T>(from_f32<T>(v)) > 0.f ? 1u : 0u) << k;   // the stored (rounded) value decides
      }
      store8<T>(y + r * C + c, xv[u]);
      if (MASK) mask[r * cv + m.cg] = (unsigned char)bits;
    }
  }
}

// dz = dy * (relu mask);  dres = dz (if wanted);
// train: dx = gamma*invstd*(dz - sum_dz/rows - xhat*sum_dzxhat/rows);  eval: dx = gamma*invstd*dz
// The ReLU mask comes from (in this order) the mask bytes of the forward (MSRC 2), the stored output y (MSRC 1), or is
// recomputed from z (MSRC 0, layers without a residual branch).
template <typename T, int MSRC, bool DRES>
__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                           const unsigned char* __restrict__ mask,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                           T* __restrict__ dx, T* __restrict__ dres, int64_t rows, int C,
                                                           int relu, int training, float inv_rows, int c_valid) {
  const int cv = C >> 3;
  const RowMap m = bn_row_map(cv);
  if (m.row >= m.lanes) return;
  const int c = m.cg << 3;
  BnAffine<T> af;
  af.init(mean, invstd, gamma, MSRC == 0 ? beta : nullptr, c, c_valid);
  float gi[8], kb[8], kg[8];
  {
    float dg[8], db[8];
    load8<float>(dgamma + c, dg); load8<float>(dbeta + c, db);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      gi[k] = af.g[k] * af.is[k];
      kb[k] = training ? db[k] * inv_rows : 0.f;
      kg[k] = training ? dg[k] * inv_rows : 0.f;
    }
  }
  for (int64_t r0 = m.row; r0 < rows; r0 += kBnU * m.lanes) {
    float dv[kBnU][8], xv[kBnU][8], yv[kBnU][8];
    unsigned mb[kBnU];
#pragma unroll
    for (int u = 0; u < kBnU; ++u) {
      const int64_t r = min(r0 + u * m.lanes, rows - 1);
      load8<T>(dy + r * C + c, dv[u]);
      load8<T>(x + r * C + c, xv[u]);
      if (MSRC == 1) load8<T>(y + r * C + c, yv[u]);
      if (MSRC == 2) mb[u] = mask[r * cv + m.cg];
    }
#pragma unroll
    for (int u = 0; u < kBnU; ++u) {
      const int64_t r = r0 + u * m.lanes;
      if (r >= rows) break;
      float o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xh = af.xhat(xv[u][k], k);
        bool on = true;
        if (relu) on = MSRC == 2 ? ((mb[u] >> k) & 1u) != 0u : (MSRC == 1 ? yv[u][k] > 0.f : af.apply(xv[u][k], k) > 0.f);
        const float dz = on ? dv[u][k] : 0.f;
        dv[u][k] = dz;
        o[k] = gi[k] * (dz - kb[k] - xh * kg[k]);
      }
      store8<T>(dx + r * C + c, o);
      if (DRES) store8<T>(dres + r * C + c, dv[u]);
    }
  }
}

// The same for the layer in front of a 3x3 / 2 / 1 max-pool, generic geometry (odd H or W): the incoming gradient is
// gathered from the pooled gradient (pooled_dy8_k3s2p1); the mask is recomputed from z.
template <typename T>
__global__ void bn_apply_bwd_pool_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                         const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                         const float* __restrict__ dbeta, T* __restrict__ dx,
                                         int64_t rows, int C, int relu, int training, float inv_rows,
                                         const float* __restrict__ beta, PoolGeom pg) {
  const int cv = C >> 3;
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < rows * cv; it += gs) {
    const int c = (int)(it % cv) << 3;
    const int64_t off = (it / cv) * C + c;
    float dv[8], xv[8], dg[8], db[8], o[8];
    const int64_t r = it / cv;
    const int w = (int)(r % pg.W), h = (int)((r / pg.W) % pg.H);
    pooled_dy8_k3s2p1<T>(dy, pg.idx, r / ((int64_t)pg.W * pg.H), h, w, c, C, pg.Ho, pg.Wo, dv);
    load8<T>(x + off, xv);
    BnAffine<T> af;
    af.init(mean, invstd, gamma, beta, c, C);
    load8<float>(dgamma + c, dg); load8<float>(dbeta + c, db);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float xh = af.xhat(xv[k], k);
      const float dz = (relu && !(af.apply(xv[k], k) > 0.f)) ? 0.f : dv[k];
      o[k] = training ? af.g[k] * af.is[k] * (dz - db[k] * inv_rows - xh * dg[k] * inv_rows) : af.g[k] * af.is[k] * dz;
    }
    store8<T>(dx + off, o);
  }
}

// bn_apply_bwd<POOL> for even H and W, one thread per 2 x 2 input pixels (and 8 channels): the four pooling windows that
// can have selected any of them -- (qh, qw) .. (qh+1, qw+1) -- are loaded once for the quad instead of four (clamped)
// windows per pixel; an even row can only be tap row 1 of window qh, an odd row tap row 2 of qh or tap row 0 of qh+1.
// One 2 x 2 quad of input pixels (2qh + ph, 2qw + pw) of a 3x3 / 2 / 1 max-pool (even H and W): the four pooling windows
// that can have selected any of them are (qh + a, qw + b), a, b in {0, 1} -- loaded once for the quad; an even row can only
// be tap row 1 of window qh, an odd row tap row 2 of window qh or tap row 0 of window qh + 1 (columns alike).
template <typename T>
struct PoolQuad {
  unsigned long long pk[2][2];
  float v[2][2][8];
  bool va[2], vb[2];
  __device__ __forceinline__ void load(const T* __restrict__ dyp, const PoolGeom& pg, int64_t n, int qh, int qw, int c, int C) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int hc = min(qh + a, pg.Ho - 1), wc = min(qw + b, pg.Wo - 1);
        const int64_t o = ((n * pg.Ho + hc) * pg.Wo + wc) * C + c;
        pk[a][b] = *reinterpret_cast<const unsigned long long*>(pg.idx + o);
        load8<T>(dyp + o, v[a][b]);
      }
    va[0] = true; va[1] = qh + 1 < pg.Ho;
    vb[0] = true; vb[1] = qw + 1 < pg.Wo;
  }
  // gradient of pixel (ph, pw) of the quad
  template <int PH, int PW>
  __device__ __forceinline__ void grad(float (&dv)[8]) const {
#pragma unroll
    for (int e = 0; e < 8; ++e) dv[e] = 0.f;
#pragma unroll
    for (int a = 0; a <= PH; ++a)
#pragma unroll
      for (int b = 0; b <= PW; ++b) {
        if (!(va[a] && vb[b])) continue;
        const unsigned ki = PH == 0 ? 1u : (a == 0 ? 2u : 0u), kj = PW == 0 ? 1u : (b == 0 ? 2u : 0u);
        const unsigned tap = ki * 3 + kj;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (((pk[a][b] >> (8 * e)) & 0xffu) == tap) dv[e] += v[a][b][e];
      }
  }
};

// bn_colstats<MODE 1, POOL> for even H and W in the quad form: partial[b][0][c] = sum dz, partial[b][1][c] = sum dz * xhat over
// the block's quads (dz = the gathered pool gradient under the recomputed ReLU mask).  Same block shape as bn_colstats_kernel.
template <typename T>
__global__ __launch_bounds__(256) void bn_colstats_pool_quad_kernel(const T* __restrict__ x, const T* __restrict__ dyp,
                                                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    int64_t quads, int C, int quads_per_block, int relu, int vc_log2,
                                                                    float* __restrict__ partial, PoolGeom pg) {
  __shared__ float red[2][256][8];
  const int vc = 1 << vc_log2, nrl = 256 >> vc_log2;
  const int cl = threadIdx.x & (vc - 1), rl = threadIdx.x >> vc_log2;
  const int c = (blockIdx.x * vc + cl) * 8;
  const int H2 = pg.H >> 1, W2 = pg.W >> 1;
  const int64_t q0 = (int64_t)blockIdx.y * quads_per_block, q1 = min(quads, q0 + quads_per_block);
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c < C) {
    BnAffine<T> af;
    af.init(mean, invstd, gamma, beta, c, C);
    for (int64_t q = q0 + rl; q < q1; q += nrl) {
      const int qw = (int)(q % W2), qh = (int)((q / W2) % H2);
      const int64_t n = q / ((int64_t)W2 * H2);
      PoolQuad<T> pq;
      pq.load(dyp, pg, n, qh, qw, c, C);
      float xv[2][2][8];
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int pw = 0; pw < 2; ++pw)
          load8<T>(x + ((n * pg.H + 2 * qh + ph) * pg.W + 2 * qw + pw) * C + c, xv[ph][pw]);
      auto acc = [&](const float (&dv)[8], const float (&xq)[8]) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float xh = af.xhat(xq[k], k);
          const float dz = (relu && !(af.apply(xq[k], k) > 0.f)) ? 0.f : dv[k];
          a[k] += dz;
          b[k] = fmaf(dz, xh, b[k]);
        }
      };
      float dv[8];
      pq.template grad<0, 0>(dv); acc(dv, xv[0][0]);
      pq.template grad<0, 1>(dv); acc(dv, xv[0][1]);
      pq.template grad<1, 0>(dv); acc(dv, xv[1][0]);
      pq.template grad<1, 1>(dv); acc(dv, xv[1][1]);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[0][threadIdx.x][k] = a[k]; red[1][threadIdx.x][k] = b[k]; }
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * vc * 8; t += 256) {
    const int st = t / (vc * 8), col = t % (vc * 8), cvi = col >> 3, k = col & 7;
    const int cc = (blockIdx.x * vc + cvi) * 8 + k;
    if (cc < C) {
      float acc2 = 0.f;
      for (int r = 0; r < nrl; ++r) acc2 += red[st][(r << vc_log2) + cvi][k];
      partial[((int64_t)blockIdx.y * 2 + st) * C + cc] = acc2;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_bwd_pool_quad_kernel(const T* __restrict__ dyp, const T* __restrict__ x,
                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                              const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                              T* __restrict__ dx, int64_t N, int C, int relu, int training, float inv_rows,
                                              PoolGeom pg) {
  const int cv = C >> 3, H2 = pg.H >> 1, W2 = pg.W >> 1;
  const int64_t quads = N * H2 * W2;
  const RowMap m = bn_row_map(cv);                       // "row" = one 2 x 2 quad of input pixels
  if (m.row >= m.lanes) return;
  const int c = m.cg << 3;
  BnAffine<T> af;
  af.init(mean, invstd, gamma, beta, c, C);
  float gi[8], kb[8], kg[8];
  {
    float dg[8], db[8];
    load8<float>(dgamma + c, dg); load8<float>(dbeta + c, db);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      gi[k] = af.g[k] * af.is[k];
      kb[k] = training ? db[k] * inv_rows : 0.f;
      kg[k] = training ? dg[k] * inv_rows : 0.f;
    }
  }
  for (int64_t q = m.row; q < quads; q += m.lanes) {
    const int qw = (int)(q % W2), qh = (int)((q / W2) % H2);
    const int64_t n = q / ((int64_t)W2 * H2);
    PoolQuad<T> pq;
    pq.load(dyp, pg, n, qh, qw, c, C);
    float xv[2][2][8];
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int pw = 0; pw < 2; ++pw)
        load8<T>(x + ((n * pg.H + 2 * qh + ph) * pg.W + 2 * qw + pw) * C + c, xv[ph][pw]);
    auto emit = [&](const float (&dv)[8], const float (&xq)[8], int ph, int pw) {
      float o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xh = af.xhat(xq[k], k);
        const float dz = (relu && !(af.apply(xq[k], k) > 0.f)) ? 0.f : dv[k];
        o[k] = gi[k] * (dz - kb[k] - xh * kg[k]);
      }
      store8<T>(dx + ((n * pg.H + 2 * qh + ph) * pg.W + 2 * qw + pw) * C + c, o);
    };
    float dv[8];
    pq.template grad<0, 0>(dv); emit(dv, xv[0][0], 0, 0);
    pq.template grad<0, 1>(dv); emit(dv, xv[0][1], 0, 1);
    pq.template grad<1, 0>(dv); emit(dv, xv[1][0], 1, 0);
    pq.template grad<1, 1>(dv); emit(dv, xv[1][1], 1, 1);
  }
}

// ------------------------------------------------------------------ scalar fallbacks (C % 8 != 0)
// R(2+1)D mid-plane counts (45, 230, 460, 921) are not multiples of 8.  One thread per column.
template <typename T, int MODE>
__global__ void bn_colstats_scalar_kernel(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                          int64_t rows, int C, int rows_per_block, int relu,
                                          float* __restrict__ partial, const float* __restrict__ gamma = nullptr,
                                          const float* __restrict__ beta = nullptr) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float a = 0.f, b = 0.f;
  const float mu = MODE == 1 ? mean[c] : 0.f, is = MODE == 1 ? invstd[c] : 0.f;
  for (int64_t r = r0; r < r1; ++r) {
    const float xv = to_f32<T>(x[r * C + c]);
    if (MODE == 0) { a += xv; b = fmaf(xv, xv, b); }
    else {
      float dz = to_f32<T>(dy[r * C + c]);
      const float yv = y ? to_f32<T>(y[r * C + c]) : (relu ? fmaf((xv - mu) * is, gamma[c], beta[c]) : 1.f);
      if (relu && !(yv > 0.f)) dz = 0.f;
      a += dz;
      b = fmaf(dz, (xv - mu) * is, b);
    }
  }
  partial[((int64_t)blockIdx.y * 2 + 0) * C + c] = a;
  partial[((int64_t)blockIdx.y * 2 + 1) * C + c] = b;
}

template <typename T>
__global__ void bn_apply_fwd_scalar_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                           const float* __restrict__ beta, const T* __restrict__ residual,
                                           T* __restrict__ y, int64_t rows, int C, int relu) {
  const int64_t total = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    float v = fmaf((to_f32<T>(x[i]) - mean[c]) * invstd[c], gamma[c], beta[c]);
    if (residual) v += to_f32<T>(residual[i]);
    y[i] = from_f32<T>(relu ? fmaxf(v, 0.f) : v);
  }
}

template <typename T>
__global__ void bn_apply_bwd_scalar_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                           const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                           const float* __restrict__ dbeta, T* __restrict__ dx, T* __restrict__ dres,
                                           int64_t rows, int C, int relu, int training, float inv_rows,
                                           const float* __restrict__ beta = nullptr) {
  const int64_t total = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    float dz = to_f32<T>(dy[i]);
    const float xh = (to_f32<T>(x[i]) - mean[c]) * invstd[c];
    const float yv = y ? to_f32<T>(y[i]) : (relu ? fmaf(xh, gamma[c], beta[c]) : 1.f);
    if (relu && !(yv > 0.f)) dz = 0.f;
    const float o = training ? gamma[c] * invstd[c] * (dz - dbeta[c] * inv_rows - xh * dgamma[c] * inv_rows)
                             : gamma[c] * invstd[c] * dz;
    dx[i] = from_f32<T>(o);
    if (dres) dres[i] = from_f32<T>(dz);
  }
}

template <typename T, typename D, bool NCHW>
__global__ void col2im_scalar_kernel(const T* __restrict__ dcol, D* __restrict__ dx, int N, int C, int H, int W,
                                     int kh, int kw, int sh, int sw, int ph, int pw, int Ho, int Wo, int64_t ld) {
  const int64_t total = (int64_t)N * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c, w, h;
    int64_t n;
    if (NCHW) {   // i enumerates dx in NCHW order
      w = (int)(i % W); h = (int)((i / W) % H); c = (int)((i / ((int64_t)W * H)) % C); n = i / ((int64_t)W * H * C);
    } else {
      c = (int)(i % C);
      const int64_t px = i / C;
      w = (int)(px % W); h = (int)((px / W) % H); n = px / ((int64_t)W * H);
    }
    // Only the taps with (h + ph - ki) % sh == 0 contribute: ki = ki0, ki0 + sh, ... -- walked directly, and the (at most
    // four) taps of a row requested together before they are added (same order of additions as the tap-by-tap form, whose
    // `continue`s made every load wait for the one before: 34 us for the 0.9 M input-gradient elements of the
    // frametransformer's CLS chunks)
    float acc = 0.f;
    const int ki0 = (h + ph) % sh, kj0 = (w + pw) % sw;
    for (int ki = ki0; ki < kh; ki += sh) {
      const int hh = h + ph - ki;
      if (hh < 0) break;
      const int ho = hh / sh;
      if (ho >= Ho) continue;
      const T* row = dcol + ((n * Ho + ho) * Wo) * ld + (int64_t)(ki * kw) * C + c;
      for (int kj = kj0; kj < kw; kj += 4 * sw) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kk = kj + u * sw, ww = w + pw - kk;
          const int wo = ww / sw;
          const bool ok = kk < kw && ww >= 0 && wo < Wo;
          v[u] = ok ? to_f32<T>(row[(int64_t)wo * ld + (int64_t)kk * C]) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
      }
    }
    dx[i] = from_f32<D>(acc);
  }
}

// ------------------------------------------------------------------ max pool (NHWC), first max wins
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, unsigned char* __restrict__ idx,
                                   int N, int C, int H, int W, int k, int stride, int pad, int Ho, int Wo) {
  const int64_t total = (int64_t)N * Ho * Wo * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t r = i / C;
    const int wo = (int)(r % Wo), ho = (int)((r / Wo) % Ho);
    const int64_t n = r / ((int64_t)Wo * Ho);
    float best = -INFINITY;
    int bi = -1;                        // first maximum in (ki, kj) scan order wins (torch semantics)
    for (int ki = 0; ki < k; ++ki)
      for (int kj = 0; kj < k; ++kj) {
        const int h = ho * stride - pad + ki, w = wo * stride - pad + kj;
        if (h < 0 || h >= H || w < 0 || w >= W) continue;
        const float v = to_f32<T>(x[((n * H + h) * W + w) * C + c]);
        if (bi < 0 || v > best) { best = v; bi = ki * k + kj; }
      }
    y[i] = from_f32<T>(best);
    idx[i] = (unsigned char)(bi < 0 ? 0 : bi);
  }
}

template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ idx,
                                   T* __restrict__ dx, int N, int C, int H, int W, int k, int stride, int pad,
                                   int Ho, int Wo) {
  const int64_t total = (int64_t)N * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t px = i / C;
    const int w = (int)(px % W), h = (int)((px / W) % H);
    const int64_t n = px / ((int64_t)W * H);
    float acc = 0.f;
    for (int ki = 0; ki < k; ++ki) {
      const int hh = h + pad - ki;
      if (hh < 0 || hh % stride) continue;
      const int ho = hh / stride;
      if (ho >= Ho) continue;
      for (int kj = 0; kj < k; ++kj) {
        const int ww = w + pad - kj;
        if (ww < 0 || ww % stride) continue;
        const int wo = ww / stride;
        if (wo >= Wo) continue;
        const int64_t o = ((n * Ho + ho) * Wo + wo) * C + c;
        if (idx[o] == ki * k + kj) acc += to_f32<T>(dy[o]);
      }
    }
    dx[i] = from_f32<T>(acc);
  }
}

// vectorised forms (C % 8 == 0): one thread owns 8 channels of one pixel, 16-byte loads, 8-byte index stores
template <typename T>
__global__ void maxpool_fwd_vec_kernel(const T* __restrict__ x, T* __restrict__ y, unsigned char* __restrict__ idx,
                                       int N, int C, int H, int W, int k, int stride, int pad, int Ho, int Wo) {
  const int cv = C >> 3;
  const int64_t items = (int64_t)N * Ho * Wo * cv;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(it % cv) << 3;
    const int64_t r = it / cv;
    const int wo = (int)(r % Wo), ho = (int)((r / Wo) % Ho);
    const int64_t n = r / ((int64_t)Wo * Ho);
    float best[8];
    int bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = -1; }
    for (int ki = 0; ki < k; ++ki) {
      const int h = ho * stride - pad + ki;
      if (h < 0 || h >= H) continue;
      for (int kj = 0; kj < k; ++kj) {
        const int w = wo * stride - pad + kj;
        if (w < 0 || w >= W) continue;
        float v[8];
        load8<T>(x + ((n * H + h) * W + w) * C + c, v);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (bi[e] < 0 || v[e] > best[e]) { best[e] = v[e]; bi[e] = ki * k + kj; }
      }
    }
    store8<T>(y + r * C + c, best);
    unsigned long long packed = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) packed |= (unsigned long long)(unsigned char)(bi[e] < 0 ? 0 : bi[e]) << (8 * e);
    *reinterpret_cast<unsigned long long*>(idx + r * C + c) = packed;
  }
}

template <typename T>
__global__ void maxpool_bwd_vec_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ idx,
                                       T* __restrict__ dx, int N, int C, int H, int W, int k, int stride, int pad,
                                       int Ho, int Wo) {
  const int cv = C >> 3;
  const int64_t items = (int64_t)N * H * W * cv;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(it % cv) << 3;
    const int64_t px = it / cv;
    const int w = (int)(px % W), h = (int)((px / W) % H);
    const int64_t n = px / ((int64_t)W * H);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int ki = 0; ki < k; ++ki) {
      const int hh = h + pad - ki;
      if (hh < 0 || hh % stride) continue;
      const int ho = hh / stride;
      if (ho >= Ho) continue;
      for (int kj = 0; kj < k; ++kj) {
        const int ww = w + pad - kj;
        if (ww < 0 || ww % stride) continue;
        const int wo = ww / stride;
        if (wo >= Wo) continue;
        const int64_t o = ((n * Ho + ho) * Wo + wo) * C + c;
        const unsigned long long packed = *reinterpret_cast<const unsigned long long*>(idx + o);
        float v[8];
        load8<T>(dy + o, v);
        const unsigned tap = (unsigned)(ki * k + kj);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (((packed >> (8 * e)) & 0xffu) == tap) acc[e] += v[e];
      }
    }
    store8<T>(dx + px * C + c, acc);
  }
}

// The ResNet stem pool (kernel 3, stride 2, padding 1: custom_resnet.py:107) in straight-line form: an input pixel lies in
// one window per axis when its coordinate is even (tap 1) and in two when it is odd (taps 0 and 2), so at most four
// windows can have chosen it.  All four (dy, argmax) pairs are requested up front with clamped coordinates and the
// invalid ones masked afterwards -- the generic kernel's tap loops with their divisions and early exits issue one
// dependent load pair at a time (393 us for the 256-frame stem map against 113 us of traffic).
// gradient of input pixel (n, h, w), channels c .. c+7, of a 3x3 / stride 2 / pad 1 max-pool from the pooled gradient dy
// and the stored argmax taps
template <typename T>
__device__ __forceinline__ void pooled_dy8_k3s2p1(const T* __restrict__ dy, const unsigned char* __restrict__ idx, int64_t n,
                                                  int h, int w, int c, int C, int Ho, int Wo, float* acc) {
  // axis candidates: (window, tap); odd coordinate: ((x+1)/2, tap 0) and ((x-1)/2, tap 2); even: (x/2, tap 1)
  const int hoA = (h + 1) >> 1, hoB = (h - 1) >> 1, woA = (w + 1) >> 1, woB = (w - 1) >> 1;
  const bool hodd = h & 1, wodd = w & 1;
  const int ho[2] = {hodd ? hoA : (h >> 1), hoB}, kih[2] = {hodd ? 0 : 1, 2};
  const int wo[2] = {wodd ? woA : (w >> 1), woB}, kjw[2] = {wodd ? 0 : 1, 2};
  const bool hv[2] = {ho[0] < Ho, hodd && hoB >= 0}, wv[2] = {wo[0] < Wo, wodd && woB >= 0};
  unsigned long long pk[4];
  float v[4][8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int hc = min(max(ho[a], 0), Ho - 1), wc = min(max(wo[b], 0), Wo - 1);
      const int64_t o = ((n * Ho + hc) * Wo + wc) * C + c;
      pk[a * 2 + b] = *reinterpret_cast<const unsigned long long*>(idx + o);
      load8<T>(dy + o, v[a * 2 + b]);
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (!(hv[a] && wv[b])) continue;
      const unsigned tap = (unsigned)(kih[a] * 3 + kjw[b]);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (((pk[a * 2 + b] >> (8 * e)) & 0xffu) == tap) acc[e] += v[a * 2 + b][e];
    }
}

template <typename T>
__global__ void maxpool_bwd_k3s2p1_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ idx,
                                          T* __restrict__ dx, int N, int C, int H, int W, int Ho, int Wo) {
  const int cv = C >> 3;
  const int64_t items = (int64_t)N * H * W * cv;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(it % cv) << 3;
    const int64_t px = it / cv;
    const int w = (int)(px % W), h = (int)((px / W) % H);
    const int64_t n = px / ((int64_t)W * H);
    float acc[8];
    pooled_dy8_k3s2p1<T>(dy, idx, n, h, w, c, C, Ho, Wo, acc);
    store8<T>(dx + px * C + c, acc);
  }
}

// BatchNorm + ReLU + 3x3 / stride 2 / pad 1 max-pool in one pass over z (the ResNet stem, custom_resnet.py:100-105,138-142):
// the normalised map (411 MB at 256 frames of 224^2) is neither written nor read back.  Values are rounded to T before the
// comparison, ties go to the first tap: the same output and argmax as bn_apply_fwd followed by maxpool_fwd.
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_k3s2p1_kernel(const T* __restrict__ z, const float* __restrict__ mean,
                                              const float* __restrict__ invstd, const float* __restrict__ gamma,
                                              const float* __restrict__ beta, T* __restrict__ y,
                                              unsigned char* __restrict__ idx, int N, int C, int H, int W, int Ho, int Wo,
                                              int relu) {
  const int cv = C >> 3;
  const int64_t outs = (int64_t)N * Ho * Wo;
  const RowMap m = bn_row_map(cv);                       // "row" = one pooled output pixel
  if (m.row >= m.lanes) return;
  const int c = m.cg << 3;
  BnAffine<T> af;
  af.init(mean, invstd, gamma, beta, c, C);
  // 16-bit maps under a ReLU (the stem: the only caller in the workloads): the launch is bound by VECTOR work (3.2 M outputs x
  // 64 channels x 9 taps of affine + ReLU + rounding + compare / select: 154 us against ~100 us of HBM time), so the maximum
  // and its tap are found as ONE unsigned key per element: (rounded value's 16 bits << 16) | (15 - tap).  Non-negative
  // 16-bit floats order like unsigned integers, equal values prefer the earlier tap through the low bits, taps outside the
  // image are key 0 and lose to every tap inside (low bits >= 7).  Same output and argmax as the compare form below.
  if constexpr (sizeof(T) == 2) if (relu) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef short i16x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((ext_vector_type(2))) T T2;
    for (int64_t r = m.row; r < outs; r += m.lanes) {
      const int wo = (int)(r % Wo), ho = (int)((r / Wo) % Ho);
      const int64_t n = r / ((int64_t)Wo * Ho);
      typedef __attribute__((ext_vector_type(8))) T T8;
      T8 v[9];
#pragma unroll
      for (int ki = 0; ki < 3; ++ki)
#pragma unroll
        for (int kj = 0; kj < 3; ++kj) {
          const int h = min(max(ho * 2 - 1 + ki, 0), H - 1), w = min(max(wo * 2 - 1 + kj, 0), W - 1);
          v[ki * 3 + kj] = *reinterpret_cast<const T8*>(z + ((n * H + h) * W + w) * C + c);
        }
      unsigned best[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
      for (int ki = 0; ki < 3; ++ki)
#pragma unroll
        for (int kj = 0; kj < 3; ++kj) {
          const int tap = ki * 3 + kj;
          const int h = ho * 2 - 1 + ki, w = wo * 2 - 1 + kj;
          const bool ok = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
          const unsigned low = ok ? 15u - tap : 0u;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x2 x = {(float)v[tap][2 * e], (float)v[tap][2 * e + 1]};
            const f32x2 y = __builtin_elementwise_fma(x, f32x2{af.s[2 * e], af.s[2 * e + 1]}, f32x2{af.t[2 * e], af.t[2 * e + 1]});
            T2 q = {(T)y[0], (T)y[1]};
            q = __builtin_bit_cast(T2, __builtin_elementwise_max(__builtin_bit_cast(i16x2, q), i16x2{0, 0}));   // ReLU on the rounded pair
            unsigned pk = __builtin_bit_cast(unsigned, q);
            pk = ok ? pk : 0u;
            best[2 * e] = max(best[2 * e], (pk << 16) | low);
            best[2 * e + 1] = max(best[2 * e + 1], (pk & 0xFFFF0000u) | low);
          }
        }
      T8 o;
      unsigned long long packed = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o[e] = __builtin_bit_cast(T, (unsigned short)(best[e] >> 16));
        packed |= (unsigned long long)(15u - (best[e] & 15u)) << (8 * e);
      }
      *reinterpret_cast<T8*>(y + r * C + c) = o;
      *reinterpret_cast<unsigned long long*>(idx + r * C + c) = packed;
    }
    return;
  }
  for (int64_t r = m.row; r < outs; r += m.lanes) {
    const int wo = (int)(r % Wo), ho = (int)((r / Wo) % Ho);
    const int64_t n = r / ((int64_t)Wo * Ho);
    // all nine taps in flight at once (clamped coordinates; out-of-image taps are skipped below)
    float v[9][8];
#pragma unroll
    for (int ki = 0; ki < 3; ++ki)
#pragma unroll
      for (int kj = 0; kj < 3; ++kj) {
        const int h = min(max(ho * 2 - 1 + ki, 0), H - 1), w = min(max(wo * 2 - 1 + kj, 0), W - 1);
        load8<T>(z + ((n * H + h) * W + w) * C + c, v[ki * 3 + kj]);
      }
    float best[8];
    int bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = -1; }
#pragma unroll
    for (int ki = 0; ki < 3; ++ki)
#pragma unroll
      for (int kj = 0; kj < 3; ++kj) {
        const int h = ho * 2 - 1 + ki, w = wo * 2 - 1 + kj;
        const bool ok = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t = af.apply(v[ki * 3 + kj][e], e);
          if (relu) t = fmaxf(t, 0.f);
          t = to_f32<T>(from_f32<T>(t));
          if (ok && (bi[e] < 0 || t > best[e])) { best[e] = t; bi[e] = ki * 3 + kj; }
        }
      }
    store8<T>(y + r * C + c, best);
    unsigned long long packed = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) packed |= (unsigned long long)(unsigned char)(bi[e] < 0 ? 0 : bi[e]) << (8 * e);
    *reinterpret_cast<unsigned long long*>(idx + r * C + c) = packed;
  }
}

// NCHW frames -> NHWC rows of Cpad = 8 channels (C real, the rest zero): one 16-byte store per pixel, the planes read
// with unit stride across the lanes.
template <typename S, typename D>
__global__ void nchw_to_nhwc_pad8_kernel(const S* __restrict__ x, D* __restrict__ y, int64_t N, int C, int64_t HW) {
  const int64_t total = N * HW;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = it / HW, px = it - n * HW;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < C; ++c) v[c] = to_f32<S>(x[(n * C + c) * HW + px]);
    store8<D>(y + it * 8, v);
  }
}

// Cpad = 4: two horizontally adjacent pixels per 16-byte chunk (x0 c0 c1 c2 0 | x1 c0 c1 c2 0) -- the [N, H, W/2, 8] view
// in which a stride-2 stem is a stride-(2,1) convolution over pixel PAIRS (dvt_conv_weight_pairs).  W even.
template <typename S, typename D>
__global__ void nchw_to_nhwc_pair4_kernel(const S* __restrict__ x, D* __restrict__ y, int64_t N, int C, int64_t H, int64_t W) {
  const int64_t W2 = W >> 1, HW = H * W, total = N * H * W2;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = it / (H * W2), r = it - n * (H * W2);
    const int64_t h = r / W2, w2 = r - h * W2;
    const int64_t px = h * W + 2 * w2;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < C; ++c) {
      v[c] = to_f32<S>(x[(n * C + c) * HW + px]);
      v[4 + c] = to_f32<S>(x[(n * C + c) * HW + px + 1]);
    }
    store8<D>(y + it * 8, v);
  }
}

// Stem weights in the pixel-pair form: wp[co, px*4 + c, ki, p] = w[co, c, ki, 2p - off + px] (zero outside 0 <= kj < kw,
// c < Cin); off = pw & 1 makes the first pixel of every pair an even input column.  bwd: the adjoint gather.
__global__ void conv_weight_pairs_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int kh,
                                         int kw, int kwp, int off) {
  const int total = Cout * 8 * kh * kwp;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int p = i % kwp, ki = (i / kwp) % kh, e = (i / (kwp * kh)) % 8, co = i / (kwp * kh * 8);
    const int px = e >> 2, c = e & 3, kj = 2 * p - off + px;
    wp[i] = (c < Cin && kj >= 0 && kj < kw) ? w[((co * Cin + c) * kh + ki) * kw + kj] : 0.f;
  }
}
__global__ void conv_weight_pairs_bwd_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Cout, int Cin, int kh,
                                             int kw, int kwp, int off, int accumulate) {
  const int total = Cout * Cin * kh * kw;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int kj = i % kw, ki = (i / kw) % kh, c = (i / (kw * kh)) % Cin, co = i / (kw * kh * Cin);
    const int p = (kj + off) >> 1, px = (kj + off) & 1;
    const float g = dwp[((co * 8 + px * 4 + c) * kh + ki) * kwp + p];
    dw[i] = accumulate ? dw[i] + g : g;
  }
}

// ------------------------------------------------------------------ batched 2-D transpose [B, R, Cc] -> [B, Cc, R]
template <typename T>
__global__ void transpose_kernel(const T* __restrict__ src, T* __restrict__ dst, int R, int Cc) {
  __shared__ T tile[32][33];
  const int64_t b = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 8 rows per pass
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    if (r < R && c < Cc) tile[j][tx] = src[(b * R + r) * Cc + c];
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (r < R && c < Cc) dst[(b * Cc + c) * R + r] = tile[tx][j];
  }
}

}  // namespace

extern "C" {

int dvt_im2col(const void* x, int x_dtype, int x_nchw, void* out, int out_dtype, int64_t N, int C, int H,
               int W, int kh, int kw, int sh, int sw, int ph, int pw, int64_t ld, dvt_stream_t stream) {
  DVT_REQUIRE(x && out && N >= 0 && C > 0 && H > 0 && W > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0,
              "dvt_im2col: bad arguments");
  DVT_REQUIRE(ld >= (int64_t)kh * kw * C, "dvt_im2col: ld smaller than kh*kw*C");
  const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
  DVT_REQUIRE(Ho > 0 && Wo > 0, "dvt_im2col: empty output");
  if (N == 0) return DVT_OK;
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = N * Ho * Wo;
  const bool vec = !x_nchw && x_dtype == out_dtype && C % 8 == 0 && ld % 8 == 0 && ld == (int64_t)kh * kw * C &&
                   dvt_aligned16(x) && dvt_aligned16(out);
  if (x_nchw && C == 3 && kh == 7 && kw == 7 && ld % 8 == 0 && ld < (1 << 20) && dvt_aligned16(out)) {
#define DVT_STEM_CASE(SD, S, DD, D)                                                                                   \
  if (x_dtype == SD && out_dtype == DD) {                                                                             \
    hipLaunchKernelGGL((im2col_nchw_stem_kernel<S, D, 3, 7, 7>), dim3(cgrid(rows * (ld >> 3))), dim3(kB), 0, st,      \
                       (const S*)x, (D*)out, (int)N, H, W, sh, sw, ph, pw, Ho, Wo, (int)ld);                           \
    DVT_LAUNCH_CHECK("dvt_im2col(stem)");                                                                             \
    return DVT_OK;                                                                                                    \
  }
    DVT_STEM_CASE(DVT_F32, float, DVT_F32, float)
    DVT_STEM_CASE(DVT_F32, float, DVT_BF16, bf16)
    DVT_STEM_CASE(DVT_F32, float, DVT_F16, f16)
    DVT_STEM_CASE(DVT_BF16, bf16, DVT_BF16, bf16)
    DVT_STEM_CASE(DVT_F16, f16, DVT_F16, f16)
    DVT_STEM_CASE(DVT_BF16, bf16, DVT_F32, float)
    DVT_STEM_CASE(DVT_F16, f16, DVT_F32, float)
#undef DVT_STEM_CASE
  }
  if (vec) {
    DVT_DISPATCH_DTYPE(x_dtype, T, hipLaunchKernelGGL((im2col_nhwc_vec_kernel<T>), dim3(cgrid(rows * kh * kw * (C >> 3))),
                                                      dim3(kB), 0, st, (const T*)x, (T*)out, (int)N, C, H, W, kh, kw,
                                                      sh, sw, ph, pw, Ho, Wo, ld));
  } else {
#define DVT_IM2COL_CASE(SD, S, DD, D)                                                                        \
  if (x_dtype == SD && out_dtype == DD) {                                                                    \
    if (x_nchw) hipLaunchKernelGGL((im2col_kernel<S, D, true>), dim3(cgrid(rows * ld)), dim3(kB), 0, st,     \
                                   (const S*)x, (D*)out, (int)N, C, H, W, kh, kw, sh, sw, ph, pw, Ho, Wo, ld);  \
    else hipLaunchKernelGGL((im2col_kernel<S, D, false>), dim3(cgrid(rows * ld)), dim3(kB), 0, st,           \
                            (const S*)x, (D*)out, (int)N, C, H, W, kh, kw, sh, sw, ph, pw, Ho, Wo, ld);         \
    DVT_LAUNCH_CHECK("dvt_im2col");                                                                          \
    return DVT_OK;                                                                                           \
  }
    DVT_IM2COL_CASE(DVT_F32, float, DVT_F32, float)
    DVT_IM2COL_CASE(DVT_F32, float, DVT_BF16, bf16)
    DVT_IM2COL_CASE(DVT_F32, float, DVT_F16, f16)
    DVT_IM2COL_CASE(DVT_BF16, bf16, DVT_BF16, bf16)
    DVT_IM2COL_CASE(DVT_F16, f16, DVT_F16, f16)
    DVT_IM2COL_CASE(DVT_BF16, bf16, DVT_F32, float)
    DVT_IM2COL_CASE(DVT_F16, f16, DVT_F32, float)
#undef DVT_IM2COL_CASE
    DVT_UNSUPPORTED("dvt_im2col: dtype pair (%d, %d)", x_dtype, out_dtype);
  }
  DVT_LAUNCH_CHECK("dvt_im2col");
  return DVT_OK;
}

int dvt_col2im_nchw(const void* dcol, int dtype, void* dx, int dx_dtype, int64_t N, int C, int H, int W, int kh,
                    int kw, int sh, int sw, int ph, int pw, int64_t ld, dvt_stream_t stream) {
  DVT_REQUIRE(dcol && dx && N >= 0 && C > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0, "dvt_col2im_nchw: bad arguments");
  const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
  if (N == 0) return DVT_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(cgrid(N * H * W * C)), block(kB);
#define DVT_C2I_CASE(SD, S, DD, D)                                                                              \
  if (dtype == SD && dx_dtype == DD) {                                                                          \
    hipLaunchKernelGGL((col2im_scalar_kernel<S, D, true>), grid, block, 0, st, (const S*)dcol, (D*)dx, (int)N, C, H, \
                       W, kh, kw, sh, sw, ph, pw, Ho, Wo, ld);                                                  \
    DVT_LAUNCH_CHECK("dvt_col2im_nchw");                                                                        \
    return DVT_OK;                                                                                              \
  }
  DVT_C2I_CASE(DVT_F32, float, DVT_F32, float)
  DVT_C2I_CASE(DVT_BF16, bf16, DVT_F32, float)
  DVT_C2I_CASE(DVT_F16, f16, DVT_F32, float)
  DVT_C2I_CASE(DVT_BF16, bf16, DVT_BF16, bf16)
  DVT_C2I_CASE(DVT_F16, f16, DVT_F16, f16)
  DVT_C2I_CASE(DVT_F32, float, DVT_BF16, bf16)
  DVT_C2I_CASE(DVT_F32, float, DVT_F16, f16)
#undef DVT_C2I_CASE
  DVT_UNSUPPORTED("dvt_col2im_nchw: dtype pair (%d, %d)", dtype, dx_dtype);
}

int dvt_col2im(const void* dcol, void* dx, int64_t N, int C, int H, int W, int kh, int kw, int sh, int sw,
               int ph, int pw, int64_t ld, const void* add, int add_stride, int dtype, dvt_stream_t stream) {
  DVT_REQUIRE(dcol && dx && N >= 0 && C > 0 && add_stride >= 0, "dvt_col2im: bad arguments");
  const bool cvec = C % 8 == 0 && ld % 8 == 0 && dvt_aligned16(dcol) && dvt_aligned16(dx) && dvt_aligned16(add);
  if (add && !cvec) DVT_UNSUPPORTED("dvt_col2im: the fused second gradient path needs C %% 8 == 0 and 16-byte aligned buffers");
  DVT_REQUIRE(sh > 0 && sw > 0 && ph >= 0 && pw >= 0, "dvt_col2im: bad stride / padding");
  const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
  if (N == 0) return DVT_OK;
  hipStream_t st = (hipStream_t)stream;
  if (cvec) {
    DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((col2im_nhwc_kernel<T>), dim3(cgrid(N * H * W * (C >> 3))), dim3(kB), 0,
                                                    st, (const T*)dcol, (T*)dx, (int)N, C, H, W, kh, kw, sh, sw, ph, pw,
                                                    Ho, Wo, ld, (const T*)add, add_stride));
  } else {
    DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((col2im_scalar_kernel<T, T, false>), dim3(cgrid(N * H * W * C)), dim3(kB), 0,
                                                    st, (const T*)dcol, (T*)dx, (int)N, C, H, W, kh, kw, sh, sw, ph, pw,
                                                    Ho, Wo, ld));
  }
  DVT_LAUNCH_CHECK("dvt_col2im");
  return DVT_OK;
}

int dvt_conv_weight_pack(const float* w, void* dst, int dst_dtype, int Cout, int Cin, int kh, int kw, int64_t ld,
                         dvt_stream_t stream) {
  DVT_REQUIRE(w && dst && Cout > 0 && Cin > 0 && ld >= (int64_t)Cin * kh * kw, "dvt_conv_weight_pack: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  DVT_DISPATCH_DTYPE(dst_dtype, T, hipLaunchKernelGGL((weight_pack_kernel<T>), dim3(cgrid((int64_t)Cout * ld)), dim3(kB),
                                                      0, st, w, (T*)dst, Cout, Cin, kh, kw, ld));
  DVT_LAUNCH_CHECK("dvt_conv_weight_pack");
  return DVT_OK;
}

int dvt_conv_weight_pack_group(const dvt_pack_entry* entries, int count, dvt_stream_t stream) {
  DVT_REQUIRE(count >= 0 && (count == 0 || entries), "dvt_conv_weight_pack_group: bad arguments");
  for (int base = 0; base < count; base += kPackGroup) {
    PackGroup g{};
    int blocks = 0;
    g.n = count - base < kPackGroup ? count - base : kPackGroup;
    for (int i = 0; i < g.n; ++i) {
      const dvt_pack_entry& q = entries[base + i];
      DVT_REQUIRE(q.src && q.dst && q.cout_l > 0 && q.cin_l > 0 && q.kh > 0 && q.kw > 0 && q.cout_p >= q.cout_l &&
                      q.cin_p >= q.cin_l && q.kind >= 0 && q.kind <= 2 && (q.kind != 0 || q.ld >= q.kh * q.kw * q.cin_p) &&
                      (q.kind != 2 || (q.cls_sh > 0 && q.cls_sw > 0 && q.cls_rh >= 0 && q.cls_rh < q.cls_sh && q.cls_rw >= 0 &&
                                       q.cls_rw < q.cls_sw && pack_cls_nt(q.kh, q.cls_rh, q.cls_sh) > 0 &&
                                       pack_cls_nt(q.kw, q.cls_rw, q.cls_sw) > 0)) &&
                      (q.dtype == DVT_F32 || dvt_is_16bit(q.dtype)),
                  "dvt_conv_weight_pack_group: bad entry %d", base + i);
      g.e[i] = q;
      g.begin[i] = blocks;
      int rows_per_block;
      blocks += pack_plan(q, &rows_per_block);
    }
    g.begin[g.n] = blocks;
    hipLaunchKernelGGL(weight_pack_group_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
    DVT_LAUNCH_CHECK("dvt_conv_weight_pack_group");
  }
  return DVT_OK;
}

int dvt_pad3_f32(const float* src, float* dst, int A, int B, int K, int Ap, int Bp, dvt_stream_t stream) {
  DVT_REQUIRE(src && dst && A > 0 && B > 0 && K > 0 && Ap >= A && Bp >= B, "dvt_pad3_f32: bad arguments");
  hipLaunchKernelGGL(pad3_kernel, dim3(cgrid((int64_t)Ap * Bp * K)), dim3(kB), 0, (hipStream_t)stream, src, dst, A, B, K, Ap, Bp);
  DVT_LAUNCH_CHECK("dvt_pad3_f32");
  return DVT_OK;
}

int dvt_unpad3_f32(const float* src, float* dst, int A, int B, int K, int Bp, int accumulate, dvt_stream_t stream) {
  DVT_REQUIRE(src && dst && A > 0 && B > 0 && K > 0 && Bp >= B, "dvt_unpad3_f32: bad arguments");
  hipLaunchKernelGGL(unpad3_kernel, dim3(cgrid((int64_t)A * B * K)), dim3(kB), 0, (hipStream_t)stream, src, dst, A, B, K, Bp,
                     accumulate);
  DVT_LAUNCH_CHECK("dvt_unpad3_f32");
  return DVT_OK;
}

int dvt_conv_weight_unpack_grad_t(const float* gt, float* dw, int Cout, int Cin, int kh, int kw, int accumulate,
                                  dvt_stream_t stream) {
  DVT_REQUIRE(gt && dw && Cout > 0 && Cin > 0 && kh > 0 && kw > 0, "dvt_conv_weight_unpack_grad_t: bad arguments");
  hipLaunchKernelGGL(weight_unpack_t_kernel, dim3(cgrid((int64_t)Cout * Cin * kh * kw)), dim3(kB), 0, (hipStream_t)stream, gt,
                     dw, Cout, Cin, kh, kw, accumulate);
  DVT_LAUNCH_CHECK("dvt_conv_weight_unpack_grad_t");
  return DVT_OK;
}

int dvt_conv_weight_pack_dgrad(const float* w, void* dst, int dst_dtype, int Cout, int Cin, int kh, int kw,
                               dvt_stream_t stream) {
  DVT_REQUIRE(w && dst && Cout > 0 && Cin > 0 && kh > 0 && kw > 0, "dvt_conv_weight_pack_dgrad: bad arguments");
  DVT_DISPATCH_DTYPE(dst_dtype, T, hipLaunchKernelGGL((weight_pack_dgrad_kernel<T>), dim3(cgrid((int64_t)Cout * Cin * kh * kw)),
                                                      dim3(kB), 0, (hipStream_t)stream, w, (T*)dst, Cout, Cin, kh, kw));
  DVT_LAUNCH_CHECK("dvt_conv_weight_pack_dgrad");
  return DVT_OK;
}

int dvt_conv_weight_unpack_grad(const float* g, float* dw, int Cout, int Cin, int kh, int kw, int64_t ld,
                                int accumulate, dvt_stream_t stream) {
  DVT_REQUIRE(g && dw && Cout > 0 && Cin > 0 && ld >= (int64_t)Cin * kh * kw, "dvt_conv_weight_unpack_grad: bad arguments");
  hipLaunchKernelGGL(weight_unpack_kernel, dim3(cgrid((int64_t)Cout * Cin * kh * kw)), dim3(kB), 0, (hipStream_t)stream,
                     g, dw, Cout, Cin, kh, kw, ld, accumulate);
  DVT_LAUNCH_CHECK("dvt_conv_weight_unpack_grad");
  return DVT_OK;
}

size_t dvt_bn_workspace_bytes(int64_t rows, int C) {
  (void)rows;
  return (((size_t)2048 * 2 + 2) * (size_t)C + 8) * sizeof(float);
}

// Column-vector lanes per block (log2) of the vectorised column-statistics kernel.
static int bn_vc_log2(int C) {
  int l = 3;                       // 8 column vectors = 64 channels
  while (l < 5 && (16 << l) <= C) ++l;
  return l;
}

// Column vectors per block of bn_colstats_kernel: C / 8 when that is at most 32, else its largest divisor up to 32 (every
// lane busy); widths without a divisor of at least 8 keep the power-of-two form with a ragged last block column.
static int bn_vc(int C) {
  const int cv = C >> 3;
  if (C % 8 || cv <= 0) return 1 << bn_vc_log2(C);
  if (cv <= 32) return cv;
  for (int v = 32; v >= 8; --v)
    if (cv % v == 0) return v;
  return 1 << bn_vc_log2(C);
}

// Row partition: about 1024 workgroups in total (4 per CU), at least 4 passes of the row lanes per workgroup.  (2048 until
// round 5: the column-statistics kernel streams as fast with 1024 -- tools/dev/bn_colstats_price.hip: 5.9 TB/s either way -- and
// the finalize launch behind it, 37 of them per frametransformer step, reads half the partial rows: 7.5 -> ~5 us each;
// same box, frametransformer 15.98 -> 15.86 ms, pyramid 13.36 -> 13.29; 512 gives the gain back in the statistics kernel.)
static int bn_parts(int64_t rows, int C, int* rpb) {
  const int vc = bn_vc(C);
  const int64_t gx = dvt_cdiv(C, 8 * vc);
  const int nrl = 256 / vc;
  int64_t parts = 1024 / gx;
  const int64_t cap = rows / (4 * nrl);
  if (parts > cap) parts = cap;
  if (parts < 1) parts = 1;
  *rpb = (int)dvt_cdiv(rows, parts);
  return (int)dvt_cdiv(rows, *rpb);
}

int dvt_bn_stats(const void* x, float* mean, float* invstd, float* running_mean, float* running_var,
                 void* workspace, int64_t rows, int C, int c_valid, float eps, float momentum, int dtype, dvt_stream_t stream) {
  DVT_REQUIRE(x && mean && invstd && workspace && rows > 0 && C > 0 && c_valid <= C, "dvt_bn_stats: bad arguments");
  if (c_valid <= 0) c_valid = C;
  hipStream_t st = (hipStream_t)stream;
  int rpb;
  const int parts = bn_parts(rows, C, &rpb);
  const int vc = bn_vc(C);
  const dim3 grid((unsigned)dvt_cdiv(C, 8 * vc), (unsigned)parts);
  const dim3 grid_s((unsigned)dvt_cdiv(C, 256), (unsigned)parts);
  if (C % 8 == 0 && dvt_aligned16(x) && dvt_aligned16(workspace)) {
    DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((bn_colstats_kernel<T, 0>), grid, dim3(256), 0, st, (const T*)x,
                                                    (const T*)nullptr, (const T*)nullptr, (const float*)nullptr,
                                                    (const float*)nullptr, rows, C, rpb, 0, vc, (float*)workspace));
  } else {
    DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((bn_colstats_scalar_kernel<T, 0>), grid_s, dim3(256), 0, st, (const T*)x,
                                                    (const T*)nullptr, (const T*)nullptr, (const float*)nullptr,
                                                    (const float*)nullptr, rows, C, rpb, 0, (float*)workspace));
  }
  DVT_LAUNCH_CHECK("dvt_bn_stats");
  const float unbias = rows > 1 ? (float)rows / (float)(rows - 1) : 1.f;
  bn_finalize_launch<0>(st, (const float*)workspace, parts, C, 1.0f / (float)rows, eps, mean, invstd, running_mean,
                     running_var, momentum, unbias, 0, nullptr, nullptr, c_valid);
  DVT_LAUNCH_CHECK("dvt_bn_stats(finalize)");
  return DVT_OK;
}

int dvt_bn_stats_from_partials(float* partial, int64_t parts, float* mean, float* invstd, float* running_mean,
                               float* running_var, int64_t rows, int C, int c_valid, float eps, float momentum,
                               dvt_stream_t stream) {
  DVT_REQUIRE(partial && mean && invstd && parts > 0 && parts < (1ll << 31) && rows > 0 && C > 0 && c_valid <= C,
              "dvt_bn_stats_from_partials: bad arguments");
  if (c_valid <= 0) c_valid = C;
  const float unbias = rows > 1 ? (float)rows / (float)(rows - 1) : 1.0f;
  hipStream_t st = (hipStream_t)stream;
  const float* src = partial;
  int np = (int)parts;
  if (parts > 256) {   // (this is why `partial` is not const: its 64-row tail is written)
    // fold in place is not possible (blocks read what others write): the folded rows go behind the partial rows, which the
    // caller sized with dvt_conv2d_implicit_stats_bytes (parts + 64 rows)
    const int folds = 64;
    const int per_block = (int)dvt_cdiv(parts, folds);
    float* folded = partial + (size_t)parts * 2 * C;
    hipLaunchKernelGGL(bn_partial_fold_kernel, dim3((unsigned)dvt_cdiv(C, 32), (unsigned)folds), dim3(1024), 0, st, partial,
                       (int)parts, C, per_block, folded);
    DVT_LAUNCH_CHECK("dvt_bn_stats_from_partials(fold)");
    src = folded;
    np = folds;
  }
  bn_finalize_launch<0>(st, src, np, C,
                     1.0f / (float)rows, eps, mean, invstd, running_mean, running_var, momentum, unbias, 0, nullptr, nullptr,
                     c_valid);
  DVT_LAUNCH_CHECK("dvt_bn_stats_from_partials");
  return DVT_OK;
}

int dvt_bn_eval_invstd(const float* running_var, float* invstd, int C, float eps, dvt_stream_t stream) {
  DVT_REQUIRE(running_var && invstd && C > 0, "dvt_bn_eval_invstd: bad arguments");
  hipLaunchKernelGGL(rsqrt_eps_kernel, dim3((unsigned)dvt_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream,
                     running_var, invstd, C, eps);
  DVT_LAUNCH_CHECK("dvt_bn_eval_invstd");
  return DVT_OK;
}

// grid of a row-streaming kernel: a multiple of C/8 threads is not needed (bn_row_map idles the remainder), 8 workgroups/CU
int dvt_bn_apply_fwd(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                     const void* residual, void* y, void* relu_mask, int64_t rows, int C, int c_valid, int relu, int dtype,
                     dvt_stream_t stream) {
  DVT_REQUIRE(x && mean && invstd && gamma && beta && y && rows >= 0 && C > 0 && c_valid <= C, "dvt_bn_apply_fwd: bad arguments");
  if (rows == 0) return DVT_OK;
  if (c_valid <= 0) c_valid = C;
  hipStream_t st = (hipStream_t)stream;
  if (C % 8 == 0 && dvt_aligned16(x) && dvt_aligned16(y) && dvt_aligned16(residual) && dvt_aligned16(mean) &&
      dvt_aligned16(invstd) && dvt_aligned16(gamma) && dvt_aligned16(beta)) {
    const dim3 grid(cgrid(dvt_cdiv(rows, kBnU) * (C >> 3)));
#define DVT_BN_FWD(RES, MASK)                                                                                      \
  DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((bn_apply_fwd_kernel<T, RES, MASK>), grid, dim3(kB), 0, st, (const T*)x, mean, \
                                                  invstd, gamma, beta, (const T*)residual, (T*)y, (unsigned char*)relu_mask, \
                                                  rows, C, relu, c_valid))
    if (residual && relu_mask) DVT_BN_FWD(true, true);
    else if (residual) DVT_BN_FWD(true, false);
    else if (relu_mask) DVT_BN_FWD(false, true);
    else DVT_BN_FWD(false, false);
#undef DVT_BN_FWD
  } else {
    if (relu_mask || c_valid != C) DVT_UNSUPPORTED("dvt_bn_apply_fwd: the ReLU mask output / channel padding need C %% 8 == 0 and 16-byte aligned buffers");
    DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((bn_apply_fwd_scalar_kernel<T>), dim3(cgrid(rows * C)), dim3(kB), 0, st,
                                                    (const T*)x, mean, invstd, gamma, beta, (const T*)residual, (T*)y,
                                                    rows, C, relu));
  }
  DVT_LAUNCH_CHECK("dvt_bn_apply_fwd");
  return DVT_OK;
}

int dvt_bn_bwd(const void* dy, const void* x, const void* y, const void* relu_mask, const float* mean, const float* invstd,
               const float* gamma, const float* beta, void* dx, void* dres, float* dgamma, float* dbeta, void* workspace,
               int64_t rows, int C, int c_valid, int relu, int training, int accumulate, int dtype, dvt_stream_t stream) {
  DVT_REQUIRE(dy && x && mean && invstd && gamma && dx && dgamma && dbeta && workspace && rows > 0 && C > 0 && c_valid <= C,
              "dvt_bn_bwd: bad arguments");
  if (c_valid <= 0) c_valid = C;
  const bool cvec = C % 8 == 0 && dvt_aligned16(dy) && dvt_aligned16(x) && dvt_aligned16(y) && dvt_aligned16(dx) &&
                    dvt_aligned16(dres) && dvt_aligned16(mean) && dvt_aligned16(invstd) && dvt_aligned16(gamma) &&
                    dvt_aligned16(workspace);
  DVT_REQUIRE(!relu || y || relu_mask || (beta && !dres),
              "dvt_bn_bwd: relu needs the forward's mask bytes or output y, or beta to recompute the mask (no residual branch)");
  if ((relu_mask || c_valid != C) && !cvec)
    DVT_UNSUPPORTED("dvt_bn_bwd: the ReLU mask input / channel padding need C %% 8 == 0 and 16-byte aligned buffers");
  hipStream_t st = (hipStream_t)stream;
  int rpb;
  int parts = bn_parts(rows, C, &rpb);
  const int vcl = bn_vc(C);                            // (column vectors per block, not a logarithm any more)
  const dim3 grid((unsigned)dvt_cdiv(C, 8 * vcl), (unsigned)parts);
  const dim3 grid_s((unsigned)dvt_cdiv(C, 256), (unsigned)parts);
  // scratch: [parts][2][C] partials, then [2][C] for this launch's (not accumulated) dgamma/dbeta
  float* part = (float*)workspace;
  if (cvec) {
#define DVT_BN_CS(MSRC)                                                                                                      \
  DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((bn_colstats_kernel<T, 1, false, MSRC>), grid, dim3(256), 0, st, (const T*)x, \
                                                  (const T*)dy, (const T*)y, mean, invstd, rows, C, rpb, relu, vcl, part, gamma, \
                                                  beta, PoolGeom{nullptr, 0, 0, 0, 0}, (const unsigned char*)relu_mask, c_valid))
    if (relu && relu_mask) DVT_BN_CS(2);
    else if (relu && y) DVT_BN_CS(1);
    else DVT_BN_CS(0);
#undef DVT_BN_CS
  } else {
    DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((bn_colstats_scalar_kernel<T, 1>), grid_s, dim3(256), 0, st, (const T*)x,
                                                    (const T*)dy, (const T*)y, mean, invstd, rows, C, rpb, relu, part,
                                                    gamma, beta));
  }
  DVT_LAUNCH_CHECK("dvt_bn_bwd(stats)");
  // keep the local dgamma/dbeta 16-byte aligned behind the partials
  float* loc = part + (((size_t)parts * 2 * C + 3) & ~(size_t)3);
  // dgamma / dbeta are published (overwritten or accumulated) by the same launch that leaves this launch's own sums in `loc`
  bn_finalize_launch<1>(st, part, parts, C, 0.f, 0.f, loc, loc + C, (float*)nullptr, (float*)nullptr, 0.f, 0.f, accumulate, dgamma, dbeta,
                     c_valid);
  DVT_LAUNCH_CHECK("dvt_bn_bwd(finalize)");
  if (cvec) {
    const dim3 agrid(cgrid(dvt_cdiv(rows, kBnU) * (C >> 3)));
    const int msrc = !relu ? 0 : (relu_mask ? 2 : (y ? 1 : 0));
#define DVT_BN_BWD(MSRC, DRES)                                                                                               \
  DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((bn_apply_bwd_kernel<T, MSRC, DRES>), agrid, dim3(kB), 0, st, (const T*)dy,  \
                                                  (const T*)x, (const T*)y, (const unsigned char*)relu_mask, mean, invstd,   \
                                                  gamma, beta, loc, loc + C, (T*)dx, (T*)dres, rows, C, relu, training,      \
                                                  1.0f / (float)rows, c_valid))
    if (msrc == 2 && dres) DVT_BN_BWD(2, true);
    else if (msrc == 2) DVT_BN_BWD(2, false);
    else if (msrc == 1 && dres) DVT_BN_BWD(1, true);
    else if (msrc == 1) DVT_BN_BWD(1, false);
    else if (dres) DVT_BN_BWD(0, true);
    else DVT_BN_BWD(0, false);
#undef DVT_BN_BWD
  } else {
    DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((bn_apply_bwd_scalar_kernel<T>), dim3(cgrid(rows * C)), dim3(kB), 0, st,
                                                    (const T*)dy, (const T*)x, (const T*)y, mean, invstd, gamma, loc,
                                                    loc + C, (T*)dx, (T*)dres, rows, C, relu, training,
                                                    1.0f / (float)rows, beta));
  }
  DVT_LAUNCH_CHECK("dvt_bn_bwd(apply)");
  return DVT_OK;
}

int dvt_bn_relu_maxpool_fwd(const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                            void* y, void* idx, int64_t N, int C, int H, int W, int relu, int dtype, dvt_stream_t stream) {
  DVT_REQUIRE(z && mean && invstd && gamma && beta && y && idx && N >= 0 && C > 0 && H > 0 && W > 0,
              "dvt_bn_relu_maxpool_fwd: bad arguments");
  DVT_REQUIRE(C % 8 == 0 && dvt_aligned16(z) && dvt_aligned16(y) && ((uintptr_t)idx & 7) == 0 && dvt_aligned16(mean) &&
              dvt_aligned16(invstd) && dvt_aligned16(gamma) && dvt_aligned16(beta),
              "dvt_bn_relu_maxpool_fwd: needs C %% 8 == 0 and 16-byte aligned buffers");
  if (N == 0) return DVT_OK;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((bn_relu_maxpool_k3s2p1_kernel<T>), dim3(cgrid(N * Ho * Wo * (C >> 3))), dim3(kB), 0,
                                                  (hipStream_t)stream, (const T*)z, mean, invstd, gamma, beta, (T*)y,
                                                  (unsigned char*)idx, (int)N, C, H, W, Ho, Wo, relu));
  DVT_LAUNCH_CHECK("dvt_bn_relu_maxpool_fwd");
  return DVT_OK;
}

int dvt_bn_bwd_pooled(const void* dy_pool, const void* idx, const void* x, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, void* dx, float* dgamma, float* dbeta, void* workspace, int64_t N,
                      int C, int H, int W, int relu, int training, int accumulate, int dtype, dvt_stream_t stream) {
  DVT_REQUIRE(dy_pool && idx && x && mean && invstd && gamma && beta && dx && dgamma && dbeta && workspace && N > 0 && C > 0 &&
              H > 0 && W > 0, "dvt_bn_bwd_pooled: bad arguments");
  DVT_REQUIRE(C % 8 == 0 && dvt_aligned16(dy_pool) && dvt_aligned16(x) && dvt_aligned16(dx) && ((uintptr_t)idx & 7) == 0 &&
              dvt_aligned16(mean) && dvt_aligned16(invstd) && dvt_aligned16(gamma) && dvt_aligned16(beta) &&
              dvt_aligned16(workspace), "dvt_bn_bwd_pooled: needs C %% 8 == 0 and 16-byte aligned buffers");
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = N * H * W;
  PoolGeom pg{(const unsigned char*)idx, H, W, (H + 2 - 3) / 2 + 1, (W + 2 - 3) / 2 + 1};
  int rpb;
  const int parts = bn_parts(rows, C, &rpb);
  const int vcl = bn_vc_log2(C);
  const dim3 grid((unsigned)dvt_cdiv(C, 8 << vcl), (unsigned)parts);
  float* part = (float*)workspace;
  const bool quad = H % 2 == 0 && W % 2 == 0;
  if (quad) {
    DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((bn_colstats_pool_quad_kernel<T>), grid, dim3(256), 0, st, (const T*)x,
                                                    (const T*)dy_pool, mean, invstd, gamma, beta, rows / 4, C, (int)dvt_cdiv(rpb, 4), relu,
                                                    vcl, part, pg));
  } else {
    DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((bn_colstats_kernel<T, 1, true>), grid, dim3(256), 0, st, (const T*)x,
                                                    (const T*)dy_pool, (const T*)nullptr, mean, invstd, rows, C, rpb, relu,
                                                    1 << vcl, part, gamma, beta, pg));
  }
  DVT_LAUNCH_CHECK("dvt_bn_bwd_pooled(stats)");
  float* loc = part + (((size_t)parts * 2 * C + 3) & ~(size_t)3);
  // dgamma / dbeta are published (overwritten or accumulated) by the same launch that leaves this launch's own sums in `loc`
  bn_finalize_launch<1>(st, (const float*)part,
                     parts, C, 0.f, 0.f, loc, loc + C, (float*)nullptr, (float*)nullptr, 0.f, 0.f, accumulate, dgamma, dbeta);
  DVT_LAUNCH_CHECK("dvt_bn_bwd_pooled(finalize)");
  if (quad) {
    DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((bn_apply_bwd_pool_quad_kernel<T>), dim3(cgrid(rows / 4 * (C >> 3))), dim3(kB), 0,
                                                    st, (const T*)dy_pool, (const T*)x, mean, invstd, gamma, beta, loc, loc + C,
                                                    (T*)dx, N, C, relu, training, 1.0f / (float)rows, pg));
  } else {
    DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((bn_apply_bwd_pool_kernel<T>), dim3(cgrid(rows * (C >> 3))), dim3(kB), 0, st,
                                                    (const T*)dy_pool, (const T*)x, mean, invstd, gamma, loc, loc + C, (T*)dx,
                                                    rows, C, relu, training, 1.0f / (float)rows, beta, pg));
  }
  DVT_LAUNCH_CHECK("dvt_bn_bwd_pooled(apply)");
  return DVT_OK;
}

int dvt_maxpool_fwd(const void* x, void* y, void* idx, int64_t N, int C, int H, int W, int k, int stride, int pad,
                    int dtype, dvt_stream_t stream) {
  DVT_REQUIRE(x && y && idx && N >= 0 && C > 0 && k > 0 && k * k <= 255 && stride > 0, "dvt_maxpool_fwd: bad arguments");
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  if (N == 0) return DVT_OK;
  hipStream_t st = (hipStream_t)stream;
  if (C % 8 == 0 && dvt_aligned16(x) && dvt_aligned16(y) && ((uintptr_t)idx & 7) == 0) {
    DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((maxpool_fwd_vec_kernel<T>), dim3(cgrid(N * Ho * Wo * (C >> 3))), dim3(kB),
                                                    0, st, (const T*)x, (T*)y, (unsigned char*)idx, (int)N, C, H, W, k,
                                                    stride, pad, Ho, Wo));
  } else {
    DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((maxpool_fwd_kernel<T>), dim3(cgrid(N * Ho * Wo * C)), dim3(kB), 0, st,
                                                    (const T*)x, (T*)y, (unsigned char*)idx, (int)N, C, H, W, k, stride,
                                                    pad, Ho, Wo));
  }
  DVT_LAUNCH_CHECK("dvt_maxpool_fwd");
  return DVT_OK;
}

int dvt_maxpool_bwd(const void* dy, const void* idx, void* dx, int64_t N, int C, int H, int W, int k, int stride,
                    int pad, int dtype, dvt_stream_t stream) {
  DVT_REQUIRE(dy && idx && dx && N >= 0 && C > 0 && k > 0 && stride > 0, "dvt_maxpool_bwd: bad arguments");
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  if (N == 0) return DVT_OK;
  hipStream_t st = (hipStream_t)stream;
  if (C % 8 == 0 && dvt_aligned16(dy) && dvt_aligned16(dx) && ((uintptr_t)idx & 7) == 0 && k == 3 && stride == 2 && pad == 1) {
    DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((maxpool_bwd_k3s2p1_kernel<T>), dim3(cgrid(N * H * W * (C >> 3))), dim3(kB), 0,
                                                    st, (const T*)dy, (const unsigned char*)idx, (T*)dx, (int)N, C, H, W, Ho, Wo));
  } else if (C % 8 == 0 && dvt_aligned16(dy) && dvt_aligned16(dx) && ((uintptr_t)idx & 7) == 0) {
    DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((maxpool_bwd_vec_kernel<T>), dim3(cgrid(N * H * W * (C >> 3))), dim3(kB), 0,
                                                    st, (const T*)dy, (const unsigned char*)idx, (T*)dx, (int)N, C, H, W,
                                                    k, stride, pad, Ho, Wo));
  } else {
    DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((maxpool_bwd_kernel<T>), dim3(cgrid(N * H * W * C)), dim3(kB), 0, st,
                                                    (const T*)dy, (const unsigned char*)idx, (T*)dx, (int)N, C, H, W, k,
                                                    stride, pad, Ho, Wo));
  }
  DVT_LAUNCH_CHECK("dvt_maxpool_bwd");
  return DVT_OK;
}

int dvt_nchw_to_nhwc_pad(const void* x, int x_dtype, void* y, int y_dtype, int64_t N, int C, int H, int W, int Cpad,
                         dvt_stream_t stream) {
  DVT_REQUIRE(x && y && N >= 0 && C > 0 && H > 0 && W > 0, "dvt_nchw_to_nhwc_pad: bad arguments");
  DVT_REQUIRE((Cpad == 8 && C <= 8) || (Cpad == 4 && C <= 4 && W % 2 == 0),
              "dvt_nchw_to_nhwc_pad: Cpad must be 8 (C <= 8) or 4 (C <= 4, W even)");
  DVT_REQUIRE(dvt_aligned16(y) && dvt_is_16bit(y_dtype), "dvt_nchw_to_nhwc_pad: output must be 16-bit and 16-byte aligned");
  if (N == 0) return DVT_OK;
  hipStream_t st = (hipStream_t)stream;
  const int64_t HW = (int64_t)H * W;
  if (Cpad == 4) {
#define DVT_PAIR4(SD, S, DD, D)                                                                                      \
  if (x_dtype == SD && y_dtype == DD) {                                                                             \
    hipLaunchKernelGGL((nchw_to_nhwc_pair4_kernel<S, D>), dim3(cgrid(N * HW / 2)), dim3(kB), 0, st, (const S*)x, (D*)y, N, C, \
                       (int64_t)H, (int64_t)W);                                                                     \
    DVT_LAUNCH_CHECK("dvt_nchw_to_nhwc_pad");                                                                       \
    return DVT_OK;                                                                                                  \
  }
    DVT_PAIR4(DVT_F32, float, DVT_BF16, bf16)
    DVT_PAIR4(DVT_F32, float, DVT_F16, f16)
    DVT_PAIR4(DVT_BF16, bf16, DVT_BF16, bf16)
    DVT_PAIR4(DVT_F16, f16, DVT_F16, f16)
    DVT_PAIR4(DVT_BF16, bf16, DVT_F16, f16)
    DVT_PAIR4(DVT_F16, f16, DVT_BF16, bf16)
#undef DVT_PAIR4
    DVT_UNSUPPORTED("dvt_nchw_to_nhwc_pad: dtype pair (%d, %d) unsupported", x_dtype, y_dtype);
  }
#define DVT_PAD8(SD, S, DD, D)                                                                                      \
  if (x_dtype == SD && y_dtype == DD) {                                                                             \
    hipLaunchKernelGGL((nchw_to_nhwc_pad8_kernel<S, D>), dim3(cgrid(N * HW)), dim3(kB), 0, st, (const S*)x, (D*)y, N, C, HW); \
    DVT_LAUNCH_CHECK("dvt_nchw_to_nhwc_pad");                                                                       \
    return DVT_OK;                                                                                                  \
  }
  DVT_PAD8(DVT_F32, float, DVT_BF16, bf16)
  DVT_PAD8(DVT_F32, float, DVT_F16, f16)
  DVT_PAD8(DVT_BF16, bf16, DVT_BF16, bf16)
  DVT_PAD8(DVT_F16, f16, DVT_F16, f16)
  DVT_PAD8(DVT_BF16, bf16, DVT_F16, f16)
  DVT_PAD8(DVT_F16, f16, DVT_BF16, bf16)
#undef DVT_PAD8
  DVT_UNSUPPORTED("dvt_nchw_to_nhwc_pad: dtype pair (%d, %d) unsupported", x_dtype, y_dtype);
}

int dvt_conv_weight_pairs(const float* w, float* wp, int Cout, int Cin, int kh, int kw, int pw, int kwp, dvt_stream_t stream) {
  DVT_REQUIRE(w && wp && Cout > 0 && Cin > 0 && Cin <= 4 && kh > 0 && kw > 0 && pw >= 0 && kwp > 0,
              "dvt_conv_weight_pairs: bad arguments");
  DVT_REQUIRE(2 * kwp - (pw & 1) >= kw, "dvt_conv_weight_pairs: kwp pairs do not cover the kernel width");
  hipLaunchKernelGGL(conv_weight_pairs_kernel, dim3(cgrid((int64_t)Cout * 8 * kh * kwp)), dim3(kB), 0, (hipStream_t)stream, w, wp,
                     Cout, Cin, kh, kw, kwp, pw & 1);
  DVT_LAUNCH_CHECK("dvt_conv_weight_pairs");
  return DVT_OK;
}

int dvt_conv_weight_pairs_bwd(const float* dwp, float* dw, int Cout, int Cin, int kh, int kw, int pw, int kwp, int accumulate,
                              dvt_stream_t stream) {
  DVT_REQUIRE(dwp && dw && Cout > 0 && Cin > 0 && Cin <= 4 && kh > 0 && kw > 0 && pw >= 0 && kwp > 0,
              "dvt_conv_weight_pairs_bwd: bad arguments");
  DVT_REQUIRE(2 * kwp - (pw & 1) >= kw, "dvt_conv_weight_pairs_bwd: kwp pairs do not cover the kernel width");
  hipLaunchKernelGGL(conv_weight_pairs_bwd_kernel, dim3(cgrid((int64_t)Cout * Cin * kh * kw)), dim3(kB), 0, (hipStream_t)stream,
                     dwp, dw, Cout, Cin, kh, kw, kwp, pw & 1, accumulate);
  DVT_LAUNCH_CHECK("dvt_conv_weight_pairs_bwd");
  return DVT_OK;
}

int dvt_transpose_last2(const void* src, void* dst, int64_t B, int R, int Cc, int dtype, dvt_stream_t stream) {
  DVT_REQUIRE(src && dst && B >= 0 && R > 0 && Cc > 0 && B < 65536, "dvt_transpose_last2: bad arguments");
  if (B == 0) return DVT_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)dvt_cdiv(Cc, 32), (unsigned)dvt_cdiv(R, 32), (unsigned)B);
  DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((transpose_kernel<T>), grid, dim3(256), 0, st, (const T*)src, (T*)dst, R, Cc));
  DVT_LAUNCH_CHECK("dvt_transpose_last2");
  return DVT_OK;
}

}  // extern "C"
