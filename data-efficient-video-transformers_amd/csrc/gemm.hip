// gemm.hip -- the GEMM family behind every Linear on the clip path.
//
//   C[M,N] = epilogue( alpha * sum_k A(m,k) B(k,n) )
//
// bf16 fast path (gemm_mfma_kernel): 128x128x64 block tile, 4 waves (2x2, 64x64
// each = 4x4 v_mfma_f32_16x16x32_bf16 accumulators), fp32 accumulate.
//   * operands are staged HBM -> registers -> LDS (16-byte accesses), double
//     buffered: the global loads of tile k+1 are issued before the MFMAs of
//     tile k and written to the other LDS stage after them (one barrier / tile).
//   * an operand whose reduction index is contiguous in memory ("k-major":
//     activations x, weights W[N,K] in forward) is kept as [rows][64 k] with the
//     16-byte chunk index XOR-swizzled by (row & 7) and read with ds_read_b128.
//   * an operand whose reduction index is the row of the matrix ("mn-major":
//     W in dgrad, dY and x in wgrad) is kept as [64 k][128 mn] and read through
//     ds_read_b64_tr_b16 (hardware transpose); 32-byte units are XOR-swizzled by
//     (k&3)|((k>>3)&1)<<2 so that each 32-lane half hits 8 distinct bank groups.
//     No operand is ever transposed through HBM.
//   * the MFMA is issued as D = Bfrag x Afrag (C^T orientation) so that a lane
//     ends up with 4 consecutive n of one row m; the accumulators are staged
//     through LDS as fp32 and leave as whole 256-byte row segments, with
//     bias / GELU / GELU' / ReLU / residual applied on the way out.
//   * split-K (wgrad: M,N small, K = tokens) writes fp32 slabs that a second
//     kernel sums in a fixed order (bitwise reproducible, no atomics).
// generic path (gemm_generic_kernel): any dtype / shape / alignment, fp32 FMA,
// 64x64x16 tiles -- the fp32 parity mode and odd shapes (19-class head).
#include "common.h"
#include <type_traits>
#include "gemm_common.h"
#include <stdlib.h>

namespace {

enum { BM = 128, BN = 128, BK = 64, NTHREADS = 256 };
constexpr int kStageBytes = (BM * BK + BN * BK) * 2;   // 32 KiB
constexpr int kCPad = BN + 4;                           // fp32 staging row stride (floats)
constexpr int kSmemBytes = BM * kCPad * 4;              // 67,584 B >= 2 stages (65,536 B)

// ---- global -> registers (4 x 16 B per thread per operand tile)
template <bool KMAJOR>
__device__ __forceinline__ void g2r(const bf16* __restrict__ base, int64_t ld, int mn0, int mn_lim,
                                    int k0, int k_lim, int tid, bf16x8 (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + NTHREADS * i;
    int gk, gmn;
    int64_t off;
    if (KMAJOR) {
      gmn = mn0 + (idx >> 3);
      gk = k0 + ((idx & 7) << 3);
      off = (int64_t)gmn * ld + gk;
    } else {
      gk = k0 + (idx >> 4);
      gmn = mn0 + ((idx & 15) << 3);
      off = (int64_t)gk * ld + gmn;
    }
    const bool ok = (gmn < mn_lim) && (gk < k_lim);
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(base + (ok ? off : 0));
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    r[i] = ok ? v : z;
  }
}

// ---- registers -> LDS stage
template <bool KMAJOR>
__device__ __forceinline__ void r2s(char* tile, int tid, const bf16x8 (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + NTHREADS * i;
    int off;
    if (KMAJOR) {
      const int row = idx >> 3, c = idx & 7;
      off = row * (BK * 2) + ((c ^ (row & 7)) << 4);
    } else {
      const int k = idx >> 4, c = idx & 15;
      off = k * (BM * 2) + ((((c >> 1) ^ swz_mn(k)) << 5) | ((c & 1) << 4));
    }
    *reinterpret_cast<bf16x8*>(tile + off) = r[i];
  }
}

// ---- LDS -> MFMA fragment: lane (g = lane>>4, li = lane&15) gets
//      element j <-> (index = base + li, k = kk*32 + 8g + j)
template <typename E, bool KMAJOR>
__device__ __forceinline__ typename Elem16<E>::v8 lds_frag(const char* tile, int base, int kk, int g, int li) {
  typedef typename Elem16<E>::v8 V8;
  if (KMAJOR) {
    const int row = base + li;
    const int c = kk * 4 + g;
    return *reinterpret_cast<const V8*>(tile + row * (BK * 2) + ((c ^ (row & 7)) << 4));
  } else {
    const int q = li >> 2, p = li & 3;
    const int u = base >> 4;  // 32-byte unit of the 16-column block
    typename Elem16<E>::v4 half[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int k = kk * 32 + 8 * g + 4 * hf + q;
      const char* a = tile + k * (BM * 2) + ((u ^ swz_mn(k)) << 5) + 8 * p;
      half[hf] = Elem16<E>::tr_read(a);
    }
    // concatenation, not element inserts: the two 64-bit reads land in adjacent VGPR pairs
    return __builtin_shufflevector(half[0], half[1], 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

template <typename E, bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_mfma_kernel(const GemmParams p) {
  typedef typename Elem16<E>::v8 V8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int g = lane >> 4, li = lane & 15;

  const int tile = blockIdx.x;
  const int m0 = (tile / p.tiles_n) * BM;
  const int n0 = (tile % p.tiles_n) * BN;
  const int kbeg = blockIdx.z * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nk = (kend - kbeg + BK - 1) / BK;

  f32x4 acc[4][4];  // [u: n sub-tile][t: m sub-tile]
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 ra[4], rb[4];
  if (nk > 0) {
    g2r<A_KMAJOR>(p.A, p.lda, m0, p.M, kbeg, kend, tid, ra);
    g2r<B_KMAJOR>(p.B, p.ldb, n0, p.N, kbeg, kend, tid, rb);
    r2s<A_KMAJOR>(smem, tid, ra);
    r2s<B_KMAJOR>(smem + BM * BK * 2, tid, rb);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const char* sa = smem + (kt & 1) * kStageBytes;
    const char* sb = sa + BM * BK * 2;
    const bool more = kt + 1 < nk;
    if (more) {
      const int k0 = kbeg + (kt + 1) * BK;
      g2r<A_KMAJOR>(p.A, p.lda, m0, p.M, k0, kend, tid, ra);
      g2r<B_KMAJOR>(p.B, p.ldb, n0, p.N, k0, kend, tid, rb);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      V8 af[4], bfr[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) af[t] = lds_frag<E, A_KMAJOR>(sa, wm * 64 + t * 16, kk, g, li);
#pragma unroll
      for (int u = 0; u < 4; ++u) bfr[u] = lds_frag<E, B_KMAJOR>(sb, wn * 64 + u * 16, kk, g, li);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[u][t] = Elem16<E>::mma(bfr[u], af[t], acc[u][t]);
    }
    if (more) {
      char* da = smem + ((kt + 1) & 1) * kStageBytes;
      r2s<A_KMAJOR>(da, tid, ra);
      r2s<B_KMAJOR>(da + BM * BK * 2, tid, rb);
    }
    __syncthreads();
  }

  // ---- accumulators -> LDS (fp32, padded rows): lane holds C[m = .. + li][n = .. + 4g + r]
  float* cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int m = wm * 64 + t * 16 + li;
      const int n = wn * 64 + u * 16 + 4 * g;
      *reinterpret_cast<f32x4*>(cs + m * kCPad + n) = acc[u][t] * p.alpha;
    }
  __syncthreads();

  // ---- coalesced epilogue: 8 consecutive n per thread, 16 threads per row.  A real loop
  //      (not unrolled): the epilogue code exists once and stays I-cache resident.
  const int c = (tid & 15) << 3;
  const int n = n0 + c;
  float bias[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (p.bias && n < p.N && !p.slab) load8<float>(p.bias + n, bias);
#pragma unroll 1
  for (int i = 0; i < 8; ++i) {
    const int row = (tid >> 4) + 16 * i;
    const int m = m0 + row;
    if (m >= p.M || n >= p.N) continue;
    float v[8];
    {
      const f32x4 a = *reinterpret_cast<const f32x4*>(cs + row * kCPad + c);
      const f32x4 b = *reinterpret_cast<const f32x4*>(cs + row * kCPad + c + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] = a[k]; v[4 + k] = b[k]; }
    }
    if (p.slab) {  // raw split-K partial
      store8<float>(p.slab + ((int64_t)blockIdx.z * p.M + m) * p.N + n, v);
      continue;
    }
    float ld[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pre[8];
    if (p.epilogue == DVT_EPI_RESIDUAL) load8<E>((const E*)p.residual + (int64_t)m * p.ldr + n, ld);
    if (p.epilogue == DVT_EPI_DGELU || p.epilogue == DVT_EPI_DRELU)
      load8<E>((const E*)p.aux + (int64_t)m * p.ldaux + n, ld);
    epi_apply8(p.epilogue, v, bias, ld, pre);
    if (p.epilogue == DVT_EPI_GELU && p.aux) store8<E>((E*)p.aux + (int64_t)m * p.ldaux + n, pre);
    if (p.out_f32) {
      float* o = (float*)p.C + (int64_t)m * p.ldc + n;
      if (p.accumulate) {
        float old[8];
        load8<float>(o, old);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += old[k];
      }
      store8<float>(o, v);
    } else {
      store8<E>((E*)p.C + (int64_t)m * p.ldc + n, v);
    }
  }
}

constexpr int kConvSplitRows = 64;   // output rows per workgroup of conv_split_reduce_kernel = per BatchNorm partial row
// out[m][n] = sum_z slab[z][m][n] (+ residual[m][n]) rounded to E; optional BatchNorm partial sums of the stored values, one row
// {sum, sum of squares} per kConvSplitRows output rows.  block = 32 column groups of 8 x 8 row lanes.
template <typename E>
__global__ __launch_bounds__(256) void conv_split_reduce_kernel(const float* __restrict__ slab, int splits, int M, int N,
                                                                E* __restrict__ out, const E* __restrict__ residual,
                                                                float* __restrict__ bn_partial) {
  __shared__ float red[2][8][32][8];
  const int cgl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int n = (blockIdx.x * 32 + cgl) * 8;
  const int r0 = blockIdx.y * kConvSplitRows;
  const int64_t MN = (int64_t)M * N;
  float bs[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (n < N) {
    for (int r = r0 + rl; r < min(M, r0 + kConvSplitRows); r += 8) {
      const int64_t e = (int64_t)r * N + n;
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int z = 0; z < splits; z += 4) {              // four slices per round trip, summed in slice order
        float v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (z + u < splits) load8<float>(slab + (int64_t)(z + u) * MN + e, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (z + u < splits) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += v[u][k];
          }
      }
      if (residual) {
        float rv[8];
        load8<E>(residual + e, rv);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += rv[k];
      }
      if (bn_partial) {                            // of the values as stored (one statistics convention on every route)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float f = (float)(E)acc[k];
          bs[k] += f;
          bq[k] = fmaf(f, f, bq[k]);
        }
      }
      store8<E>(out + e, acc);
    }
  }
  if (!bn_partial) return;
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[0][rl][cgl][k] = bs[k]; red[1][rl][cgl][k] = bq[k]; }
  __syncthreads();
  if (rl < 2 && n < N) {                                 // row lane 0: sums, row lane 1: sums of squares; fixed order
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      t[k] = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t[k] += red[rl][q][cgl][k];
    }
    store8<float>(bn_partial + ((int64_t)blockIdx.y * 2 + rl) * N + n, t);
  }
}

// C (+)= sum_z slab[z]   (fixed order => reproducible)
template <typename OutT>
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, int splits, int M, int N,
                                     OutT* __restrict__ C, int64_t ldc, int accumulate,
                                     const float* __restrict__ cs_slab, float* __restrict__ cs_out,
                                     int cs_accumulate) {
  const int64_t nvec = (int64_t)M * N / 8;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t MN = (int64_t)M * N;
  if (cs_slab) {   // fused bias gradient: cs_out[m] (+)= sum_z cs_slab[z][m]
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += stride) {
      float t = 0.f;
      for (int z = 0; z < splits; ++z) t += cs_slab[(int64_t)z * M + m];
      cs_out[m] = cs_accumulate ? cs_out[m] + t : t;
    }
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    const int64_t e = i * 8;
    const int m = (int)(e / N), n = (int)(e % N);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // eight slices per round trip (a rolled loop waits for every slice before it requests the next: `splits` dependent
    // memory latencies per output vector, 12 us for 21 slices); the additions keep the slice order => same bits
    constexpr int ZB = 8;
    int z = 0;
    for (; z + ZB <= splits; z += ZB) {
      float v[ZB][8];
#pragma unroll
      for (int u = 0; u < ZB; ++u) load8<float>(slab + (int64_t)(z + u) * MN + e, v[u]);
#pragma unroll
      for (int u = 0; u < ZB; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += v[u][k];
    }
    for (; z < splits; ++z) {
      float v[8];
      load8<float>(slab + (int64_t)z * MN + e, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += v[k];
    }
    OutT* o = C + (int64_t)m * ldc + n;
    if (accumulate) {
      float old[8];
      load8<OutT>(o, old);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += old[k];
    }
    store8<OutT>(o, acc);
  }
}

// Split-K reduce with the fused epilogue (bias / activation / residual), bf16 output:
// lets small-M GEMMs (the 33-token temporal encoder, M = 264) spread their K loop over
// many workgroups instead of running one latency-bound loop per tile.
template <typename E>
__global__ void splitk_reduce_epi_kernel(const float* __restrict__ slab, int splits, const GemmParams p) {
  const int64_t nvec = (int64_t)p.M * p.N / 8;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t MN = (int64_t)p.M * p.N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    const int64_t e = i * 8;
    const int m = (int)(e / p.N), n = (int)(e % p.N);
    // epilogue operands first, then the slices four per round trip (a rolled loop costs one memory latency per slice)
    float bias[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ld[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pre[8];
    if (p.bias) load8<float>(p.bias + n, bias);
    if (p.epilogue == DVT_EPI_RESIDUAL) load8<E>((const E*)p.residual + (int64_t)m * p.ldr + n, ld);
    if (p.epilogue == DVT_EPI_DGELU || p.epilogue == DVT_EPI_DRELU)
      load8<E>((const E*)p.aux + (int64_t)m * p.ldaux + n, ld);
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int ZB = 4;
    int z = 0;
    for (; z + ZB <= splits; z += ZB) {
      float t[ZB][8];
#pragma unroll
      for (int u = 0; u < ZB; ++u) load8<float>(slab + (int64_t)(z + u) * MN + e, t[u]);
#pragma unroll
      for (int u = 0; u < ZB; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += t[u][k];
    }
    for (; z < splits; ++z) {
      float t[8];
      load8<float>(slab + (int64_t)z * MN + e, t);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += t[k];
    }
    // (the slabs already carry alpha)
    epi_apply8(p.epilogue, v, bias, ld, pre);
    if (p.epilogue == DVT_EPI_GELU && p.aux) store8<E>((E*)p.aux + (int64_t)m * p.ldaux + n, pre);
    if (p.out_f32) {
      float* o = (float*)p.C + (int64_t)m * p.ldc + n;
      if (p.accumulate) {
        float old[8];
        load8<float>(o, old);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += old[k];
      }
      store8<float>(o, v);
    } else {
      store8<E>((E*)p.C + (int64_t)m * p.ldc + n, v);
    }
  }
}

// ---------------------------------------------------------------- generic path
struct GenericParams {
  const void* A;
  const void* B;
  void* C;
  int M, N, K;
  int64_t sam, sak, sbk, sbn, ldc;
  int epilogue, out_f32, accumulate;
  const float* bias;
  const void* residual;
  int64_t ldr;
  void* aux;
  int64_t ldaux;
  float alpha;
};

__device__ __forceinline__ float acc_fma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double acc_fma(float a, float b, double c) { return fma((double)a, (double)b, c); }

template <typename T>
__global__ __launch_bounds__(256) void gemm_generic_kernel(const GenericParams p) {
  constexpr int TM = 64, TN = 64, TK = 16;
  __shared__ float As[TK][TM + 1];
  __shared__ float Bs[TK][TN + 1];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const T* A = (const T*)p.A;
  const T* B = (const T*)p.B;
  // fp32 operands (the parity mode) accumulate in double: the products are exact in double, so the result is the
  // correctly rounded dot product -- the mode's deviation from the reference then comes from storage rounding alone
  // (behind BatchNorm'd ReLU stacks the forward noise decides how many ReLU masks flip, tests/test_gpu_cnn.py)
  using Acc = typename std::conditional<std::is_same<T, float>::value, double, float>::type;
  Acc acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0;

  for (int k0 = 0; k0 < p.K; k0 += TK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;  // 1024 elements per tile
      {
        // A tile: pick the faster-varying index to follow the contiguous dimension
        int mm, kk;
        if (p.sak == 1) { kk = idx & 15; mm = idx >> 4; } else { mm = idx & 63; kk = idx >> 6; }
        const int gm = m0 + mm, gk = k0 + kk;
        As[kk][mm] = (gm < p.M && gk < p.K) ? to_f32<T>(A[(int64_t)gm * p.sam + (int64_t)gk * p.sak]) : 0.f;
      }
      {
        int nn, kk;
        if (p.sbk == 1) { kk = idx & 15; nn = idx >> 4; } else { nn = idx & 63; kk = idx >> 6; }
        const int gn = n0 + nn, gk = k0 + kk;
        Bs[kk][nn] = (gn < p.N && gk < p.K) ? to_f32<T>(B[(int64_t)gk * p.sbk + (int64_t)gn * p.sbn]) : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = acc_fma(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= p.N) continue;
      const float bias = p.bias ? p.bias[n] : 0.f;
      float res = 0.f, aux = 0.f, pre = 0.f;
      if (p.epilogue == DVT_EPI_RESIDUAL) res = to_f32<T>(((const T*)p.residual)[(int64_t)m * p.ldr + n]);
      if (p.epilogue == DVT_EPI_DGELU || p.epilogue == DVT_EPI_DRELU)
        aux = to_f32<T>(((const T*)p.aux)[(int64_t)m * p.ldaux + n]);
      float v = epi_apply(p.epilogue, (float)(acc[i][j] * (Acc)p.alpha), bias, res, aux, pre);
      if (p.epilogue == DVT_EPI_GELU && p.aux) ((T*)p.aux)[(int64_t)m * p.ldaux + n] = from_f32<T>(pre);
      if (p.out_f32) {
        float* o = (float*)p.C + (int64_t)m * p.ldc + n;
        *o = p.accumulate ? *o + v : v;
      } else {
        ((T*)p.C)[(int64_t)m * p.ldc + n] = from_f32<T>(v);
      }
    }
  }
}

// Tiny outputs (the 19-class heads: [8, 512] x [19, 512]^T and its two gradients): the 64x64-tile kernel above runs
// them as ONE workgroup looping over K (24 us for 78 kFLOP).  Here every output element has its own wave (long K: the
// lanes stride over k, then a wave reduction) or its own thread (short K).  fp32 accumulation in a fixed order.
template <typename T, bool WAVE>
__global__ __launch_bounds__(256) void gemm_tiny_kernel(const GenericParams p) {
  const T* A = (const T*)p.A;
  const T* B = (const T*)p.B;
  const int64_t total = (int64_t)p.M * p.N;
  const int64_t o = WAVE ? (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6) : (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= total) return;
  const int m = (int)(o / p.N), n = (int)(o % p.N);
  using Acc = typename std::conditional<std::is_same<T, float>::value, double, float>::type;
  Acc acc_ = 0;
  if (WAVE) {
    for (int k = threadIdx.x & 63; k < p.K; k += 64)
      acc_ = acc_fma(to_f32<T>(A[(int64_t)m * p.sam + (int64_t)k * p.sak]), to_f32<T>(B[(int64_t)k * p.sbk + (int64_t)n * p.sbn]), acc_);
  } else {
    for (int k = 0; k < p.K; ++k)
      acc_ = acc_fma(to_f32<T>(A[(int64_t)m * p.sam + (int64_t)k * p.sak]), to_f32<T>(B[(int64_t)k * p.sbk + (int64_t)n * p.sbn]), acc_);
  }
  float acc = (float)acc_;
  if (WAVE) {
    acc = wave_sum(acc);
    if (threadIdx.x & 63) return;
  }
  const float bias = p.bias ? p.bias[n] : 0.f;
  float res = 0.f, aux = 0.f, pre = 0.f;
  if (p.epilogue == DVT_EPI_RESIDUAL) res = to_f32<T>(((const T*)p.residual)[(int64_t)m * p.ldr + n]);
  if (p.epilogue == DVT_EPI_DGELU || p.epilogue == DVT_EPI_DRELU) aux = to_f32<T>(((const T*)p.aux)[(int64_t)m * p.ldaux + n]);
  const float v = epi_apply(p.epilogue, acc * p.alpha, bias, res, aux, pre);
  if (p.epilogue == DVT_EPI_GELU && p.aux) ((T*)p.aux)[(int64_t)m * p.ldaux + n] = from_f32<T>(pre);
  if (p.out_f32) {
    float* out = (float*)p.C + (int64_t)m * p.ldc + n;
    *out = p.accumulate ? *out + v : v;
  } else {
    ((T*)p.C)[(int64_t)m * p.ldc + n] = from_f32<T>(v);
  }
}

// ---------------------------------------------------------------- colsum (bias gradients)
// partial[b][n] = sum over this block's row range; then a second pass sums partials.
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, int64_t ldx,
                                                             int64_t M, int64_t N, int rows_per_block,
                                                             float* __restrict__ partial, float* __restrict__ out,
                                                             int accumulate) {
  // thread handles 8 columns; blockDim.x = 256 threads = 32 column-chunks x 8 row lanes
  __shared__ float red[8][32][8];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int64_t c = ((int64_t)blockIdx.x * 32 + cl) * 8;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = min(M, r0 + rows_per_block);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c < N) {
    for (int64_t r = r0 + rl; r < r1; r += 8) {
      float v[8];
      load8<T>(x + r * ldx + c, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += v[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[rl][cl][k] = acc[k];
  __syncthreads();
  if (rl == 0 && c < N) {
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      t[k] = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) t[k] += red[r][cl][k];
    }
    if (out) {                                     // single row range (few rows): final result, no second pass
      if (accumulate) {
        float o[8];
        load8<float>(out + c, o);
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] += o[k];
      }
      store8<float>(out + c, t);
    } else {
      store8<float>(partial + (int64_t)blockIdx.y * N + c, t);
    }
  }
}

// out[n] (+)= sum_p partial[p][n]; block = 32 columns x 8 part-lanes.
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int nparts,
                                                           int64_t N, float* __restrict__ out,
                                                           int accumulate) {
  __shared__ float red[8][33];
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int64_t n = (int64_t)blockIdx.x * 32 + cl;
  float a0 = 0.f, a1 = 0.f;
  if (n < N) {
    int p = pl;
    for (; p + 8 < nparts; p += 16) {
      a0 += partial[(int64_t)p * N + n];
      a1 += partial[(int64_t)(p + 8) * N + n];
    }
    for (; p < nparts; p += 8) a0 += partial[(int64_t)p * N + n];
  }
  red[pl][cl] = a0 + a1;
  __syncthreads();
  if (pl == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += red[i][cl];
    out[n] = accumulate ? out[n] + t : t;
  }
}

template <typename T>
__global__ void colsum_generic_kernel(const T* __restrict__ x, int64_t ldx, int64_t M, int64_t N,
                                      float* __restrict__ out, int accumulate) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float t = 0.f;
  for (int64_t m = 0; m < M; ++m) t += to_f32<T>(x[m * ldx + n]);
  out[n] = accumulate ? out[n] + t : t;
}

constexpr int kColsumParts = 128;

// ---------------------------------------------------------------- host side
bool mfma_eligible(const dvt_gemm_desc* d) {
  if (!dvt_is_16bit(d->in_dtype)) return false;
  if (!(d->out_dtype == d->in_dtype || d->out_dtype == DVT_F32)) return false;
  // 16-byte operand loads run along k for k-major operands and along m / n for mn-major ones: K only has to be a
  // multiple of 8 when some operand is k-major (the register-staged kernel zero-fills a ragged K tail).
  if (d->N % 8 || ((d->a_kmajor || d->b_kmajor) && d->K % 8)) return false;
  if (d->lda % 8 || d->ldb % 8 || d->ldc % 8) return false;
  if (!d->a_kmajor && d->M % 8) return false;
  if (!dvt_aligned16(d->A) || !dvt_aligned16(d->B) || !dvt_aligned16(d->C)) return false;
  if (d->bias && !dvt_aligned16(d->bias)) return false;
  if (d->residual && (!dvt_aligned16(d->residual) || d->ldr % 8)) return false;
  if (d->aux && (!dvt_aligned16(d->aux) || d->ldaux % 8)) return false;
  if (d->M > (1 << 30) || d->N > (1 << 30) || d->K > (1 << 30)) return false;
  return true;
}

// Launch-bound regime (the 33-token temporal encoder, the heads): gemm_small.hip's panel-streaming kernel, no split-K.
bool small_regime(const dvt_gemm_desc* d) {
  if (d->split_k != 0 || d->K <= 0) return false;
  if ((double)d->M * (double)d->N * (double)d->K > 2147483648.0) return false;
  if (dvt_cdiv(d->M, 32) * dvt_cdiv(d->N, 64) > 4096) return false;
  const int tm = dvt_gemm_small_tile(d->M, d->N, d->a_kmajor != 0, d->b_kmajor != 0);
  if (tm == 0) return false;
  // The panel kernel never splits K: a deep product on a handful of tiles (the weight gradient of a narrow
  // convolution, K = output pixels) belongs to the split-K path.
  // (K up to 4096 stays here for a single row tile: the data gradient of the frametransformer encoder's QKV projection -- 28
  //  rows, K = 2688 -- was a split-K launch plus its reduce, 12 us, instead of one 7 us panel launch)
  if (d->K > (dvt_cdiv(d->M, tm) == 1 ? 4096 : 2048) && dvt_cdiv(d->M, tm) * dvt_cdiv(d->N, 64) < 96) return false;
  return true;
}

struct GemmPlan {
  bool use256;
  int cfg;   // LDS-DMA kernel configuration (gemm256.hip)
  int split;
  int kps;   // K per split (multiple of 64)
};

// Tile size and split-K factor.  Split-K only for the plain epilogue (weight
// gradients: few output tiles, K = number of tokens).
GemmPlan plan_gemm(const dvt_gemm_desc* d) {
  GemmPlan pl;
  const bool plain = d->epilogue == DVT_EPI_NONE && !d->bias && d->alpha == 1.0f;
  const bool can_split = plain;   // LDS-DMA path: slabs are summed without an epilogue
  const int64_t cus = dvt_num_cus();
  // --- 256x256 LDS-DMA kernel
  if (d->K % 64 == 0) {
    const int64_t t256 = dvt_cdiv(d->M, 256) * dvt_cdiv(d->N, 256);
    int64_t s = 1;
    if (d->split_k > 0) s = d->split_k;
    else if (can_split && t256 < (cus * 3) / 4) {
      s = cus / t256;                           // floor: tiles x slices must fit one round of workgroups (12 tiles x 22
                                                // slices = 264 > 256 CUs ran a second, nearly empty round: 2x the time)
      const int64_t maxs = d->K / 512;
      if (s > maxs) s = maxs;
      if (s > 256) s = 256;   // very deep K (convolution weight gradients over N*H*W rows): up to one slice per CU
      if (s < 1) s = 1;
    }
    if (!can_split) s = 1;
    const int64_t kps = dvt_cdiv(dvt_cdiv(d->K, s), 64) * 64;
    s = dvt_cdiv(d->K, kps);
    if (t256 * s >= 96) {
      pl.use256 = true;
      pl.split = (int)s;
      pl.kps = (int)kps;
      // cfg 0 (8 waves of 128 x 64) except for the arithmetic-heavy GELU / GELU' epilogues: with 16 waves of 64 x 64
      // (cfg 3) the epilogue's vector work of one wave overlaps the store latency of three others (FF1 206 -> 195 us).
      // cfg 1 (2 workgroups / CU) measured slower or equal on every Linear shape.
      // cfg 5 (cfg 0's tile, wave rows in antiphase) elsewhere: 0..5 % faster than cfg 0 on every metric shape
      // (interleaved medians, tools/gemm_bench.hip; 4096^3: 1,316 -> 1,358 TF/s).
      const bool heavy_epi = d->epilogue == DVT_EPI_GELU || d->epilogue == DVT_EPI_DGELU;
      pl.cfg = heavy_epi && s == 1 ? 3 : 5;
      // 224-row tiles (cfg 8: 7/8 of the MFMAs, 15/16 of the operand bytes of a 256-row tile) when they need no more rounds of
      // workgroups than 256-row tiles do and the last 256-row round is far from full: N = 512 at 50,432 rows is 394 tiles =
      // 2 rounds for 1.54 rounds of work, 452 tiles of 224 rows are 2 shorter rounds (3..7 % per launch).
      if (pl.cfg == 5 && s == 1 && d->a_kmajor && (d->epilogue == DVT_EPI_NONE || d->epilogue == DVT_EPI_RESIDUAL) &&
          d->out_dtype != DVT_F32) {
        const int64_t r256 = dvt_cdiv(t256, cus), r224 = dvt_cdiv(dvt_cdiv(d->M, 224) * dvt_cdiv(d->N, 256), cus);
        if (r224 <= r256 && r256 * cus - t256 > cus / 4) pl.cfg = 8;
      }
      return pl;
    }
  }
  // --- 128x128 register-staged kernel
  // Any epilogue may be split here: splitk_reduce_epi_kernel applies it after the sum.
  int64_t s = 1;
  if (d->split_k > 1) s = d->split_k;
  else if (d->split_k == 0) {
    const int64_t tiles = dvt_cdiv(d->M, BM) * dvt_cdiv(d->N, BN);
    if (tiles < cus && d->K >= 256) {
      s = dvt_cdiv(cus, tiles);
      const int64_t maxs = d->K / 128;     // at least two 64-wide k-tiles per slice
      if (s > maxs) s = maxs;
      if (s > 128) s = 128;
      if (s < 1) s = 1;
    }
  }
  const int64_t kps = dvt_cdiv(dvt_cdiv(d->K, s), BK) * BK;
  pl.use256 = false;
  pl.cfg = 0;
  pl.split = (int)dvt_cdiv(d->K, kps);
  pl.kps = (int)kps;
  return pl;
}

int check_desc(const dvt_gemm_desc* d) {
  DVT_REQUIRE(d, "dvt_gemm: null descriptor");
  DVT_REQUIRE(d->A && d->B && d->C, "dvt_gemm: null operand pointer");
  DVT_REQUIRE(d->M >= 0 && d->N >= 0 && d->K >= 0, "dvt_gemm: negative dimension");
  DVT_REQUIRE(d->in_dtype == DVT_F32 || dvt_is_16bit(d->in_dtype), "dvt_gemm: in_dtype %d unsupported",
              d->in_dtype);
  DVT_REQUIRE(d->out_dtype == d->in_dtype || d->out_dtype == DVT_F32,
              "dvt_gemm: out_dtype must equal in_dtype or be f32");
  DVT_REQUIRE(d->epilogue >= DVT_EPI_NONE && d->epilogue <= DVT_EPI_DRELU, "dvt_gemm: bad epilogue %d",
              d->epilogue);
  DVT_REQUIRE(!(d->accumulate && d->out_dtype != DVT_F32), "dvt_gemm: accumulate needs f32 output");
  DVT_REQUIRE(d->epilogue != DVT_EPI_RESIDUAL || d->residual, "dvt_gemm: RESIDUAL epilogue without residual");
  DVT_REQUIRE(!(d->epilogue == DVT_EPI_DGELU || d->epilogue == DVT_EPI_DRELU) || d->aux,
              "dvt_gemm: DGELU/DRELU epilogue without aux");
  const int64_t min_lda = d->a_kmajor ? d->K : d->M, min_ldb = d->b_kmajor ? d->K : d->N;
  DVT_REQUIRE(d->lda >= min_lda && d->ldb >= min_ldb && d->ldc >= d->N, "dvt_gemm: leading dimension too small");
  return DVT_OK;
}

}  // namespace

extern "C" {

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// The ONE routing decision, shared by dvt_gemm_workspace_bytes and dvt_gemm, so that the size reported is the size used:
// SMALL = panel-streaming kernel (no workspace), MFMA = LDS-DMA / register-staged kernels under `pl`, GENERIC = fp32 FMA.
enum GemmRouteKind { ROUTE_SMALL, ROUTE_MFMA, ROUTE_GENERIC };
struct GemmRoute { GemmRouteKind kind; GemmPlan pl; };
static GemmRoute route_gemm(const dvt_gemm_desc* d) {
  GemmRoute r{ROUTE_GENERIC, GemmPlan{}};
  if (!(mfma_eligible(d) && d->K > 0)) return r;
  if (small_regime(d)) { r.kind = ROUTE_SMALL; return r; }
  r.kind = ROUTE_MFMA;
  r.pl = plan_gemm(d);
  return r;
}

// the pending reduce as a launch of its own, any form of it (plain, fused bias gradient, convolution scatter)
__global__ void splitk_reduce_pending_kernel(const dvt_splitk_pending q) {
  splitk_reduce_f32_part(q, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

// Many slabs of a small product (the per-workgroup partials of dvt_conv3x3_c64_wgrad: 256 x [576][64]): one thread per
// 8-element vector summing all of them is 4,608 threads on 18 CUs and 32 dependent batches of loads (41 us).  Here a block
// = 32 vectors x 8 slab groups: every thread sums splits / 8 slabs, the eight group sums of a vector meet in LDS and are
// added in group order (fixed order: reproducible); 144 blocks, 4 batches per thread.
__global__ __launch_bounds__(256) void splitk_reduce_wide_kernel(const dvt_splitk_pending q) {
  __shared__ float red[8][32][9];
  const int vl = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int64_t nvec = q.M * q.N / 8, MN = q.M * q.N;
  const int64_t i = (int64_t)blockIdx.x * 32 + vl;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (i < nvec) {
    const int per = (q.splits + 7) / 8;
    const int z1 = min(q.splits, (grp + 1) * per);
    int z = grp * per;
    constexpr int ZB = 8;
    for (; z + ZB <= z1; z += ZB) {
      float v[ZB][8];
#pragma unroll
      for (int u = 0; u < ZB; ++u) load8<float>(q.slab + (int64_t)(z + u) * MN + i * 8, v[u]);
#pragma unroll
      for (int u = 0; u < ZB; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += v[u][k];
    }
    for (; z < z1; ++z) {
      float v[8];
      load8<float>(q.slab + (int64_t)z * MN + i * 8, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += v[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[grp][vl][k] = acc[k];
  __syncthreads();
  if (grp != 0 || i >= nvec) return;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float t = red[0][vl][k];
#pragma unroll
    for (int gidx = 1; gidx < 8; ++gidx) t += red[gidx][vl][k];
    acc[k] = t;
  }
  const int64_t e = i * 8, m = e / q.N, n = e % q.N;
  if (q.conv_taps > 0) {                                   // (m = tap * Cin + ci, n = co) -> the parameter's C[co][ci][tap]
    const int64_t tap = m / q.conv_cin, ci = m - tap * q.conv_cin;
    const int64_t cin_l = q.conv_cin_l > 0 ? q.conv_cin_l : q.conv_cin, cout_l = q.conv_cout_l > 0 ? q.conv_cout_l : q.N;
    if (ci >= cin_l) return;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (n + k < cout_l) {
        float* o = q.C + ((n + k) * cin_l + ci) * q.conv_taps + tap;
        *o = q.accumulate ? *o + acc[k] : acc[k];
      }
    return;
  }
  float* o = q.C + m * q.ldc + n;
  if (q.accumulate) {
    float old[8];
    load8<float>(o, old);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += old[k];
  }
  store8<float>(o, acc);
}

// The convolution-scatter reduce of a LARGE weight gradient as a launch of its own (layers whose data gradient is itself
// split-K and carries nothing; R(2+1)D layers 3 - 4: [4608][1152] x 2 slices = 42 MB of slabs).  The element-order form
// above writes 8 floats per thread, each 41 KB from the next (parameter layout [co][ci][tap], product layout
// [tap * Cin + ci][co]): 56 us per launch at 0.75 TB/s.  Here a workgroup owns 32 input channels x all taps x 32 output
// channels: the slab rows are read along co (128-byte segments, slices summed in slice order = the same bits), transposed
// through LDS, and written as 32 x taps contiguous floats per output channel: six launches of 16 - 56 us -> 6 - 15 us each,
// frametransformer 15.58 -> 15.42 ms same box.  (Carried reduces keep the element-order form in the carrier's grid tail:
// performing them here instead measured 0 .. +0.1 ms, `gpurun_out/r5_ab_nocarry.log`.)
constexpr int kScCi = 32, kScCo = 32;
__global__ __launch_bounds__(256) void splitk_reduce_conv_tiled_kernel(const dvt_splitk_pending q) {
  extern __shared__ float sc_tile[];                 // [taps * 32][33]
  const int taps = q.conv_taps, cin = q.conv_cin;
  const int cin_l = q.conv_cin_l > 0 ? q.conv_cin_l : q.conv_cin, cout_l = q.conv_cout_l > 0 ? q.conv_cout_l : (int)q.N;
  const int ci0 = blockIdx.x * kScCi, co0 = blockIdx.y * kScCo;
  const int64_t MN = q.M * q.N;
  const int nrows = taps * kScCi;
  for (int r = threadIdx.x >> 3; r < nrows; r += 32) {                    // 8 lanes x 4 floats per row
    const int tap = r / kScCi, cil = r - tap * kScCi, c4 = (threadIdx.x & 7) * 4;
    const int64_t m = (int64_t)tap * cin + ci0 + cil;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (ci0 + cil < cin && co0 + c4 < q.N) {
      const float* src = q.slab + m * q.N + co0 + c4;
      for (int z = 0; z < q.splits; ++z) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (int64_t)z * MN);
        acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) sc_tile[r * (kScCo + 1) + c4 + k] = acc[k];
  }
  __syncthreads();
  // per output channel: (ci, tap) pairs of this tile are contiguous in the parameter
  const int nout = kScCi * taps;
  for (int idx = threadIdx.x; idx < kScCo * nout; idx += 256) {
    const int col = idx / nout, j = idx - col * nout;
    const int cil = j / taps, tap = j - cil * taps;
    const int co = co0 + col, ci = ci0 + cil;
    if (co < cout_l && ci < cin_l) {
      float* o = q.C + ((int64_t)co * cin_l + ci) * taps + tap;
      const float v = sc_tile[(tap * kScCi + cil) * (kScCo + 1) + col];
      *o = q.accumulate ? *o + v : v;
    }
  }
}

static int launch_pending_reduce(const dvt_splitk_pending* q, hipStream_t st) {
  if (!q || !q->valid) return DVT_OK;
  if (q->splits >= 64 && !q->cs_slab && q->M * q->N <= ((int64_t)1 << 20)) {
    hipLaunchKernelGGL(splitk_reduce_wide_kernel, dim3((unsigned)dvt_cdiv(q->M * q->N / 8, 32)), dim3(256), 0, st, *q);
    DVT_LAUNCH_CHECK("dvt_gemm(splitk reduce, many slabs)");
    return DVT_OK;
  }
  const int64_t nvec = q->M * q->N / 8;
  int64_t blocks = dvt_cdiv(nvec, 256);
  const int64_t cap = (int64_t)dvt_num_cus() * 8;
  if (blocks > cap) blocks = cap;
  if (q->conv_taps > 0 && !q->cs_slab && q->N % 4 == 0 && q->M * q->N >= ((int64_t)1 << 20) &&
      q->M == (int64_t)q->conv_taps * q->conv_cin && (size_t)q->conv_taps * kScCi * (kScCo + 1) * 4 <= 64 * 1024) {
    const dim3 grid((unsigned)dvt_cdiv(q->conv_cin, kScCi), (unsigned)dvt_cdiv(q->N, kScCo));
    hipLaunchKernelGGL(splitk_reduce_conv_tiled_kernel, grid, dim3(256), (size_t)q->conv_taps * kScCi * (kScCo + 1) * 4, st, *q);
    DVT_LAUNCH_CHECK("dvt_gemm(splitk reduce, convolution scatter, tiled)");
    return DVT_OK;
  }
  if (q->conv_taps > 0) {
    hipLaunchKernelGGL(splitk_reduce_pending_kernel, dim3((unsigned)blocks), dim3(256), 0, st, *q);
    DVT_LAUNCH_CHECK("dvt_gemm(splitk reduce, convolution scatter)");
    return DVT_OK;
  }
  hipLaunchKernelGGL((splitk_reduce_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, q->slab, q->splits, (int)q->M,
                     (int)q->N, q->C, q->ldc, q->accumulate, q->cs_slab, q->cs_out, q->cs_accumulate);
  DVT_LAUNCH_CHECK("dvt_gemm(splitk reduce)");
  return DVT_OK;
}

// workspace = [split-K slabs][bias-gradient slabs or stand-alone colsum scratch]
// The fused bias gradient writes one row of M floats per K slice (gemm256.hip: colsum_slab[slice * M + m]); the planner
// allows up to 256 slices, so the scratch is sized from the plan, never from a fixed slice bound.
int dvt_splitk_reduce_pending(const dvt_splitk_pending* pending, dvt_stream_t stream) {
  DVT_REQUIRE(pending, "dvt_splitk_reduce_pending: null descriptor");
  if (!pending->valid) return DVT_OK;
  DVT_REQUIRE(pending->slab && pending->C && pending->splits > 0 && pending->M > 0 && pending->N > 0 && pending->N % 8 == 0,
              "dvt_splitk_reduce_pending: bad descriptor");
  return launch_pending_reduce(pending, (hipStream_t)stream);
}

static GemmParams small_params(const dvt_gemm_desc* d) {
  GemmParams p{};
  p.A = (const bf16*)d->A; p.B = (const bf16*)d->B; p.C = d->C;
  p.M = (int)d->M; p.N = (int)d->N; p.K = (int)d->K;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.epilogue = d->epilogue; p.out_f32 = d->out_dtype == DVT_F32; p.accumulate = d->accumulate;
  p.bias = d->bias; p.residual = d->residual; p.ldr = d->ldr; p.aux = d->aux; p.ldaux = d->ldaux;
  p.alpha = d->alpha; p.elem = d->in_dtype; p.k_per_split = (int)d->K; p.slab = nullptr;
  p.colsum_slab = d->colsum_out; p.accumulate_colsum = d->colsum_accumulate;
  p.res_f32 = d->residual_f32 && d->epilogue == DVT_EPI_RESIDUAL;
  return p;
}

// The pair is one launch when both products are launch-bound shapes of the panel-streaming kernel in the layouts of a
// Linear's backward and neither takes part in a deferred split-K reduce.
static bool pair_fusable(const dvt_gemm_desc* w, const dvt_gemm_desc* g) {
  if (!w || !g || !w->A || !w->B || !w->C || !g->A || !g->B || !g->C) return false;
  if (w->a_kmajor || w->b_kmajor || !g->a_kmajor || g->b_kmajor) return false;
  if (w->in_dtype != g->in_dtype || !dvt_is_16bit(w->in_dtype)) return false;
  if (w->defer_reduce || g->defer_reduce || (w->carry && w->carry->valid) || (g->carry && g->carry->valid)) return false;
  if (w->M == 0 || w->N == 0 || g->M == 0 || g->N == 0) return false;
  return route_gemm(w).kind == ROUTE_SMALL && route_gemm(g).kind == ROUTE_SMALL;
}

// 0 = panel-streaming kernel (launch-bound shapes), 1 = LDS-DMA / register-staged MFMA kernels, 2 = generic fp32 kernel
int dvt_gemm_route(const dvt_gemm_desc* d) { return d ? (int)route_gemm(d).kind : 2; }

int dvt_gemm_pair_fused(const dvt_gemm_desc* wgrad, const dvt_gemm_desc* dgrad) { return pair_fusable(wgrad, dgrad) ? 1 : 0; }

int dvt_gemm_pair(const dvt_gemm_desc* wgrad, const dvt_gemm_desc* dgrad, dvt_stream_t stream) {
  int rc0 = check_desc(wgrad);
  if (rc0) return rc0;
  rc0 = check_desc(dgrad);
  if (rc0) return rc0;
  if (pair_fusable(wgrad, dgrad)) {
    const int rc = dvt_gemm_small_launch_pair(small_params(wgrad), small_params(dgrad), (hipStream_t)stream);
    if (rc != 1) return rc;
  }
  const int rc = dvt_gemm(wgrad, stream);
  return rc ? rc : dvt_gemm(dgrad, stream);
}

size_t dvt_gemm_workspace_bytes(const dvt_gemm_desc* d) {
  if (!d) return 0;
  const GemmRoute route = route_gemm(d);
  if (route.kind == ROUTE_SMALL) return 0;        // no slabs, the bias gradient comes out of the same launch
  const bool mfma = route.kind == ROUTE_MFMA;
  const GemmPlan pl = route.pl;
  size_t cs = 0;
  if (d->colsum_out) {
    const size_t fused = mfma ? (size_t)(pl.split > 1 ? pl.split : 1) * (size_t)d->M * sizeof(float) : 0;
    const size_t alone = dvt_colsum_workspace_bytes(d->K, d->M);
    cs = fused > alone ? fused : alone;
  }
  if (!mfma) return cs;
  const size_t slab = pl.split > 1 ? (size_t)pl.split * (size_t)d->M * (size_t)d->N * sizeof(float) : 0;
  return align256(slab) + cs;
}

int dvt_gemm(const dvt_gemm_desc* d, dvt_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (d->M == 0 || d->N == 0) return DVT_OK;
  hipStream_t st = (hipStream_t)stream;
  DVT_REQUIRE(!d->colsum_out || !d->a_kmajor, "dvt_gemm: colsum_out needs an mn-major A");

  const GemmRoute route = route_gemm(d);
  if (d->defer_reduce) {
    DVT_REQUIRE(d->pending, "dvt_gemm: defer_reduce needs a pending descriptor to fill");
    d->pending->valid = 0;
  }
  // a reduce carried over from an earlier call rides in this launch's grid tail when this is a one-slice LDS-DMA launch;
  // every other route performs it first, as a launch of its own
  const dvt_splitk_pending* carry = d->carry && d->carry->valid ? d->carry : nullptr;
  bool carried = false;
  if (carry && !(route.kind == ROUTE_MFMA && route.pl.use256 && route.pl.split == 1)) {
    rc = launch_pending_reduce(carry, st);
    if (rc) return rc;
    carry = nullptr;
  }
  if (d->residual_f32 && d->epilogue == DVT_EPI_RESIDUAL) {
    DVT_REQUIRE(d->out_dtype == DVT_F32 && d->ldr % 4 == 0 && (reinterpret_cast<uintptr_t>(d->residual) & 15u) == 0,
                "dvt_gemm: an fp32 residual needs an fp32 output, ldr %% 4 == 0 and a 16-byte aligned buffer");
    if (route.kind != ROUTE_SMALL)
      DVT_UNSUPPORTED("dvt_gemm: the fp32 residual epilogue is served for launch-bound shapes only (M = %lld)", (long long)d->M);
  }
  if (route.kind == ROUTE_SMALL) {
    const GemmParams p = small_params(d);
    rc = dvt_gemm_small_launch(p, d->a_kmajor != 0, d->b_kmajor != 0, st);
    if (rc == 1)        // cannot happen: small_regime() asks the same dvt_gemm_small_tile() the launcher dispatches on
      return dvt_fail(DVT_ERR_UNSUPPORTED, "dvt_gemm: the panel-streaming kernel has no instantiation for a shape its planner accepted");
    return rc;
  }
  DVT_REQUIRE(!d->colsum_out || d->workspace, "dvt_gemm: colsum_out needs a workspace (dvt_gemm_workspace_bytes)");
  if (route.kind == ROUTE_MFMA) {
    GemmPlan pl = route.pl;
    if (pl.split > 1 && !d->workspace) {   // no scratch: fall back to an unsplit 128x128 launch
      pl.use256 = false;
      pl.split = 1;
      pl.kps = (int)(dvt_cdiv(d->K, BK) * BK);
    }
    int split = pl.split;
    GemmParams p{};
    p.A = (const bf16*)d->A; p.B = (const bf16*)d->B; p.C = d->C;
    p.M = (int)d->M; p.N = (int)d->N; p.K = (int)d->K;
    p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
    p.epilogue = d->epilogue; p.out_f32 = d->out_dtype == DVT_F32; p.accumulate = d->accumulate;
    p.bias = d->bias; p.residual = d->residual; p.ldr = d->ldr; p.aux = d->aux; p.ldaux = d->ldaux;
    p.alpha = d->alpha;
    p.elem = d->in_dtype;
    p.k_per_split = pl.kps;
    p.slab = split > 1 ? (float*)d->workspace : nullptr;
    p.tiles_n = 0;
    // bias gradient fused into the LDS-DMA weight-gradient kernel (split-K slab path, cfg 0)
    const size_t slab_bytes = split > 1 ? align256((size_t)split * (size_t)d->M * (size_t)d->N * sizeof(float)) : 0;
    float* cs_scratch = d->colsum_out ? (float*)((char*)d->workspace + slab_bytes) : nullptr;
    const bool cs_fused = d->colsum_out && pl.use256 && (pl.cfg == 0 || pl.cfg == 5) && split > 1 && !d->a_kmajor && !d->b_kmajor &&
                          d->epilogue == DVT_EPI_NONE && p.out_f32;
    p.colsum_slab = cs_fused ? cs_scratch : nullptr;
    if (d->colsum_out && !cs_fused) {   // same semantics through the stand-alone reduction
      rc = dvt_colsum(d->A, d->lda, d->colsum_out, cs_scratch, d->K, d->M, d->in_dtype, d->colsum_accumulate, stream);
      if (rc) return rc;
    }
    if (pl.use256) {
      if (carry && split == 1) {
        // ~512 KB of slabs per tail workgroup, at most 128 of them: the first ~60 start in the CUs a 1.77-round data gradient
        // leaves idle in its last round, the rest as its last tiles finish
        const int64_t slab_bytes_c = carry->M * carry->N * 4 * carry->splits;
        int64_t nb = dvt_cdiv(slab_bytes_c, (int64_t)1 << 19);
        p.pig_blocks = (int)(nb < 1 ? 1 : nb > 128 ? 128 : nb);
        p.pig = *carry;
      }
      rc = dvt_gemm_dma_launch(p, d->a_kmajor != 0, d->b_kmajor != 0, split, pl.cfg, st);
      if (rc < 0) return rc;
      if (rc == 1) pl.use256 = false;   // no instantiation for this combination
      else carried = p.pig_blocks > 0;
      p.pig_blocks = 0;
    }
    if (carry && !carried) {              // the LDS-DMA kernel did not take the launch after all
      rc = launch_pending_reduce(carry, st);
      if (rc) return rc;
    }
    if (!pl.use256) {
    const int tiles_m = (int)dvt_cdiv(d->M, BM);
    p.tiles_n = (int)dvt_cdiv(d->N, BN);
    const dim3 grid((unsigned)(tiles_m * p.tiles_n), 1, (unsigned)split), block(NTHREADS);
    const bool ak = d->a_kmajor != 0, bk = d->b_kmajor != 0;
    DVT_DISPATCH_16BIT(d->in_dtype, E, {
      static DvtLdsAttr a_tt, a_tf, a_ft, a_ff;
      dvt_lds_attr(a_tt, (const void*)gemm_mfma_kernel<E, true, true>, kSmemBytes);
      dvt_lds_attr(a_tf, (const void*)gemm_mfma_kernel<E, true, false>, kSmemBytes);
      dvt_lds_attr(a_ft, (const void*)gemm_mfma_kernel<E, false, true>, kSmemBytes);
      dvt_lds_attr(a_ff, (const void*)gemm_mfma_kernel<E, false, false>, kSmemBytes);
      if (ak && bk) hipLaunchKernelGGL((gemm_mfma_kernel<E, true, true>), grid, block, kSmemBytes, st, p);
      else if (ak && !bk) hipLaunchKernelGGL((gemm_mfma_kernel<E, true, false>), grid, block, kSmemBytes, st, p);
      else if (!ak && bk) hipLaunchKernelGGL((gemm_mfma_kernel<E, false, true>), grid, block, kSmemBytes, st, p);
      else hipLaunchKernelGGL((gemm_mfma_kernel<E, false, false>), grid, block, kSmemBytes, st, p);
    });
    DVT_LAUNCH_CHECK("dvt_gemm(mfma)");
    }
    if (split > 1) {
      const int64_t nvec = d->M * d->N / 8;
      int64_t blocks = dvt_cdiv(nvec, 256);
      const int64_t cap = (int64_t)dvt_num_cus() * 8;
      if (blocks > cap) blocks = cap;
      const bool plain = d->epilogue == DVT_EPI_NONE && !d->bias && d->alpha == 1.0f;
      if (!plain)
        DVT_DISPATCH_16BIT(d->in_dtype, E, hipLaunchKernelGGL((splitk_reduce_epi_kernel<E>), dim3((unsigned)blocks),
                                                              dim3(256), 0, st, (const float*)p.slab, split, p));
      else if (p.out_f32 && d->defer_reduce) {          // left to the call that receives *pending as its `carry`
        dvt_splitk_pending* q = d->pending;
        q->slab = p.slab; q->splits = split; q->valid = 1; q->M = p.M; q->N = p.N; q->C = (float*)d->C; q->ldc = d->ldc;
        q->accumulate = d->accumulate; q->cs_accumulate = d->colsum_accumulate;
        q->cs_slab = p.colsum_slab; q->cs_out = d->colsum_out;
        q->conv_cin = 0; q->conv_taps = 0; q->conv_cin_l = 0; q->conv_cout_l = 0;
        return DVT_OK;
      }
      else if (p.out_f32)
        hipLaunchKernelGGL((splitk_reduce_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st,
                           (const float*)p.slab, split, p.M, p.N, (float*)d->C, d->ldc, d->accumulate,
                           (const float*)p.colsum_slab, d->colsum_out, d->colsum_accumulate);
      else
        DVT_DISPATCH_16BIT(d->in_dtype, E, hipLaunchKernelGGL((splitk_reduce_kernel<E>), dim3((unsigned)blocks), dim3(256),
                                                              0, st, (const float*)p.slab, split, p.M, p.N, (E*)d->C,
                                                              d->ldc, 0, (const float*)nullptr, (float*)nullptr, 0));
      DVT_LAUNCH_CHECK("dvt_gemm(splitk reduce)");
    }
    return DVT_OK;
  }

  if (d->colsum_out) {
    rc = dvt_colsum(d->A, d->lda, d->colsum_out, d->workspace, d->K, d->M, d->in_dtype, d->colsum_accumulate, stream);
    if (rc) return rc;
  }
  GenericParams g;
  g.A = d->A; g.B = d->B; g.C = d->C;
  g.M = (int)d->M; g.N = (int)d->N; g.K = (int)d->K;
  g.sam = d->a_kmajor ? d->lda : 1; g.sak = d->a_kmajor ? 1 : d->lda;
  g.sbk = d->b_kmajor ? 1 : d->ldb; g.sbn = d->b_kmajor ? d->ldb : 1;
  g.ldc = d->ldc; g.epilogue = d->epilogue; g.out_f32 = d->out_dtype == DVT_F32;
  g.accumulate = d->accumulate; g.bias = d->bias; g.residual = d->residual; g.ldr = d->ldr;
  g.aux = d->aux; g.ldaux = d->ldaux; g.alpha = d->alpha;
  const int64_t outs = d->M * d->N;
  if (outs <= 16384 && d->K >= 128) {            // one wave per output element
    DVT_DISPATCH_DTYPE(d->in_dtype, T, hipLaunchKernelGGL((gemm_tiny_kernel<T, true>), dim3((unsigned)dvt_cdiv(outs, 4)), dim3(256), 0, st, g));
    DVT_LAUNCH_CHECK("dvt_gemm(tiny)");
    return DVT_OK;
  }
  if (outs <= 65536 && d->K <= 32) {             // one thread per output element
    DVT_DISPATCH_DTYPE(d->in_dtype, T, hipLaunchKernelGGL((gemm_tiny_kernel<T, false>), dim3((unsigned)dvt_cdiv(outs, 256)), dim3(256), 0, st, g));
    DVT_LAUNCH_CHECK("dvt_gemm(tiny)");
    return DVT_OK;
  }
  const dim3 grid((unsigned)dvt_cdiv(d->N, 64), (unsigned)dvt_cdiv(d->M, 64)), block(256);
  DVT_DISPATCH_DTYPE(d->in_dtype, T, hipLaunchKernelGGL((gemm_generic_kernel<T>), grid, block, 0, st, g));
  DVT_LAUNCH_CHECK("dvt_gemm(generic)");
  return DVT_OK;
}

// ---------------------------------------------------------------- implicit-GEMM convolution
// LDS-DMA configuration of a convolution by its output width (gemm256.hip): 256x64x32 tiles up to 64 channels,
// 256x128x32 up to 128, 256x256x64 beyond.
// Up to 64 output channels: 64-deep k-tiles (cfg 6) when a k-tile can stay inside one filter tap (C % 64 == 0: a pixel's
// channels of the tap are then one whole 128-byte line per gather), else 32-deep (cfg 4: the stem's C = 8 form, C = 32).
// 65..128 output channels: 256x128x32 with two workgroups per CU for forward / data gradient; the weight gradient (deep K,
// no epilogue to overlap) is faster on the 64-deep form of that tile (cfg 7: layer 2 of ResNet-18 104 -> 93 us, forward
// 92 -> 97 us).
static int conv_cfg(int cout, int c, bool wgrad = false) {
  // (weight gradient of a width whose last 256-wide tile column is mostly padding -- 288 mid planes: 202 -> 183 us on the
  //  256 x 128 x 32 form; every other width keeps what the sweep tools/dev/conv_wgrad_cfg_sweep.py found it on already)
  if (wgrad && cout > 128 && ((cout + 255) / 256 * 256) * 4 > ((cout + 127) / 128 * 128) * 5) return 1;
  // (the weight gradient's k-tile runs over output pixels, not channels: its 64-deep forms take any C % 8 == 0)
  return cout <= 64 ? (c % 64 == 0 || wgrad ? 6 : 4) : cout <= 128 ? (wgrad ? 7 : 1) : 0;
}
static int conv_tk(int cfg) { return cfg == 0 || cfg >= 6 ? 64 : 32; }

// Forward / data-gradient launches with few output rows (layer 4 of a ResNet-18 on 224^2 frames: 49 pixels per frame)
// leave most CUs without a 256x256 tile: the 256x128 configuration doubles the tile count (and runs two per CU).
// Fewer still (R(2+1)D-18 layers 3 - 4 on 112^2 chunks: 16,464 and 2,744 output pixels): 128x128 tiles, two workgroups
// per CU -- at under a workgroup per CU every halving of the tile halves the launch.  And a 256x128 grid of one to two
// workgroups per CU runs as ONE round on the 72 KiB form (two per CU) instead of a full round plus a nearly empty one.
static int conv_fwd_cfg(int64_t rows, int cout, int c, int taps) {
  int cfg = conv_cfg(cout, c);
  // taps that split a k-tile (C % 32 != 0: every lane derives its own tap anyway): the 64-deep form whenever the padded K is
  // whole 64-deep k-tiles (144 channels x 3 temporal taps = 432 -> 448: 190 us on the 32-deep form, 155 us on this one)
  if (cfg == 4 && c != 8 && c % 32 != 0 && (((int64_t)taps * c + 31) / 32 * 32) % 64 == 0) cfg = 6;
  const int64_t cus = dvt_num_cus();
  if (cout > 64 && c % 64 == 0) {
    const int64_t t128 = dvt_cdiv(rows, 256) * dvt_cdiv(cout, 128);
    if (t128 * 4 < cus * 3) {
      // at most one 128x128 workgroup per CU: nothing shares the CU, so the deeper ring (three k-tiles in flight instead of
      // one) costs no occupancy and covers the operand latency a single in-flight k-tile leaves exposed (R(2+1)D layer 4:
      // 72 k-tiles of 0.2 us MFMA work each took 3 us)
      return dvt_cdiv(rows, 128) * dvt_cdiv(cout, 128) <= cus ? 10 : 9;
    }
  }
  {
    // per-shape sweep of every configuration on the layer 2 - 4 shapes of both encoders (tools/dev/conv_cfg_sweep.py with a
    // forced-configuration build; gpurun_out/r5_conv_cfg_sweep2.log; in the step: frametransformer -0.06 ms, pyramid -0.04 ms,
    // gpurun_out/r5_ab_tuned.log).  It also retired the round-3 rule "256 x 128 x 32 for one to two workgroups per CU of
    // 256-wide tiles" (ResNet-18 layer 3: 78 us against 68 on the 256 x 256 tile it replaced).
    //  * 65..128 output channels, K a whole number of 64-deep k-tiles: 128 x 128 x 64, two per CU, beats 256 x 128 x 32 on
    //    every such shape (R(2+1)D layer 2 data gradients 141 -> 116, 57 -> 51 us; ResNet-18 layer 2 92 -> 87)
    if (cfg == 1 && ((int64_t)taps * c) % 64 == 0 && c % 8 == 0) return 9;
    //  * widths whose last 256-wide tile column is mostly padding (288 mid planes): 256 x 128 x 32 (200 -> 178, 89 -> 71 us)
    if (cfg == 0 && c % 32 == 0) {
      const int w256 = (cout + 255) / 256 * 256, w128 = (cout + 127) / 128 * 128;
      if (w256 * 4 > w128 * 5) return 1;
    }
  }
  if (cfg == 0 && dvt_cdiv(rows, 256) * dvt_cdiv(cout, 256) * 4 < cus * 3) return c % 64 == 0 ? 7 : 1;
  return cfg;
}

// Split-K for forward / data-gradient launches that leave CUs empty AND run a deep reduction (R(2+1)D-18 layers 3 - 4 on
// 112^2 chunks: 2,744 - 4,116 output pixels x K up to 10,368 = 88 - 198 workgroups of 128 x 128, each a chain of up to 162
// k-tiles at ~1 us per k-tile -- the latency of one workgroup's DMA ring, not MFMA time).  The reduction is cut into S
// slices (blockIdx.z) so that about one and a half workgroups sit on every CU; fp32 slabs, summed in slice order by
// conv_split_reduce_kernel, which also rounds to the map's type, adds the shortcut's gradient and leaves the BatchNorm
// partial sums (of the stored values, like the one-slice epilogue).  1 = no split.
static int conv_fwd_split(int64_t rows, int cout, int c, int taps, int64_t K) {
  const int cfg = conv_fwd_cfg(rows, cout, c, taps);
  if ((cfg != 9 && cfg != 10) || K % 64) return 1;
  const int64_t tiles = dvt_cdiv(rows, 128) * dvt_cdiv(cout, 128), cus = dvt_num_cus();
  const int64_t nk = K / 64;
  if (tiles * 5 > cus * 4 || nk < 24) return 1;    // four fifths of the CUs busy already, or too shallow to be worth a reduce
  // shallow AND wide: the fp32 slabs (rows x cout x 4 bytes per slice, written and read back) outweigh the shorter chain
  // (R(2+1)D layer 4's temporal data gradients 512 -> 1152 / 960 at 24 k-tiles: 33 -> 28, 37 -> 28 us unsplit;
  // tools/dev/conv_split_sweep.py, gpurun_out/r5_split_sweep3.log)
  if (nk < 36 && cout > 512) return 1;
  // about one and a half workgroups per CU (same-box sweep on the frametransformer step, gpurun_out/r5_sweep_split2.log:
  // 17.57 ms unsplit; 17.38 - 17.43 at one per CU, 17.34 - 17.39 at 1.5, 17.44 - 17.45 at two)
  int64_t s = (3 * cus / 2 + tiles - 1) / tiles;
  if (s > nk / 6) s = nk / 6;                      // at least six k-tiles per slice
  if (s > 8) s = 8;
  return s < 2 ? 1 : (int)s;
}

// output size of the convolution a descriptor names (trim_w: columns dropped at the right edge)
static inline void conv_out_hw(const dvt_conv_desc* d, int64_t* Ho, int64_t* Wo) {
  *Ho = (d->H + 2 * d->ph - d->kh) / d->sh + 1;
  *Wo = (d->W + 2 * d->pw - d->kw) / d->sw + 1 - (d->trim_w > 0 ? d->trim_w : 0);
  if (d->out_h > 0 && d->out_w > 0) { *Ho = d->out_h; *Wo = d->out_w; }   // (a parity class of a strided data gradient)
}

static bool conv_implicit_ok(const dvt_conv_desc* d) {
  if (!d || !d->x || !d->w || !d->y) return false;
  if (!dvt_is_16bit(d->dtype)) return false;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->kh <= 0 || d->kw <= 0 || d->sh <= 0 || d->sw <= 0 || d->ph < 0 || d->pw < 0)
    return false;
  const int tk = d->Cout <= 128 ? 32 : 64;
  // C % k-tile == 0 (a k-tile inside one filter tap), or C == 8: the stem, its 3 channels zero-extended to one 16-byte
  // chunk per (pixel, tap) by dvt_nchw_to_nhwc_pad
  // (or any C % 8 == 0: the k-tile then straddles taps and every lane derives its own, ConvRows::dma)
  if (d->C % 8 || d->Cout % 8) return false;
  (void)tk;
  // the C == 8 gather derives a lane's tap by a 16-bit multiply-shift division (ConvRows::dma): exact for these bounds only
  if (d->C == 8 && !((int64_t)d->kh * d->kw < 1024 && d->kw < 64)) return false;
  int64_t Ho, Wo;
  conv_out_hw(d, &Ho, &Wo);
  if (Ho <= 0 || Wo <= 0) return false;
  if (d->N * Ho * Wo >= ((int64_t)1 << 31) || d->N * d->H * d->W >= ((int64_t)1 << 31)) return false;
  if ((int64_t)d->kh * d->kw * d->C >= ((int64_t)1 << 31)) return false;
  return dvt_aligned16(d->x) && dvt_aligned16(d->w) && dvt_aligned16(d->y);
}

int dvt_conv2d_implicit_supported(const dvt_conv_desc* d) { return conv_implicit_ok(d) ? 1 : 0; }

// slices of the reduction this descriptor's launch uses (1: the one-launch form); scattered class launches stay unsplit
static int conv_desc_split(const dvt_conv_desc* d) {
  if (!d || d->out_rows || d->C <= 0 || d->sh <= 0 || d->sw <= 0) return 1;
  int64_t Ho, Wo;
  conv_out_hw(d, &Ho, &Wo);
  if (Ho <= 0 || Wo <= 0 || d->N <= 0) return 1;
  return conv_fwd_split(d->N * Ho * Wo, d->Cout, d->C, d->kh * d->kw, dvt_conv2d_implicit_k(d));
}

// S fp32 slabs [rows][N] for a split launch (0: none needed)
size_t dvt_conv2d_implicit_workspace_bytes(const dvt_conv_desc* d) {
  const int s = conv_desc_split(d);
  if (s <= 1) return 0;
  int64_t Ho, Wo;
  conv_out_hw(d, &Ho, &Wo);
  return (size_t)s * (size_t)(d->N * Ho * Wo) * (size_t)d->Cout * sizeof(float);
}


// Row length of the packed weights [Cout][K]: kh*kw*C, rounded up to the k-tile in the C == 8 (stem) form.
int64_t dvt_conv2d_implicit_k(const dvt_conv_desc* d) {
  if (!d || d->C <= 0 || d->kh <= 0 || d->kw <= 0) return 0;
  const int64_t K = (int64_t)d->kh * d->kw * d->C;
  const int tk = d->Cout <= 128 ? 32 : 64;
  return d->C % tk ? dvt_cdiv(K, tk) * tk : K;
}

int dvt_conv2d_implicit(const dvt_conv_desc* d, dvt_stream_t stream) {
  DVT_REQUIRE(d, "dvt_conv2d_implicit: null descriptor");
  if (!conv_implicit_ok(d))
    DVT_UNSUPPORTED("dvt_conv2d_implicit: needs a 16-bit dtype, C %% 8 == 0, Cout %% 8 == 0 and "
                    "16-byte aligned buffers");
  int64_t Ho64, Wo64;
  conv_out_hw(d, &Ho64, &Wo64);
  const int Ho = (int)Ho64, Wo = (int)Wo64;
  GemmParams p{};
  p.A = (const bf16*)d->x; p.B = (const bf16*)d->w; p.C = d->y;
  p.M = (int)(d->N * Ho * Wo); p.N = d->Cout; p.K = (int)dvt_conv2d_implicit_k(d);
  p.lda = 0; p.ldb = p.K; p.ldc = d->Cout;
  p.epilogue = d->residual ? DVT_EPI_RESIDUAL : DVT_EPI_NONE; p.out_f32 = 0; p.accumulate = 0;
  p.bias = nullptr; p.residual = d->residual; p.ldr = d->Cout; p.aux = nullptr; p.ldaux = 0;
  p.alpha = 1.0f; p.elem = d->dtype; p.k_per_split = p.K; p.slab = nullptr; p.tiles_n = 0; p.colsum_slab = nullptr;
  p.cH = d->H; p.cW = d->W; p.cC = d->C; p.cHo = Ho; p.cWo = Wo; p.ckh = d->kh; p.ckw = d->kw;
  p.csh = d->sh; p.csw = d->sw; p.cph = d->ph; p.cpw = d->pw;
  p.cmc = (unsigned)((((uint64_t)1 << 32) + (uint64_t)(d->C / 8) - 1) / (uint64_t)(d->C / 8));   // (C == 8: unused)
  p.cmk = d->kw == 1 ? 0u : (unsigned)((((uint64_t)1 << 32) + (uint64_t)d->kw - 1) / (uint64_t)d->kw);
  DVT_REQUIRE(!(d->residual && d->stats_partial), "dvt_conv2d_implicit: residual and stats_partial are exclusive");
  DVT_REQUIRE(dvt_aligned16(d->residual), "dvt_conv2d_implicit: residual must be 16-byte aligned");
  p.bn_partial = d->stats_partial;
  DVT_REQUIRE((d->out_h > 0) == (d->out_w > 0) && (!d->out_rows || (((uintptr_t)d->out_rows & 3) == 0 && !d->stats_partial)) &&
                  (!d->residual_compact || (d->out_rows && d->residual)),
              "dvt_conv2d_implicit: out_h / out_w come together; out_rows is 4-byte aligned and excludes stats_partial; "
              "residual_compact needs out_rows and a residual");
  p.orow = d->out_rows; p.res_compact = d->residual_compact;
  const int split = conv_desc_split(d);
  if (split > 1) {
    DVT_REQUIRE(d->workspace && dvt_aligned16(d->workspace),
                "dvt_conv2d_implicit: this shape runs split-K and needs dvt_conv2d_implicit_workspace_bytes of 16-byte aligned workspace");
    hipStream_t st = (hipStream_t)stream;
    if (d->carry && d->carry->valid) {             // (the grid tail of a split launch carries nothing: the reduce runs by itself)
      const int rc = dvt_splitk_reduce_pending(d->carry, stream);
      if (rc != DVT_OK) return rc;
    }
    const int64_t nk = p.K / 64;
    p.k_per_split = (int)(dvt_cdiv(nk, split) * 64);
    p.slab = (float*)d->workspace;
    p.epilogue = DVT_EPI_NONE; p.residual = nullptr; p.bn_partial = nullptr;
    const int rc = dvt_conv_dma_launch_split(p, split, st);
    if (rc != DVT_OK) return rc;
    const dim3 grid((unsigned)dvt_cdiv(p.N, 256), (unsigned)dvt_cdiv(p.M, kConvSplitRows));
    DVT_DISPATCH_16BIT(d->dtype, E, hipLaunchKernelGGL((conv_split_reduce_kernel<E>), grid, dim3(256), 0, st, (const float*)p.slab,
                                                       split, p.M, p.N, (E*)d->y, (const E*)d->residual, d->stats_partial));
    DVT_LAUNCH_CHECK("dvt_conv2d_implicit(split-K reduce)");
    return DVT_OK;
  }
  if (d->carry && d->carry->valid) {               // a pending split-K reduce rides in this launch's grid tail
    const int64_t slab_bytes_c = d->carry->M * d->carry->N * 4 * d->carry->splits;
    int64_t nb = dvt_cdiv(slab_bytes_c, (int64_t)1 << 19);
    p.pig_blocks = (int)(nb < 1 ? 1 : nb > 128 ? 128 : nb);
    p.pig = *d->carry;
  }
  return dvt_conv_dma_launch(p, conv_fwd_cfg(p.M, d->Cout, d->C, d->kh * d->kw), (hipStream_t)stream);
}

// one partial row per wave row of a 256-row tile: 2 (128 output rows each) in configurations 0 and 1, 4 (64 rows) in 4, 6, 7
int64_t dvt_conv2d_implicit_stats_parts(const dvt_conv_desc* d) {
  if (!d || d->sh <= 0 || d->sw <= 0) return 0;
  int64_t Ho, Wo;
  conv_out_hw(d, &Ho, &Wo);
  if (Ho <= 0 || Wo <= 0 || d->N <= 0) return 0;
  if (conv_desc_split(d) > 1) return dvt_cdiv(d->N * Ho * Wo, kConvSplitRows);                      // split launch: the reduce's row blocks
  const int cfg = conv_fwd_cfg(d->N * Ho * Wo, d->Cout, d->C, d->kh * d->kw);
  if (cfg == 9 || cfg == 10) return dvt_cdiv(d->N * Ho * Wo, 128) * 2;                             // 128-row tiles of two wave rows
  return dvt_cdiv(d->N * Ho * Wo, 256) * (cfg == 4 || cfg == 6 || cfg == 7 ? 4 : 2);   // wave rows per 256-row tile
}

size_t dvt_conv2d_implicit_stats_bytes(const dvt_conv_desc* d) {
  // + 64 rows: dvt_bn_stats_from_partials folds more than 256 partial rows into 64 behind them before it finalises
  return (size_t)(dvt_conv2d_implicit_stats_parts(d) + 64) * 2 * (size_t)(d && d->Cout > 0 ? d->Cout : 0) * sizeof(float);
}

// ---- weight gradient: dWt[(ki,kj,c), co] = sum_rows col[row, (ki,kj,c)] * dz[row, co], col gathered on the fly
struct ConvWgradPlan { int cfg, tk, split, kps; int64_t rows; int Ho, Wo, K; size_t slab_bytes; };

static bool conv_wgrad_plan(const dvt_conv_desc* d, ConvWgradPlan* pl) {
  if (!d || !d->x || !d->w || !d->y || !dvt_is_16bit(d->dtype)) return false;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->kh <= 0 || d->kw <= 0 || d->sh <= 0 || d->sw <= 0 || d->ph < 0 || d->pw < 0)
    return false;
  if (d->C % 8 || d->Cout % 8) return false;
  int64_t Ho, Wo;
  conv_out_hw(d, &Ho, &Wo);
  if (Ho <= 0 || Wo <= 0) return false;
  const int64_t rows = d->N * Ho * Wo;
  pl->cfg = conv_cfg(d->Cout, d->C, true);
  pl->tk = conv_tk(pl->cfg);
  // (any pixel count: rows past the end of the last k-tile are read as zeros by both operand gathers)
  if (rows >= ((int64_t)1 << 31) - 64 || d->N * d->H * d->W >= ((int64_t)1 << 31)) return false;
  pl->K = d->kh * d->kw * d->C;
  pl->rows = rows; pl->Ho = (int)Ho; pl->Wo = (int)Wo;
  const int tn = pl->cfg == 4 || pl->cfg == 6 ? 64 : pl->cfg ? 128 : 256;
  const bool two_per_cu = pl->cfg == 1 || pl->cfg == 4 || pl->cfg == 6;
  const int64_t tiles = dvt_cdiv(pl->K, 256) * dvt_cdiv(d->Cout, tn);
  const int64_t target = (int64_t)dvt_num_cus() * (two_per_cu ? 2 : 1);
  int64_t split = target / tiles;
  if (split < 1) split = 1;
  const int64_t ktiles = dvt_cdiv(rows, pl->tk);
  if (split > ktiles / 4) split = ktiles / 4 > 0 ? ktiles / 4 : 1;          // at least 4 k-tiles per slice
  pl->kps = (int)(dvt_cdiv(ktiles, split) * pl->tk);
  pl->split = (int)dvt_cdiv(rows, pl->kps);
  pl->slab_bytes = (size_t)pl->split * (size_t)pl->K * (size_t)d->Cout * sizeof(float);
  return dvt_aligned16(d->x) && dvt_aligned16(d->w) && dvt_aligned16(d->y);
}

int dvt_conv2d_implicit_wgrad_supported(const dvt_conv_desc* d) {
  ConvWgradPlan pl;
  return conv_wgrad_plan(d, &pl) ? 1 : 0;
}

size_t dvt_conv2d_implicit_wgrad_workspace_bytes(const dvt_conv_desc* d) {
  ConvWgradPlan pl;
  return conv_wgrad_plan(d, &pl) ? align256(pl.slab_bytes) : 0;
}

// x = d->x (NHWC input of the layer), dz = d->w ([rows, Cout]), dWt = d->y (f32 [kh*kw*C, Cout], overwritten)
int dvt_conv2d_implicit_wgrad(const dvt_conv_desc* d, dvt_stream_t stream) {
  ConvWgradPlan pl;
  if (!conv_wgrad_plan(d, &pl))
    DVT_UNSUPPORTED("dvt_conv2d_implicit_wgrad: needs a 16-bit dtype, C %% 8 == 0 and Cout %% 8 == 0");
  DVT_REQUIRE(d->workspace, "dvt_conv2d_implicit_wgrad: workspace (dvt_conv2d_implicit_wgrad_workspace_bytes) required");
  hipStream_t st = (hipStream_t)stream;
  GemmParams p{};
  p.A = (const bf16*)d->x; p.B = (const bf16*)d->w; p.C = d->y;
  p.M = pl.K; p.N = d->Cout; p.K = (int)pl.rows;
  p.lda = 0; p.ldb = d->Cout; p.ldc = d->Cout;
  p.epilogue = DVT_EPI_NONE; p.out_f32 = 1; p.accumulate = 0;
  p.bias = nullptr; p.residual = nullptr; p.ldr = 0; p.aux = nullptr; p.ldaux = 0;
  p.alpha = 1.0f; p.elem = d->dtype; p.k_per_split = pl.kps; p.slab = (float*)d->workspace; p.tiles_n = 0;
  p.colsum_slab = nullptr;
  p.cH = d->H; p.cW = d->W; p.cC = d->C; p.cHo = pl.Ho; p.cWo = pl.Wo; p.ckh = d->kh; p.ckw = d->kw;
  p.csh = d->sh; p.csw = d->sw; p.cph = d->ph; p.cpw = d->pw;
  int rc = dvt_conv_wgrad_dma_launch(p, pl.split, pl.cfg, st);
  if (rc) return rc;
  if (d->defer_reduce) {                           // left to the data-gradient launch of the same layer (d->pending -> its carry)
    DVT_REQUIRE(d->pending, "dvt_conv2d_implicit_wgrad: defer_reduce needs a pending descriptor to fill");
    dvt_splitk_pending* q = d->pending;
    q->slab = p.slab; q->splits = pl.split; q->valid = 1; q->M = p.M; q->N = p.N; q->C = (float*)d->y; q->ldc = d->Cout;
    q->accumulate = 0; q->cs_accumulate = 0; q->cs_slab = nullptr; q->cs_out = nullptr;
    q->conv_cin = 0; q->conv_taps = 0; q->conv_cin_l = 0; q->conv_cout_l = 0;
    if (d->wgrad_master_layout) {
      q->conv_cin = d->C; q->conv_taps = d->kh * d->kw; q->accumulate = d->wgrad_accumulate;
      q->conv_cin_l = d->wgrad_cin_l; q->conv_cout_l = d->wgrad_cout_l;
    }
    return DVT_OK;
  }
  if (d->wgrad_master_layout) {
    dvt_splitk_pending q{};
    q.slab = p.slab; q.splits = pl.split; q.valid = 1; q.M = p.M; q.N = p.N; q.C = (float*)d->y; q.ldc = d->Cout;
    q.accumulate = d->wgrad_accumulate; q.conv_cin = d->C; q.conv_taps = d->kh * d->kw;
    q.conv_cin_l = d->wgrad_cin_l; q.conv_cout_l = d->wgrad_cout_l;
    return launch_pending_reduce(&q, st);
  }
  const int64_t nvec = (int64_t)p.M * p.N / 8;
  int64_t blocks = dvt_cdiv(nvec, 256);
  const int64_t cap = (int64_t)dvt_num_cus() * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL((splitk_reduce_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, (const float*)p.slab, pl.split,
                     p.M, p.N, (float*)d->y, (int64_t)d->Cout, 0, (const float*)nullptr, (float*)nullptr, 0);
  DVT_LAUNCH_CHECK("dvt_conv2d_implicit_wgrad(reduce)");
  return DVT_OK;
}

size_t dvt_colsum_workspace_bytes(int64_t M, int64_t N) {
  (void)M;
  return (size_t)kColsumParts * (size_t)(N > 0 ? N : 0) * sizeof(float);
}

int dvt_colsum(const void* x, int64_t ldx, float* out, void* workspace, int64_t M, int64_t N,
               int dtype, int accumulate, dvt_stream_t stream) {
  DVT_REQUIRE(x && out && M >= 0 && N > 0 && ldx >= N, "dvt_colsum: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const bool vec = workspace && N % 8 == 0 && ldx % 8 == 0 && dvt_aligned16(x) && dvt_aligned16(workspace) && M >= 64;
  if (vec) {
    int parts = (int)(M / 64 < kColsumParts ? M / 64 : kColsumParts);
    if (parts < 1 || M <= 1024) parts = 1;       // few rows (33-token sequences): one launch writes the result directly
    const int rpb = (int)dvt_cdiv(M, parts);
    parts = (int)dvt_cdiv(M, rpb);
    const dim3 grid((unsigned)dvt_cdiv(N, 256), (unsigned)parts);
    DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((colsum_partial_kernel<T>), grid, dim3(256), 0, st,
                                                    (const T*)x, ldx, M, N, rpb, (float*)workspace,
                                                    parts == 1 ? out : (float*)nullptr, accumulate));
    DVT_LAUNCH_CHECK("dvt_colsum(partial)");
    if (parts == 1) return DVT_OK;
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)dvt_cdiv(N, 32)), dim3(256), 0, st,
                       (const float*)workspace, parts, N, out, accumulate);
    DVT_LAUNCH_CHECK("dvt_colsum(final)");
    return DVT_OK;
  }
  DVT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((colsum_generic_kernel<T>), dim3((unsigned)dvt_cdiv(N, 256)),
                                                  dim3(256), 0, st, (const T*)x, ldx, M, N, out, accumulate));
  DVT_LAUNCH_CHECK("dvt_colsum(generic)");
  return DVT_OK;
}

}  // extern "C"
