// gemm_common.h -- parameter block and epilogue shared by the GEMM kernels.
#pragma once
#include "common.h"

struct GemmParams {
  const bf16* A;
  const bf16* B;
  void* C;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int epilogue, out_f32, accumulate;
  const float* bias;
  const void* residual;
  int64_t ldr;
  int res_f32;          // gemm_small.hip only: the residual operand is fp32 (fp32 residual stream of the launch-bound zone)
  void* aux;
  int64_t ldaux;
  float alpha;
  int k_per_split;  // multiple of BK; == K rounded up when not splitting
  float* slab;      // != nullptr: write raw fp32 partials to slab[z][M][N]
  int tiles_n;
  int stream_out;       // C (and the saved pre-activation) written with streaming stores: outputs too large to stay in the Infinity Cache
  int elem;             // DVT_BF16 or DVT_F16: element type of A, B (and of C / residual / aux when not f32)
  float* colsum_slab;
  int accumulate_colsum;   // gemm_small.hip only: colsum_slab is the final bias-gradient vector; += when set
  float* bn_partial;    // != nullptr (bf16 output, no epilogue): partial[(tile_m * 2 + wave_m)][{sum, sum of squares}][N] of C's columns
  // implicit convolution, bf16 output: row m of the product is row orow[m] of C (and of the residual unless res_compact):
  // the parity classes of a strided convolution's data gradient scatter into the full-size gradient (dvt_conv_desc.out_rows)
  const int* orow;
  int res_compact;
  // a split-K reduce of an EARLIER launch carried in `pig_blocks` extra workgroups at the end of this grid (gemm256.hip)
  int pig_blocks;
  dvt_splitk_pending pig;
  // implicit-GEMM convolution (A operand gathered from an NHWC map instead of read from a column matrix)
  unsigned cmc, cmk;   // ceil(2^32 / (cC / 8)), ceil(2^32 / ckw) (0 for ckw == 1): the per-lane tap of ConvRows' general form
  int cH, cW, cC, cHo, cWo, ckh, ckw, csh, csw, cph, cpw;   // != nullptr (mn-major A, slab output): partial sum_k A(m,k) per K slice, [splits][M]
};

__device__ __forceinline__ int swz_mn(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }
// 32-byte-unit swizzle of an mn-major image row by its k index.  Rows of >= 256 bytes have eight units to permute
// (swz_mn); a 64-wide image (128-byte rows: four units; rows k and k+2 alias on the 256-byte bank row) uses two bits
// chosen so that the eight k rows a 32-lane half of ds_read_b64_tr_b16 touches ({q, 8+q} + 4 hf) hit eight bank groups.
template <int ROWS> __device__ __forceinline__ int swz_mn_r(int k) {
  return ROWS >= 128 ? swz_mn(k) : (((k >> 1) & 1) | (((k >> 3) & 1) << 1));
}

__device__ __forceinline__ float epi_apply(int epi, float acc, float bias, float res, float aux_in,
                                           float& aux_out) {
  switch (epi) {
    case DVT_EPI_GELU: return gelu_erf_both_f(acc + bias, aux_out);   // aux = gelu'(acc + bias)
    case DVT_EPI_RELU: return fmaxf(acc + bias, 0.f);
    case DVT_EPI_RESIDUAL: return acc + bias + res;
    case DVT_EPI_DGELU: return acc * aux_in;                           // (aux = the derivative the forward stored)
    case DVT_EPI_DRELU: return aux_in > 0.f ? acc : 0.f;
    default: return acc + bias;
  }
}


// C (+)= sum_z slab[z] in slice order for the vectors [rb * nthreads + tid, ...) strided by nrb * nthreads: the body of
// splitk_reduce_kernel<float>, callable from the tail workgroups of a GEMM launch (same order of additions => same bits).
__device__ __forceinline__ void splitk_reduce_f32_part(const dvt_splitk_pending& q, int64_t first, int64_t stride) {
  const int64_t nvec = q.M * q.N / 8, MN = q.M * q.N;
  if (q.cs_slab) {
    for (int64_t m = first; m < q.M; m += stride) {
      float t = 0.f;
      for (int z = 0; z < q.splits; ++z) t += q.cs_slab[(int64_t)z * q.M + m];
      q.cs_out[m] = q.cs_accumulate ? q.cs_out[m] + t : t;
    }
  }
  for (int64_t i = first; i < nvec; i += stride) {
    const int64_t e = i * 8;
    const int64_t m = e / q.N, n = e % q.N;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int ZB = 8;
    int z = 0;
    for (; z + ZB <= q.splits; z += ZB) {
      float v[ZB][8];
#pragma unroll
      for (int u = 0; u < ZB; ++u) load8<float>(q.slab + (int64_t)(z + u) * MN + e, v[u]);
#pragma unroll
      for (int u = 0; u < ZB; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += v[u][k];
    }
    for (; z < q.splits; ++z) {
      float v[8];
      load8<float>(q.slab + (int64_t)z * MN + e, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += v[k];
    }
    if (q.conv_taps > 0) {
      // convolution weight gradient: (m = tap * Cin + ci, n = co) goes to the parameter's own layout C[co][ci][tap]
      const int64_t tap = m / q.conv_cin, ci = m - tap * q.conv_cin;
      const int64_t cin_l = q.conv_cin_l > 0 ? q.conv_cin_l : q.conv_cin, cout_l = q.conv_cout_l > 0 ? q.conv_cout_l : q.N;
      if (ci < cin_l) {                            // (padded input channels have no parameter entries)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (n + k < cout_l) {
            float* o = q.C + ((n + k) * cin_l + ci) * q.conv_taps + tap;
            *o = q.accumulate ? *o + acc[k] : acc[k];
          }
        }
      }
      continue;
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
}

// 8 consecutive outputs at once, the switch hoisted out of the element loop.
__device__ __forceinline__ void epi_apply8(int epi, float (&v)[8], const float (&bias)[8],
                                           const float (&ld)[8], float (&pre)[8]) {
  switch (epi) {
    case DVT_EPI_GELU:
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = gelu_erf_both_f(v[k] + bias[k], pre[k]);
      break;
    case DVT_EPI_RELU:
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k] + bias[k], 0.f);
      break;
    case DVT_EPI_RESIDUAL:
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = v[k] + bias[k] + ld[k];
      break;
    case DVT_EPI_DGELU:
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] *= ld[k];
      break;
    case DVT_EPI_DRELU:
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = ld[k] > 0.f ? v[k] : 0.f;
      break;
    default:
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += bias[k];
  }
}

// LDS-DMA kernels (gemm256.hip).  cfg 0: 256x256x64, 1 workgroup / CU;  cfg 1: 256x128x32,
// 2 workgroups / CU.  K (and each K split) must be a multiple of 64.
int dvt_conv_dma_launch(const GemmParams& p, int cfg, hipStream_t st);
int dvt_conv_wgrad_dma_launch(const GemmParams& p, int split, int cfg, hipStream_t st);
int dvt_gemm_dma_launch(const GemmParams& p, bool a_kmajor, bool b_kmajor, int split, int cfg, hipStream_t st);
int dvt_conv_dma_launch_c6(const GemmParams& p, int cfg, hipStream_t st);                                             // gemm256_pp.hip
int dvt_conv_dma_launch_split(const GemmParams& p, int split, hipStream_t st);                                         // gemm256_pp.hip
int dvt_conv_wgrad_dma_launch_c6(const GemmParams& p, int split, int cfg, hipStream_t st);                             // gemm256_pp.hip
int dvt_gemm_dma_launch_224(const GemmParams& p, bool b_kmajor, hipStream_t st);                                          // gemm256_pp.hip
int dvt_gemm_dma_launch_pp(const GemmParams& p, bool a_kmajor, bool b_kmajor, int split, hipStream_t st);   // gemm256_pp.hip
// launch-bound shapes (gemm_small.hip): tile height (0 = shape / layout not taken), launch (1 = no instantiation)
int dvt_gemm_small_tile(int64_t M, int64_t N, bool a_kmajor, bool b_kmajor);
int dvt_gemm_small_launch(const GemmParams& p, bool a_kmajor, bool b_kmajor, hipStream_t st);
int dvt_gemm_small_launch_pair(const GemmParams& pw, const GemmParams& pd, hipStream_t st);
