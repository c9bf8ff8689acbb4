// Dev experiment (not shipped): A-STATIONARY GEMM for the K = 512 Linears with a large output (FF1 + GELU:
// [M, 512] x [2048, 512]^T, two outputs of 206 MB each at the metric shape).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/dev/gemm_astat.hip <csrc>/gemm256_pp.hip <csrc>/runtime.hip -o gemm_astat
//
// A wave keeps its 32 rows x K = 512 of A in registers (32 x 16-byte fragments = 128 VGPRs, loaded once, straight from
// global memory in MFMA operand layout).  The workgroup streams the weight matrix through LDS in chunks of CN = 32 output
// columns (32 x 512 x 2 B = 32 KiB, double-buffered, LDS-DMA).  Per chunk a wave issues 64 MFMAs (2 row blocks x 2
// column blocks x 16 k-steps) into 16 accumulator registers, applies the epilogue and stores 32 x 32 outputs (x 2 for
// GELU's saved pre-activation): the store stream is steady (4 KiB per wave per chunk, draining under the next chunk's
// MFMAs) instead of a 256 KiB burst per 256 x 256 tile, and there is ONE barrier per chunk (64 MFMAs), none per k-tile.
// Price: every wave reads the whole 32 KiB chunk from LDS -- 8 waves x 32 KiB per chunk against 8 x 64 MFMAs: LDS read
// time (128 B/clk) equals MFMA time (16 clk each) at 2 waves per SIMD, i.e. the two pipes must overlap perfectly for 100 %.
#include "../../data-efficient-video-transformers_amd/csrc/gemm256.hip"
#include <vector>
#include <algorithm>
#include <string.h>
#include <random>
#include <math.h>

namespace {

constexpr int AK = 512;                 // reduction length (compile-time: the A panel lives in registers)
constexpr int CN = 32;                  // output columns per chunk
constexpr int kChunkBytes = CN * AK * 2;

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void swap16(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else if (N == 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
  else if (N == 60) asm volatile("s_waitcnt vmcnt(60)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#pragma clang diagnostic pop

// EPI: DVT_EPI_NONE (bias), DVT_EPI_GELU (bias, GELU, pre-activation saved to aux)
template <typename E, int NW, int EPI>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void gemm_astat_kernel(const GemmParams p) {
  typedef typename Elem16<E>::v8 V8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int m0 = blockIdx.x * (NW * 32) + wid * 32;
  const E* __restrict__ A = (const E*)p.A;
  const E* __restrict__ W = (const E*)p.B;
  const int nchunks = p.N / CN;
  constexpr int kRowsPerWave = CN / NW;                    // chunk rows a wave DMAs (1 KiB each)
  constexpr int NST = EPI == DVT_EPI_GELU ? 4 : 2;         // stores per wave per chunk

  auto issue = [&](int ch) {
    char* buf = smem + (ch & 1) * kChunkBytes;
#pragma unroll
    for (int i = 0; i < kRowsPerWave; ++i) {
      const int r = wid * kRowsPerWave + i;
      const E* src = W + (int64_t)(ch * CN + r) * AK + ((lane ^ (r & 15)) << 3);
      dvt_dma16(src, buf + r * 1024);
    }
  };
  issue(0);
  float* sbias = reinterpret_cast<float*>(smem + 2 * kChunkBytes);     // the whole bias vector: read per chunk without a
  for (int i = tid; i < p.N; i += NW * 64) sbias[i] = p.bias[i];        // vector-memory operation in the counted stream
  // the A panel: 2 row blocks x 16 k-steps, lane (g, li) <-> row li, k = 32 kk + 8 g .. + 7
  V8 af[2][16];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
      af[rb][kk] = *reinterpret_cast<const V8*>(A + (int64_t)(m0 + rb * 16 + li) * p.lda + kk * 32 + g * 8);
  // column offset of this lane's 8 consecutive outputs inside a chunk after the row exchange (see the epilogue)
  const int ccol = (g & 1) * 16 + (g >> 1) * 8;
  wait_vmcnt<0>();
  __syncthreads();

  long long tl_wait = 0, tl_mfma = 0, tl_epi = 0, tl_t0 = 0;
  const bool tl = p.slab != nullptr;
  const long long tl_start = tl ? __builtin_amdgcn_s_memtime() : 0;
  for (int ch = 0; ch < nchunks; ++ch) {
    // chunk ch has landed: it was requested one iteration ago and only this wave's NST stores of the previous chunk are
    // younger -- they keep draining under this chunk's MFMAs
    if (tl) tl_t0 = __builtin_amdgcn_s_memtime();
    if (ch) wait_vmcnt<NST>();
    __builtin_amdgcn_s_barrier();
    if (tl) { const long long t = __builtin_amdgcn_s_memtime(); tl_wait += t - tl_t0; tl_t0 = t; }
    if (ch + 1 < nchunks) issue(ch + 1);
    const char* buf = smem + (ch & 1) * kChunkBytes;
    f32x4 acc[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      V8 bf[2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int n = cb * 16 + li, c = kk * 4 + g;
        bf[cb] = *reinterpret_cast<const V8*>(buf + n * 1024 + ((c ^ (n & 15)) << 4));
      }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = Elem16<E>::mma(bf[cb], af[rb][kk], acc[rb][cb]);
    }
    // epilogue: lane (g, li) holds C[row li][4g .. 4g+3] of each 16 x 16 block.  Exchanging the odd lane rows of column
    // block 0 with the even ones of block 1 leaves every lane with 8 CONSECUTIVE columns of one row (16-byte stores,
    // each output row's 64 bytes written by four lanes of one instruction)
    if (tl) {
      const float probe = acc[1][1][3] * p.alpha;            // (waits for the last MFMA)
      asm volatile("" ::"v"(probe));
      const long long t = __builtin_amdgcn_s_memtime(); tl_mfma += t - tl_t0; tl_t0 = t;
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      float lo[4], hi[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {      // (the scaling is a compiler-visible VALU read of the MFMA result: the hazard
        lo[k] = acc[rb][0][k] * p.alpha;  //  recogniser does not see into the inline-asm exchange that follows)
        hi[k] = acc[rb][1][k] * p.alpha;
        swap16(lo[k], hi[k]);
      }
      float v[8], pre[8], bias[8];
      {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(sbias + ch * CN + ccol);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(sbias + ch * CN + ccol + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { bias[k] = b0[k]; bias[4 + k] = b1[k]; }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] = lo[k]; v[4 + k] = hi[k]; }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (EPI == DVT_EPI_GELU) { pre[k] = v[k] + bias[k]; v[k] = gelu_erf_f(pre[k]); }
        else v[k] += bias[k];
      }
      const int64_t m = m0 + rb * 16 + li;
      const int n = ch * CN + ccol;
      if (p.tiles_n != -7) {                    // (-7: the experiment "epilogue arithmetic without its stores")
        if (EPI == DVT_EPI_GELU) store8_nt<E>((E*)p.aux + m * p.ldaux + n, pre);
        store8_nt<E>((E*)p.C + m * p.ldc + n, v);
      } else {
        asm volatile("" ::"v"(v[0] + v[7] + pre[1]));
      }
    }
    if (tl) { const long long t = __builtin_amdgcn_s_memtime(); tl_epi += t - tl_t0; }
  }
  if (tl && lane == 0) {
    float* o = p.slab + ((int64_t)blockIdx.x * NW + wid) * 4;
    o[0] = (float)tl_wait; o[1] = (float)tl_mfma; o[2] = (float)tl_epi; o[3] = (float)(__builtin_amdgcn_s_memtime() - tl_start);
  }
}

// Variant 3: lockstep as variant 1, but a chunk's outputs are only CONVERTED in its epilogue; their stores are issued
// one at a time between the MFMAs of the NEXT chunk.  Measured on variant 1: the GELU arithmetic is ~500 cycles per wave and
// chunk, its four 1 KiB stores another ~1460 per pair of waves -- a CU's store path moves ~22 B/clk, and a wave stalls at
// issue while it is busy -- so the stores belong under the MFMA stream, not behind it.
template <typename E, int EPI>
__global__ __launch_bounds__(512, 1) void gemm_astat3_kernel(const GemmParams p) {
  typedef typename Elem16<E>::v8 V8;
  typedef int i4t __attribute__((ext_vector_type(4)));
  constexpr int NW = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int m0 = blockIdx.x * (NW * 32) + wid * 32;
  const E* __restrict__ A = (const E*)p.A;
  const E* __restrict__ W = (const E*)p.B;
  const int nchunks = p.N / CN;
  constexpr int kRowsPerWave = CN / NW;
  constexpr int NOUT = EPI == DVT_EPI_GELU ? 2 : 1;
  constexpr int NST = 2 * NOUT;

  auto issue = [&](int ch) {
    char* buf = smem + (ch & 1) * kChunkBytes;
#pragma unroll
    for (int i = 0; i < kRowsPerWave; ++i) {
      const int r = wid * kRowsPerWave + i;
      const E* src = W + (int64_t)(ch * CN + r) * AK + ((lane ^ (r & 15)) << 3);
      dvt_dma16(src, buf + r * 1024);
    }
  };
  issue(0);
  float* sbias = reinterpret_cast<float*>(smem + 2 * kChunkBytes);
  for (int i = tid; i < p.N; i += NW * 64) sbias[i] = p.bias[i];
  V8 af[2][16];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
      af[rb][kk] = *reinterpret_cast<const V8*>(A + (int64_t)(m0 + rb * 16 + li) * p.lda + kk * 32 + g * 8);
  const int ccol = (g & 1) * 16 + (g >> 1) * 8;
  wait_vmcnt<0>();
  __syncthreads();

  i4t pend[2][NOUT];                      // the previous chunk's converted outputs: [row block][h, u]
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int o = 0; o < NOUT; ++o) pend[rb][o] = i4t{0, 0, 0, 0};
  // row base pointers of this lane's two output rows
  E* const crow0 = (E*)p.C + (int64_t)(m0 + li) * p.ldc + ccol;
  E* const crow1 = (E*)p.C + (int64_t)(m0 + 16 + li) * p.ldc + ccol;
  E* const urow0 = EPI == DVT_EPI_GELU ? (E*)p.aux + (int64_t)(m0 + li) * p.ldaux + ccol : nullptr;
  E* const urow1 = EPI == DVT_EPI_GELU ? (E*)p.aux + (int64_t)(m0 + 16 + li) * p.ldaux + ccol : nullptr;
  auto store_pending = [&](int slot, int ch_prev) {       // slot 0 .. NST-1
    const int rb = slot / NOUT, o = slot % NOUT;
    E* dst = (o == 0 ? (rb ? crow1 : crow0) : (rb ? urow1 : urow0)) + ch_prev * CN;
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(pend[rb][o]) : "memory");
  };

  for (int ch = 0; ch < nchunks; ++ch) {
    if (ch) wait_vmcnt<0>();            // (the previous period's stores were issued BEFORE the DMA finished its period)
    __builtin_amdgcn_s_barrier();
    if (ch + 1 < nchunks) issue(ch + 1);
    const char* buf = smem + (ch & 1) * kChunkBytes;
    f32x4 acc[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      V8 bf[2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int n = cb * 16 + li, c = kk * 4 + g;
        bf[cb] = *reinterpret_cast<const V8*>(buf + n * 1024 + ((c ^ (n & 15)) << 4));
      }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = Elem16<E>::mma(bf[cb], af[rb][kk], acc[rb][cb]);
      // the previous chunk's stores, spread over this chunk's k-steps
      if (ch > 0 && (kk % (16 / NST)) == 1) store_pending(kk / (16 / NST), ch - 1);
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      float lo[4], hi[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        lo[k] = acc[rb][0][k] * p.alpha;
        hi[k] = acc[rb][1][k] * p.alpha;
        swap16(lo[k], hi[k]);
      }
      float v[8], pre[8], bias[8];
      {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(sbias + ch * CN + ccol);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(sbias + ch * CN + ccol + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { bias[k] = b0[k]; bias[4 + k] = b1[k]; }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] = lo[k]; v[4 + k] = hi[k]; }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (EPI == DVT_EPI_GELU) { pre[k] = v[k] + bias[k]; v[k] = gelu_erf_f(pre[k]); }
        else v[k] += bias[k];
      }
      V8 hv, uv;
#pragma unroll
      for (int k = 0; k < 8; ++k) { hv[k] = (E)v[k]; if (EPI == DVT_EPI_GELU) uv[k] = (E)pre[k]; }
      pend[rb][0] = __builtin_bit_cast(i4t, hv);
      if (EPI == DVT_EPI_GELU) pend[rb][NOUT - 1] = __builtin_bit_cast(i4t, uv);
    }
  }
#pragma unroll
  for (int slot = 0; slot < NST; ++slot) store_pending(slot, nchunks - 1);
}

template <typename E, int EPI>
void launch_astat3(const GemmParams& p, hipStream_t st) {
  const int kSmem = 2 * kChunkBytes + p.N * 4;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_astat3_kernel<E, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem);
    attr = true;
  }
  hipLaunchKernelGGL((gemm_astat3_kernel<E, EPI>), dim3((unsigned)(p.M / 256)), dim3(512), kSmem, st, p);
}

// Variant 4: SEVEN compute waves + ONE helper wave.  The compute waves never issue a vector-memory instruction inside the
// loop: they leave a chunk's converted outputs in an LDS staging buffer; the helper wave requests the weight chunks
// (LDS-DMA) and drains the staging buffer of the previous chunk to global memory -- it is the only wave that ever stalls
// on the CU's store path (~22 B/clk), and nothing waits behind it.  224 rows per workgroup: 226 workgroups, one round.
template <typename E, int EPI>
__global__ __launch_bounds__(512, 1) void gemm_astat4_kernel(const GemmParams p) {
  typedef typename Elem16<E>::v8 V8;
  typedef int i4t __attribute__((ext_vector_type(4)));
  constexpr int NWC = 7;
  constexpr int NB = 3;                                  // weight-chunk images in LDS: requested two periods ahead
  constexpr int NOUT = EPI == DVT_EPI_GELU ? 2 : 1;
  constexpr int kStageWave = 32 * NOUT * 64;              // bytes one wave stages per chunk
  constexpr int kStage = NWC * kStageWave;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int mwg = blockIdx.x * (NWC * 32);
  const E* __restrict__ W = (const E*)p.B;
  const int nchunks = p.N / CN;
  char* const stage = smem + NB * kChunkBytes;
  float* const sbias = reinterpret_cast<float*>(stage + 2 * kStage);
  for (int i = tid; i < p.N; i += 512) sbias[i] = p.bias[i];

  if (wid == NWC) {
    // ------------------------------------------------------------------ helper wave
    auto issue = [&](int ch) {
      char* buf = smem + (ch % NB) * kChunkBytes;
#pragma unroll 8
      for (int r = 0; r < CN; ++r) {
        const E* src = W + (int64_t)(ch * CN + r) * AK + ((lane ^ (r & 15)) << 3);
        dvt_dma16(src, buf + r * 1024);
      }
    };
    issue(0);
    if (nchunks > 1) issue(1);
    wait_vmcnt<0>();
    __syncthreads();                                       // (matches the compute waves' prologue barrier)
    const bool full = mwg + NWC * 32 <= p.M;               // (a ragged last workgroup skips stores: no fixed count)
    for (int ch = 0; ch <= nchunks; ++ch) {
      __builtin_amdgcn_s_barrier();       // period ch begins: chunk ch is in LDS, staging (ch - 1) is complete (nchunks + 1 in all)
      if (ch > 0) {
        const char* sg = stage + ((ch - 1) & 1) * kStage;
        const int row = lane >> 2, cc = (lane & 3) * 8;
#pragma unroll
        for (int w = 0; w < NWC; ++w)
#pragma unroll
          for (int o = 0; o < NOUT; ++o) {
            i4t d[2];
#pragma unroll
            for (int h = 0; h < 2; ++h)
              d[h] = *reinterpret_cast<const i4t*>(sg + w * kStageWave + o * 2048 + h * 1024 + lane * 16);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int64_t m = mwg + w * 32 + h * 16 + row;
              E* dst = (o == 0 ? (E*)p.C + m * p.ldc : (E*)p.aux + m * p.ldaux) + (ch - 1) * CN + cc;
              if (m < p.M && p.tiles_n != -7)
                asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(d[h]) : "memory");
            }
          }
      }
      if (ch + 2 < nchunks) issue(ch + 2);
      // chunk ch + 1 (requested a period ago) landed; this period's stores and the requests for chunk ch + 2 are younger
      // and stay in flight -- waiting for everything put a memory round trip per period on the critical path
      if (full && ch > 0 && ch + 2 < nchunks && p.tiles_n != -7) wait_vmcnt<NWC * NOUT * 2 + CN>(); else wait_vmcnt<0>();
    }
    return;
  }
  // -------------------------------------------------------------------- compute waves
  const E* __restrict__ A = (const E*)p.A;
  const int m0 = mwg + wid * 32;
  V8 af[2][16];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int64_t m = min((int64_t)(m0 + rb * 16 + li), (int64_t)p.M - 1);
      af[rb][kk] = *reinterpret_cast<const V8*>(A + m * p.lda + kk * 32 + g * 8);
    }
  const int ccol = (g & 1) * 16 + (g >> 1) * 8;
  __syncthreads();

  for (int ch = 0; ch < nchunks; ++ch) {
    __builtin_amdgcn_s_barrier();
    const char* buf = smem + (ch % NB) * kChunkBytes;
    f32x4 acc[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.tiles_n != -8)
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      V8 bf[2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int n = cb * 16 + li, c = kk * 4 + g;
        bf[cb] = *reinterpret_cast<const V8*>(buf + n * 1024 + ((c ^ (n & 15)) << 4));
      }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = Elem16<E>::mma(bf[cb], af[rb][kk], acc[rb][cb]);
    }
    char* sg = stage + (ch & 1) * kStage + wid * kStageWave;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      float lo[4], hi[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        lo[k] = acc[rb][0][k] * p.alpha;
        hi[k] = acc[rb][1][k] * p.alpha;
        swap16(lo[k], hi[k]);
      }
      float v[8], pre[8], bias[8];
      {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(sbias + ch * CN + ccol);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(sbias + ch * CN + ccol + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { bias[k] = b0[k]; bias[4 + k] = b1[k]; }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] = lo[k]; v[4 + k] = hi[k]; }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (EPI == DVT_EPI_GELU) { pre[k] = v[k] + bias[k]; v[k] = gelu_erf_f(pre[k]); }
        else v[k] += bias[k];
      }
      V8 hv, uv;
#pragma unroll
      for (int k = 0; k < 8; ++k) { hv[k] = (E)v[k]; if (EPI == DVT_EPI_GELU) uv[k] = (E)pre[k]; }
      // staging: [output][32 rows][64 bytes]; this lane's 16 bytes of row rb * 16 + li
      *reinterpret_cast<V8*>(sg + (rb * 16 + li) * 64 + ccol * 2) = hv;
      if (EPI == DVT_EPI_GELU) *reinterpret_cast<V8*>(sg + 2048 + (rb * 16 + li) * 64 + ccol * 2) = uv;
    }
  }
  __builtin_amdgcn_s_barrier();                            // releases the helper's last drain
}

template <typename E, int EPI>
void launch_astat4(const GemmParams& p, hipStream_t st) {
  const int nout = EPI == DVT_EPI_GELU ? 2 : 1;
  const int kSmem = 3 * kChunkBytes + 2 * 7 * 32 * nout * 64 + p.N * 4;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_astat4_kernel<E, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem);
    attr = true;
  }
  hipLaunchKernelGGL((gemm_astat4_kernel<E, EPI>), dim3((unsigned)((p.M + 223) / 224)), dim3(512), kSmem, st, p);
}

// Variant 2: the eight waves form two groups of four (one wave of each per SIMD) that run in ANTIPHASE: while one group
// multiplies chunk ch, the other runs the epilogue of its previous chunk (GELU is ~20 VALU slots per output, 1.3 k cycles per
// wave and chunk against 1.0 k cycles of MFMA issue -- in lockstep the two add up on every SIMD, in antiphase they overlap).
// Two barriers per chunk period; the chunk buffers still alternate (a chunk is live for exactly one period).
template <typename E, int EPI>
__global__ __launch_bounds__(512, 1) void gemm_astat2_kernel(const GemmParams p) {
  typedef typename Elem16<E>::v8 V8;
  constexpr int NW = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2;
  const int g = lane >> 4, li = lane & 15;
  const int m0 = blockIdx.x * (NW * 32) + wid * 32;
  const E* __restrict__ A = (const E*)p.A;
  const E* __restrict__ W = (const E*)p.B;
  const int nchunks = p.N / CN;
  constexpr int kRowsPerWave = CN / NW;
  constexpr int NST = EPI == DVT_EPI_GELU ? 4 : 2;

  auto issue = [&](int ch) {
    char* buf = smem + (ch & 1) * kChunkBytes;
#pragma unroll
    for (int i = 0; i < kRowsPerWave; ++i) {
      const int r = wid * kRowsPerWave + i;
      const E* src = W + (int64_t)(ch * CN + r) * AK + ((lane ^ (r & 15)) << 3);
      dvt_dma16(src, buf + r * 1024);
    }
  };
  issue(0);
  float* sbias = reinterpret_cast<float*>(smem + 2 * kChunkBytes);
  for (int i = tid; i < p.N; i += NW * 64) sbias[i] = p.bias[i];
  V8 af[2][16];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
      af[rb][kk] = *reinterpret_cast<const V8*>(A + (int64_t)(m0 + rb * 16 + li) * p.lda + kk * 32 + g * 8);
  const int ccol = (g & 1) * 16 + (g >> 1) * 8;
  wait_vmcnt<0>();
  __syncthreads();

  f32x4 acc[2][2];
  auto mfma_chunk = [&](int ch) {
    const char* buf = smem + (ch & 1) * kChunkBytes;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      V8 bf[2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int n = cb * 16 + li, c = kk * 4 + g;
        bf[cb] = *reinterpret_cast<const V8*>(buf + n * 1024 + ((c ^ (n & 15)) << 4));
      }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = Elem16<E>::mma(bf[cb], af[rb][kk], acc[rb][cb]);
    }
  };
  auto epilogue = [&](int ch) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      float lo[4], hi[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        lo[k] = acc[rb][0][k] * p.alpha;
        hi[k] = acc[rb][1][k] * p.alpha;
        swap16(lo[k], hi[k]);
      }
      float v[8], pre[8], bias[8];
      {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(sbias + ch * CN + ccol);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(sbias + ch * CN + ccol + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { bias[k] = b0[k]; bias[4 + k] = b1[k]; }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] = lo[k]; v[4 + k] = hi[k]; }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (EPI == DVT_EPI_GELU) { pre[k] = v[k] + bias[k]; v[k] = gelu_erf_f(pre[k]); }
        else v[k] += bias[k];
      }
      const int64_t m = m0 + rb * 16 + li;
      const int n = ch * CN + ccol;
      if (EPI == DVT_EPI_GELU) store8_nt<E>((E*)p.aux + m * p.ldaux + n, pre);
      store8_nt<E>((E*)p.C + m * p.ldc + n, v);
    }
  };

  long long tl_w1 = 0, tl_h1 = 0, tl_w2 = 0, tl_h2 = 0, tl_t0 = 0;
  const bool tl = p.slab != nullptr;
  auto stamp = [&](long long& accu) {
    if (tl) {
      const float probe = acc[1][1][3] * p.alpha;            // (an MFMA half: waits for its last MFMA)
      asm volatile("" ::"v"(probe));
      const long long t = __builtin_amdgcn_s_memtime(); accu += t - tl_t0; tl_t0 = t;
    }
  };
  if (tl) tl_t0 = __builtin_amdgcn_s_memtime();
  for (int ch = 0; ch < nchunks; ++ch) {
    // chunk ch has landed (requested one period ago; only this wave's NST stores of its last epilogue are younger)
    if (ch) wait_vmcnt<NST>();
    __builtin_amdgcn_s_barrier();
    stamp(tl_w1);
    if (ch + 1 < nchunks) issue(ch + 1);
    if (grp == 0) mfma_chunk(ch);
    else if (ch) epilogue(ch - 1);
    stamp(tl_h1);
    __builtin_amdgcn_s_barrier();
    stamp(tl_w2);
    if (grp == 0) epilogue(ch);
    else mfma_chunk(ch);
    stamp(tl_h2);
  }
  if (grp == 1) epilogue(nchunks - 1);
  if (tl && lane == 0) {
    float* o = p.slab + ((int64_t)blockIdx.x * NW + wid) * 4;
    o[0] = (float)tl_w1; o[1] = (float)tl_h1; o[2] = (float)tl_w2; o[3] = (float)tl_h2;
  }
}

template <typename E, int EPI>
void launch_astat2(const GemmParams& p, hipStream_t st) {
  const int kSmem = 2 * kChunkBytes + p.N * 4;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_astat2_kernel<E, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem);
    attr = true;
  }
  hipLaunchKernelGGL((gemm_astat2_kernel<E, EPI>), dim3((unsigned)(p.M / 256)), dim3(512), kSmem, st, p);
}

template <typename E, int NW, int EPI>
void launch_astat(const GemmParams& p, hipStream_t st) {
  const int kSmem = 2 * kChunkBytes + p.N * 4;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_astat_kernel<E, NW, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem);
    attr = true;
  }
  hipLaunchKernelGGL((gemm_astat_kernel<E, NW, EPI>), dim3((unsigned)(p.M / (NW * 32))), dim3(NW * 64), kSmem, st, p);
}

}  // namespace

// pure store bandwidth of the chip (is the two-output epilogue bound by the WRITE path?): every thread streams 16-byte
// stores, grid-stride; NT selects the non-temporal form
template <bool NT>
__global__ __launch_bounds__(256) void write_bw_kernel(bf16* dst, size_t n8, float val) {
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = val + k;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    if (NT) store8_nt<bf16>(dst + i * 8, v); else store8<bf16>(dst + i * 8, v);
  }
}
// the A-stationary epilogue's store pattern in isolation: a wave owns 32 rows of a [rows, 2048] bf16 matrix and walks the
// row in SEG-column segments (SEG = 32: every instruction writes 16 rows x 64 bytes -- half cache lines; SEG = 64: 8 rows x
// 128 bytes -- whole lines)
template <int SEG>
__global__ __launch_bounds__(512) void seg_write_kernel(bf16* dst, int rows, float val) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 512 + threadIdx.x) >> 6;
  const int m0 = wave * 32;
  if (m0 >= rows) return;
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = val + k;
  constexpr int LPR = SEG / 8, RPI = 64 / LPR;             // lanes per row, rows per instruction
  for (int c = 0; c < 2048; c += SEG)
#pragma unroll
    for (int r = 0; r < 32; r += RPI)
      store8_nt<bf16>(dst + (size_t)(m0 + r + lane / LPR) * 2048 + c + (lane % LPR) * 8, v);
}

__global__ __launch_bounds__(256) void read_bw_kernel(const bf16* src, size_t n8, float* out) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  f4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x)
    acc += *reinterpret_cast<const f4*>(src + i * 8);
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}

static void fill(bf16* d, size_t n, float scale, unsigned seed) {
  std::vector<unsigned short> h(n);
  std::mt19937 rng(seed);
  std::normal_distribution<float> dist(0.f, scale);
  for (size_t i = 0; i < n; ++i) {
    float f = dist(rng);
    unsigned u; memcpy(&u, &f, 4);
    h[i] = (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
  }
  hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 50432, N = 2048, K = AK;
  bf16 *A, *B, *C0, *C1, *U0, *U1; float* bias;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2);
  hipMalloc(&C0, (size_t)M * N * 2); hipMalloc(&C1, (size_t)M * N * 2);
  hipMalloc(&U0, (size_t)M * N * 2); hipMalloc(&U1, (size_t)M * N * 2);
  hipMalloc(&bias, N * 4);
  fill(A, (size_t)M * K, 1.0f, 1); fill(B, (size_t)N * K, 0.05f, 2);
  { std::vector<float> hb(N); for (int i = 0; i < N; ++i) hb[i] = 0.01f * (i % 17 - 8); hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice); }
  GemmParams p{};
  p.A = A; p.B = B; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N; p.ldaux = N;
  p.epilogue = DVT_EPI_GELU; p.bias = bias; p.alpha = 1.f; p.k_per_split = K; p.elem = DVT_BF16;
  // reference: the shipped kernel (configuration 3 serves the GELU epilogue)
  p.C = C0; p.aux = U0;
  dvt_gemm_dma_launch(p, true, true, 1, 3, 0);
  GemmParams q = p; q.C = C1; q.aux = U1;
  hipMemset(C1, 0xFF, (size_t)M * N * 2); hipMemset(U1, 0xFF, (size_t)M * N * 2);
  launch_astat<bf16, 8, DVT_EPI_GELU>(q, 0);
  hipDeviceSynchronize();
  {
    std::vector<unsigned short> a((size_t)M * N), b((size_t)M * N);
    hipMemcpy(a.data(), C0, a.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), C1, b.size() * 2, hipMemcpyDeviceToHost);
    size_t diff = 0; for (size_t i = 0; i < a.size(); ++i) diff += a[i] != b[i];
    hipMemcpy(a.data(), U0, a.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), U1, b.size() * 2, hipMemcpyDeviceToHost);
    size_t diffu = 0; for (size_t i = 0; i < a.size(); ++i) diffu += a[i] != b[i];
    printf("bitwise mismatches vs the shipped kernel: h %zu, u %zu of %zu\n", diff, diffu, a.size());
    // both against a double-precision reference on sampled elements (u = pre-activation)
    std::vector<unsigned short> ha((size_t)M * K), hb((size_t)N * K);
    hipMemcpy(ha.data(), A, ha.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), B, hb.size() * 2, hipMemcpyDeviceToHost);
    auto f = [](unsigned short h) { unsigned u = (unsigned)h << 16; float x; memcpy(&x, &u, 4); return (double)x; };
    double e0 = 0, e1 = 0, dmax = 0; int worst = -1;
    for (int t = 0; t < 4000; ++t) {
      const size_t m = ((size_t)t * 7919) % M, n = ((size_t)t * 104729) % N;
      double r = 0.01 * ((int)(n % 17) - 8);
      for (int k = 0; k < K; ++k) r += f(ha[m * K + k]) * f(hb[n * K + k]);
      const double x0 = f(a[m * N + n]), x1 = f(b[m * N + n]);
      e0 = std::max(e0, fabs(x0 - r)); e1 = std::max(e1, fabs(x1 - r));
      if (fabs(x0 - x1) > dmax) { dmax = fabs(x0 - x1); worst = t; }
      static int shown = 0;
      if (!(fabs(x1 - r) < 0.05) && shown < 24) { printf("  bad u at m %zu n %zu (n %% 32 = %zu, m %% 32 = %zu): %g vs %g\n", m, n, n % 32, m % 32, x1, r); ++shown; }
    }
    printf("sampled u: max |shipped - ref| %.4g, max |a-stationary - ref| %.4g, max |shipped - a-stationary| %.4g (sample %d)\n", e0, e1, dmax, worst);
  }
  {
    hipMemset(C1, 0xFF, (size_t)M * N * 2); hipMemset(U1, 0xFF, (size_t)M * N * 2);
    launch_astat2<bf16, DVT_EPI_GELU>(q, 0);
    hipDeviceSynchronize();
    std::vector<unsigned short> a((size_t)M * N), b((size_t)M * N);
    hipMemcpy(a.data(), C0, a.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), C1, b.size() * 2, hipMemcpyDeviceToHost);
    size_t diff = 0; for (size_t i = 0; i < a.size(); ++i) diff += a[i] != b[i];
    hipMemcpy(a.data(), U0, a.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), U1, b.size() * 2, hipMemcpyDeviceToHost);
    size_t diffu = 0; for (size_t i = 0; i < a.size(); ++i) diffu += a[i] != b[i];
    printf("antiphase variant, bitwise mismatches vs the shipped kernel: h %zu, u %zu\n", diff, diffu);
    hipMemset(C1, 0xFF, (size_t)M * N * 2); hipMemset(U1, 0xFF, (size_t)M * N * 2);
    launch_astat3<bf16, DVT_EPI_GELU>(q, 0);
    hipDeviceSynchronize();
    hipMemcpy(a.data(), C0, a.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), C1, b.size() * 2, hipMemcpyDeviceToHost);
    diff = 0; for (size_t i = 0; i < a.size(); ++i) diff += a[i] != b[i];
    hipMemcpy(a.data(), U0, a.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), U1, b.size() * 2, hipMemcpyDeviceToHost);
    diffu = 0; for (size_t i = 0; i < a.size(); ++i) diffu += a[i] != b[i];
    printf("deferred-store variant, bitwise mismatches vs the shipped kernel: h %zu, u %zu\n", diff, diffu);
    hipMemset(C1, 0xFF, (size_t)M * N * 2); hipMemset(U1, 0xFF, (size_t)M * N * 2);
    launch_astat4<bf16, DVT_EPI_GELU>(q, 0);
    hipDeviceSynchronize();
    hipMemcpy(a.data(), C0, a.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), C1, b.size() * 2, hipMemcpyDeviceToHost);
    diff = 0; for (size_t i = 0; i < a.size(); ++i) diff += a[i] != b[i];
    hipMemcpy(a.data(), U0, a.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), U1, b.size() * 2, hipMemcpyDeviceToHost);
    diffu = 0; for (size_t i = 0; i < a.size(); ++i) diffu += a[i] != b[i];
    printf("store-wave variant, bitwise mismatches vs the shipped kernel: h %zu, u %zu\n", diff, diffu);
  }
  for (int exp_ = 0; exp_ < 3; ++exp_) {   // timeline of the lockstep variant: s_memtime ticks per phase, summed over the 64 chunks, per wave
    float* tlb; hipMalloc(&tlb, (size_t)(M / 32) * 4 * 4);
    GemmParams qt = q; qt.slab = tlb;
    if (exp_ == 1) qt.tiles_n = -7;
    printf("%s: ", exp_ == 0 ? "GELU, two outputs" : exp_ == 1 ? "GELU arithmetic, NO stores" : "bias only, one output");
    if (exp_ == 2) launch_astat<bf16, 8, DVT_EPI_NONE>(qt, 0); else launch_astat<bf16, 8, DVT_EPI_GELU>(qt, 0);
    hipDeviceSynchronize();
    std::vector<float> h((size_t)(M / 32) * 4);
    hipMemcpy(h.data(), tlb, h.size() * 4, hipMemcpyDeviceToHost);
    double sw = 0, sm = 0, se = 0, st = 0; const int nwv = M / 32;
    for (int i = 0; i < nwv; ++i) { sw += h[i * 4]; sm += h[i * 4 + 1]; se += h[i * 4 + 2]; st += h[i * 4 + 3]; }
    printf("timeline (mean ticks per wave over 64 chunks): wait+barrier %.0f, mfma loop %.0f, epilogue issue %.0f, whole loop %.0f"
           "  [per chunk: %.0f / %.0f / %.0f]\n", sw / nwv, sm / nwv, se / nwv, st / nwv, sw / nwv / 64, sm / nwv / 64, se / nwv / 64);
  }
  {   // timeline of the antiphase variant, per group
    float* tlb; hipMalloc(&tlb, (size_t)(M / 32) * 4 * 4);
    GemmParams qt = q; qt.slab = tlb;
    launch_astat2<bf16, DVT_EPI_GELU>(qt, 0);
    hipDeviceSynchronize();
    std::vector<float> h((size_t)(M / 32) * 4);
    hipMemcpy(h.data(), tlb, h.size() * 4, hipMemcpyDeviceToHost);
    for (int grp = 0; grp < 2; ++grp) {
      double s4[4] = {0, 0, 0, 0}; int cnt = 0;
      for (int i = 0; i < M / 32; ++i) if (((i % 8) >> 2) == grp) { for (int k = 0; k < 4; ++k) s4[k] += h[i * 4 + k]; ++cnt; }
      printf("antiphase timeline, group %d (first half = %s), ticks per chunk: barrier A wait %.0f, first half %.0f, barrier B wait %.0f, second half %.0f\n",
             grp, grp == 0 ? "MFMA" : "epilogue", s4[0] / cnt / 64, s4[1] / cnt / 64, s4[2] / cnt / 64, s4[3] / cnt / 64);
    }
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  {
    const size_t n8 = (size_t)M * N / 8;                       // 206 MB per buffer
    for (int variant = 0; variant < 7; ++variant) {
      std::vector<double> tt;
      for (int round = 0; round < 5; ++round) {
        hipEventRecord(e0, 0);
        for (int it = 0; it < 5; ++it) {
          if (variant == 0) { hipLaunchKernelGGL(write_bw_kernel<false>, dim3(2048), dim3(256), 0, 0, C1, n8, 1.f);
                              hipLaunchKernelGGL(write_bw_kernel<false>, dim3(2048), dim3(256), 0, 0, U1, n8, 1.f); }
          else if (variant == 1) { hipLaunchKernelGGL(write_bw_kernel<true>, dim3(2048), dim3(256), 0, 0, C1, n8, 1.f);
                                   hipLaunchKernelGGL(write_bw_kernel<true>, dim3(2048), dim3(256), 0, 0, U1, n8, 1.f); }
          else if (variant == 2) { hipLaunchKernelGGL(write_bw_kernel<true>, dim3(256 * 8), dim3(256), 0, 0, C1, n8, 1.f);
                                   hipLaunchKernelGGL(write_bw_kernel<true>, dim3(256 * 8), dim3(256), 0, 0, U1, n8, 1.f); }
          else if (variant == 3) { hipLaunchKernelGGL(read_bw_kernel, dim3(2048), dim3(256), 0, 0, C1, n8, (float*)bias);
                                   hipLaunchKernelGGL(read_bw_kernel, dim3(2048), dim3(256), 0, 0, U1, n8, (float*)bias); }
          else if (variant == 4) { hipMemsetAsync(C1, 0, n8 * 16, 0); hipMemsetAsync(U1, 0, n8 * 16, 0); }
          else if (variant == 5) { hipLaunchKernelGGL(seg_write_kernel<32>, dim3((M / 32 + 7) / 8), dim3(512), 0, 0, C1, M, 1.f);
                                   hipLaunchKernelGGL(seg_write_kernel<32>, dim3((M / 32 + 7) / 8), dim3(512), 0, 0, U1, M, 1.f); }
          else { hipLaunchKernelGGL(seg_write_kernel<64>, dim3((M / 32 + 7) / 8), dim3(512), 0, 0, C1, M, 1.f);
                 hipLaunchKernelGGL(seg_write_kernel<64>, dim3((M / 32 + 7) / 8), dim3(512), 0, 0, U1, M, 1.f); }
        }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        tt.push_back(ms * 1e3 / 5);
      }
      std::sort(tt.begin(), tt.end());
      const char* nm[7] = {"plain 16-byte stores", "nt stores", "nt stores (same grid)", "16-byte loads", "hipMemsetAsync",
                           "row-owner waves, 64-byte segments", "row-owner waves, 128-byte segments"};
      printf("%-36s 2 x 206 MB: %7.1f us  %6.0f GB/s\n", nm[variant], tt[2], 2.0 * n8 * 16 / (tt[2] * 1e-6) / 1e9);
    }
  }
  constexpr int NV = 11;
  std::vector<double> t[NV];
  const char* names[NV] = {"shipped cfg3 (256x256, 16 waves)", "shipped cfg5 (antiphase)", "A-stationary, 8 waves / WG", "A-stationary, 4 waves / WG", "A-stationary, two groups in antiphase", "A-stationary, stores under the next chunk", "A-stationary, 7 compute waves + 1 store wave", "shipped cfg1 (256x128x32, 2 WG/CU)", "shipped cfg0 (256x256, 8 waves)", "store-wave form WITHOUT its stores", "store-wave form WITHOUT its MFMAs"};
  // every variant writes its own pair of output buffers (back-to-back launches into the SAME 412 MB ran up to 40 % slower
  // than into alternating ones: the previous launch's lines are still on their way out of the cache hierarchy)
  GemmParams pv[NV];
  for (int c = 0; c < NV; ++c) {
    pv[c] = p;
    bf16 *cc, *uu;
    hipMalloc(&cc, (size_t)M * N * 2); hipMalloc(&uu, (size_t)M * N * 2);
    pv[c].C = cc; pv[c].aux = uu;
  }
  auto run = [&](int c) {
    if (c == 0) dvt_gemm_dma_launch(pv[c], true, true, 1, 3, 0);
    else if (c == 1) dvt_gemm_dma_launch(pv[c], true, true, 1, 5, 0);
    else if (c == 2) launch_astat<bf16, 8, DVT_EPI_GELU>(pv[c], 0);
    else if (c == 3) launch_astat<bf16, 4, DVT_EPI_GELU>(pv[c], 0);
    else if (c == 4) launch_astat2<bf16, DVT_EPI_GELU>(pv[c], 0);
    else if (c == 5) launch_astat3<bf16, DVT_EPI_GELU>(pv[c], 0);
    else if (c == 6) launch_astat4<bf16, DVT_EPI_GELU>(pv[c], 0);
    else if (c == 7) dvt_gemm_dma_launch(pv[c], true, true, 1, 1, 0);
    else if (c == 8) dvt_gemm_dma_launch(pv[c], true, true, 1, 0, 0);
    else { GemmParams t = pv[c]; t.tiles_n = c == 9 ? -7 : -8; launch_astat4<bf16, DVT_EPI_GELU>(t, 0); }
  };
  for (int c = 0; c < NV; ++c) for (int it = 0; it < 3; ++it) run(c);
  hipDeviceSynchronize();
  for (int round = 0; round < 7; ++round)
    for (int c = 0; c < NV; ++c) {
      const int reps = 10;
      hipEventRecord(e0, 0);
      for (int it = 0; it < reps; ++it) run(c);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      t[c].push_back(ms * 1e3 / reps);
    }
  for (int c = 0; c < NV; ++c) {
    std::sort(t[c].begin(), t[c].end());
    const double us = t[c][t[c].size() / 2];
    printf("%-36s %7.1f us %7.1f TF/s  %6.0f GB/s (A + W + 2 outputs)\n", names[c], us, 2.0 * M * N * K / (us * 1e-6) / 1e12,
           ((double)M * K * 2 + (double)N * K * 2 + 2.0 * M * N * 2) / (us * 1e-6) / 1e9);
  }
  return 0;
}
