// gemm_small.hip -- MFMA GEMM for launch-bound shapes: a few hundred rows (the 33-token temporal encoder of
// vit.py:122-128 at B = 8 is 264 rows; the heads are 8), where the whole launch lives for a few microseconds.
//
// The 128x128 register-staged kernel (gemm.hip) runs such a GEMM as one latency-bound k loop per tile (a global
// load -> LDS write -> barrier round trip per 64 of K) and fills the chip by splitting K over workgroups, which costs a
// second launch for the slab reduce: ~10 + 5 us per Linear.  Here a workgroup owns a TM x 64 output tile (TM = 32 or 64)
// and streams its two operand panels through LDS in 256-deep k chunks by LDS-DMA (global_load_lds_dwordx4, no VGPR
// staging), two chunks in flight: for K = 512 both chunks are requested before the first MFMA and the second lands
// under the first one's MFMAs.  No split-K, no second launch; the fused epilogues of the family (bias, GELU + saved
// pre-activation, ReLU, residual, GELU', ReLU', fp32 accumulate) and the fused bias gradient of the weight-gradient
// form run on the accumulators.
//
// Layouts as in the rest of the family (no operand is transposed through HBM):
//   k-major  operand [rows][256 k]: 512-byte rows; 16-byte chunks XOR-swizzled by (row & 15) -> conflict-free ds_read_b128
//   mn-major operand [256 k][64 mn]: 128-byte rows; 32-byte units XOR-swizzled by f(k) = (k>>1 & 1) | (k>>3 & 1) << 1,
//            read with ds_read_b64_tr_b16 (hardware transpose); an mn-major A needs TM = 64.
// LDS is lane-linear for the DMA, so both swizzles sit on the per-lane SOURCE address and again on the fragment read.
#include "gemm_common.h"

namespace {

constexpr int SK = 256;                      // k chunk
constexpr int STN = 64;                      // tile columns

__device__ __attribute__((aligned(16))) unsigned int dvt_small_zero16[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ int swz_mn64(int k) { return ((k >> 1) & 1) | (((k >> 3) & 1) << 1); }

// DMA one operand chunk (ROWS x 256 k) into LDS, 1 KiB per wave-instruction; k >= k_lim and rows >= mn_lim read zeros
// (a clamped row would do for rows -- they are never stored -- but zeros keep the bias-gradient sums exact).
// ck = k length of the chunk: SK, or (mn-major images only: their rows are k) any multiple of 32 up to 512.
template <bool KMAJOR, int ROWS>
__device__ __forceinline__ void dma_chunk(const bf16* __restrict__ base, int64_t ld, int mn0, int mn_lim, int k0, int k_lim,
                                          char* img, int wid, int lane, int ck = SK) {
  const int ppw = KMAJOR ? ROWS * SK * 2 / 1024 / 4 : ck * ROWS * 2 / 1024 / 4;
  const bf16* zero = reinterpret_cast<const bf16*>(dvt_small_zero16);
#pragma unroll 4
  for (int i = 0; i < ppw; ++i) {
    const int piece = wid * ppw + i;
    const bf16* src;
    if (KMAJOR) {
      const int row = piece * 2 + (lane >> 5);            // 512-byte rows: two per piece
      const int c = (lane & 31) ^ (row & 15);
      const int gk = k0 + c * 8, gmn = mn0 + row;
      src = (gmn < mn_lim && gk < k_lim) ? base + (int64_t)gmn * ld + gk : zero;
    } else {
      const int k = piece * 8 + (lane >> 3);              // 128-byte rows: eight per piece
      const int cp = lane & 7;
      const int c = ((((cp >> 1) ^ swz_mn64(k)) << 1) | (cp & 1));
      const int gk = k0 + k, gmn = mn0 + c * 8;
      src = (gk < k_lim && gmn < mn_lim) ? base + (int64_t)gk * ld + gmn : zero;
    }
    dvt_dma16(src, img + piece * 1024);
  }
}

// MFMA operand fragment of 16 rows (k-major) / 16 columns (mn-major) starting at `base`, k-step kk (32 k each) of the
// chunk: lane (g, li) gets element j <-> (base + li, kk*32 + 8g + j).
template <typename E, bool KMAJOR>
__device__ __forceinline__ typename Elem16<E>::v8 sfrag(const char* img, int base, int kk, int g, int li) {
  typedef typename Elem16<E>::v8 V8;
  if (KMAJOR) {
    const int row = base + li;
    const int c = kk * 4 + g;
    return *reinterpret_cast<const V8*>(img + row * (SK * 2) + ((c ^ (row & 15)) << 4));
  } else {
    const int q = li >> 2, pp = li & 3;
    const int u = base >> 4;
    typename Elem16<E>::v4 half[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int k = kk * 32 + 8 * g + 4 * hf + q;
      half[hf] = Elem16<E>::tr_read(img + k * 128 + ((u ^ swz_mn64(k)) << 5) + 8 * pp);
    }
    return __builtin_shufflevector(half[0], half[1], 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

template <int N> __device__ __forceinline__ void swait_vm() {
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// TM = 64: waves 2 x 2, each 32 x 32;  TM = 32: waves 1 x 4, each 32 x 16.
// One TM x 64 output tile: workgroup `bid` of the `nblk` that serve this product.
template <typename E, bool A_KMAJOR, bool B_KMAJOR, int TM>
__device__ __forceinline__ void gemm_small_tile(const GemmParams& p, const int bid, const int nblk, char* smem) {
  static_assert(A_KMAJOR || TM == 64, "an mn-major A panel is 64 wide");
  typedef typename Elem16<E>::v8 V8;
  constexpr int WM = TM / 32, WNC = 4 / WM, WCOLS = STN / WNC, NU = WCOLS / 16;
  constexpr int kA = TM * SK * 2, kB = STN * SK * 2, kStage = kA + kB;
  constexpr int kPPC = (kA + kB) / 1024 / 4;           // DMA instructions per wave per chunk
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WNC, wn = wid % WNC;
  const int g = lane >> 4, li = lane & 15;
  // XCD-aware tile mapping: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 names the L2), so all row
  // tiles of one column tile -- the readers of one 64-column weight panel -- go to one XCD: the panel is fetched from
  // HBM into ONE L2 instead of into up to eight (the weights are cold at every launch of a training step; the
  // activation panels, 264 rows in all, are the ones every XCD re-reads).  Placement is a speed matter only.
  int rt, ct;
  {
    const int tiles_m = nblk / p.tiles_n;
    if ((p.tiles_n & 7) == 0) {
      const int j = bid >> 3;
      ct = (bid & 7) + 8 * (j / tiles_m);
      rt = j % tiles_m;
    } else {
      rt = bid / p.tiles_n;
      ct = bid % p.tiles_n;
    }
  }
  const int m0 = rt * TM, n0 = ct * STN;
  // Weight-gradient form with K <= 512 (the temporal encoder: K = 264 rows): ONE chunk of K rounded up to 32 rows --
  // two 256-row chunks would spend half their DMA instructions and MFMA steps on zero rows.
  const int ck = (!A_KMAJOR && !B_KMAJOR && p.K <= 512) ? ((p.K + 31) & ~31) : SK;
  const int nch = (p.K + ck - 1) / ck;
  const int boff = (!A_KMAJOR && !B_KMAJOR) ? ck * 128 : kA;      // B image behind the A image of the same stage

  f32x4 acc[NU][2];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int ch) {
    char* st = smem + (ch & 1) * kStage;
    dma_chunk<A_KMAJOR, TM>(p.A, p.lda, m0, p.M, ch * ck, p.K, st, wid, lane, ck);
    dma_chunk<B_KMAJOR, STN>(p.B, p.ldb, n0, p.N, ch * ck, p.K, st + boff, wid, lane, ck);
  };
  // Epilogue operands (bias, residual / activation-derivative operand, the old value under accumulate) are requested
  // FIRST -- clamped addresses, no branches -- so that their round trip runs under the panel DMA instead of behind the
  // main loop (the epilogue was 2.9 k of the kernel's 10.7 k ticks, nearly all of it these dependent loads); being the
  // oldest vector-memory operations they do not disturb the counted waits of the DMA pipeline.
  typedef typename Elem16<E>::v4 V4;
  const bool want_ld = p.epilogue == DVT_EPI_RESIDUAL || p.epilogue == DVT_EPI_DGELU || p.epilogue == DVT_EPI_DRELU;
  const bool want_old = p.out_f32 && p.accumulate;
  f32x4 pre_bias[NU], pre_old[2][NU];
  f32x4 pre_ld[2][NU];                              // residual / derivative operand as fp32 (16-bit or, residual only, fp32 in memory)
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int n = min(n0 + wn * WCOLS + u * 16 + 4 * g, p.N - 4);
    pre_bias[u] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int m = min(m0 + wm * 32 + t * 16 + li, p.M - 1);
      pre_ld[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
      pre_old[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (want_ld && p.epilogue == DVT_EPI_RESIDUAL && p.res_f32) {
        pre_ld[t][u] = *reinterpret_cast<const f32x4*>((const float*)p.residual + (int64_t)m * p.ldr + n);
      } else if (want_ld) {
        const E* src = p.epilogue == DVT_EPI_RESIDUAL ? (const E*)p.residual + (int64_t)m * p.ldr + n
                                                      : (const E*)p.aux + (int64_t)m * p.ldaux + n;
        const V4 h = *reinterpret_cast<const V4*>(src);
#pragma unroll
        for (int r = 0; r < 4; ++r) pre_ld[t][u][r] = (float)h[r];
      }
      if (want_old) pre_old[t][u] = *reinterpret_cast<const f32x4*>((const float*)p.C + (int64_t)m * p.ldc + n);
    }
  }
  issue(0);
  if (nch > 1) issue(1);

  // fused bias gradient (weight-gradient form): colsum[m] = sum_k A(m, k) for the tiles of the first tile column
  const bool do_cs = !A_KMAJOR && p.colsum_slab != nullptr && n0 == 0;
  float cs = 0.f;

  for (int ch = 0; ch < nch; ++ch) {
    if (ch + 1 < nch) swait_vm<kPPC>(); else swait_vm<0>();   // chunk ch landed (ch + 1 may stay in flight; then ck == SK)
    __builtin_amdgcn_s_barrier();
    const char* sa = smem + (ch & 1) * kStage;
    const char* sb = sa + boff;
    auto kstep = [&](int kk) {
      V8 af[2], bfr[NU];
#pragma unroll
      for (int t = 0; t < 2; ++t) af[t] = sfrag<E, A_KMAJOR>(sa, wm * 32 + t * 16, kk, g, li);
#pragma unroll
      for (int u = 0; u < NU; ++u) bfr[u] = sfrag<E, B_KMAJOR>(sb, wn * WCOLS + u * 16, kk, g, li);
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[u][t] = Elem16<E>::mma(bfr[u], af[t], acc[u][t]);
    };
    if (ck == SK) {                                        // the common 256-deep chunk: straight-line code, the fragment
#pragma unroll                                             // reads of later steps are issued under the MFMAs of earlier ones
      for (int kk = 0; kk < SK / 32; ++kk) kstep(kk);
    } else {
#pragma unroll 4
      for (int kk = 0; kk < ck / 32; ++kk) kstep(kk);
    }
    if (do_cs && tid < TM) {                               // thread = one column m of the [k][64] image
#pragma unroll 8
      for (int k = 0; k < ck; ++k)
        cs += (float)*reinterpret_cast<const E*>(sa + k * 128 + (((tid >> 4) ^ swz_mn64(k)) << 5) + (tid & 15) * 2);
    }
    if (ch + 2 < nch) {
      __builtin_amdgcn_s_barrier();                        // everybody finished reading this stage
      issue(ch + 2);
    }
  }
  if (do_cs && tid < TM && m0 + tid < p.M) {
    float* o = p.colsum_slab + m0 + tid;                   // (here: the final bias gradient, not a slab)
    *o = p.accumulate_colsum ? *o + cs : cs;
  }

  // ---- epilogue on the accumulators: lane (g, li) holds C[row = .. + li][col = .. + 4g .. 4g+3].
  // ONE switch over the epilogue kind with the element loops inside it: the per-element form (a runtime switch around
  // every value) compiled to a scalar branch chain per element -- 2.4 k of the kernel's 10 k ticks at eight elements.
  float v[2][NU][4], pre[2][NU][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[t][u][r] = acc[u][t][r] * p.alpha;
        pre[t][u][r] = 0.f;
      }
#define DVT_SMALL_EPI(EXPR)                                  \
  _Pragma("unroll") for (int t = 0; t < 2; ++t)              \
  _Pragma("unroll") for (int u = 0; u < NU; ++u)             \
  _Pragma("unroll") for (int r = 0; r < 4; ++r) {            \
    const float bi = pre_bias[u][r], ld = pre_ld[t][u][r]; \
    float& x = v[t][u][r];                                   \
    (void)bi; (void)ld;                                      \
    EXPR;                                                    \
  }
  switch (p.epilogue) {
    case DVT_EPI_GELU: DVT_SMALL_EPI(x = gelu_erf_both_f(x + bi, pre[t][u][r])) break;
    case DVT_EPI_RELU: DVT_SMALL_EPI(x = fmaxf(x + bi, 0.f)) break;
    case DVT_EPI_RESIDUAL: DVT_SMALL_EPI(x = x + bi + ld) break;
    case DVT_EPI_DGELU: DVT_SMALL_EPI(x *= ld) break;
    case DVT_EPI_DRELU: DVT_SMALL_EPI(x = ld > 0.f ? x : 0.f) break;
    default: DVT_SMALL_EPI(x += bi) break;
  }
#undef DVT_SMALL_EPI
  const bool save_pre = p.epilogue == DVT_EPI_GELU && p.aux != nullptr;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int m = m0 + wm * 32 + t * 16 + li;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int n = n0 + wn * WCOLS + u * 16 + 4 * g;
      if (m >= p.M || n >= p.N) continue;                  // N % 8 == 0: a group of four is inside or outside
      if (save_pre) {
        V4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (E)pre[t][u][r];
        *reinterpret_cast<V4*>((E*)p.aux + (int64_t)m * p.ldaux + n) = o;
      }
      if (p.out_f32) {
        float* o = (float*)p.C + (int64_t)m * p.ldc + n;
        f32x4 w = {v[t][u][0], v[t][u][1], v[t][u][2], v[t][u][3]};
        if (p.accumulate) w += pre_old[t][u];
        *reinterpret_cast<f32x4*>(o) = w;
      } else {
        V4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (E)v[t][u][r];
        *reinterpret_cast<V4*>((E*)p.C + (int64_t)m * p.ldc + n) = o;
      }
    }
  }
}

template <typename E, bool A_KMAJOR, bool B_KMAJOR, int TM>
__global__ __launch_bounds__(256) void gemm_small_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_small_tile<E, A_KMAJOR, B_KMAJOR, TM>(p, blockIdx.x, gridDim.x, smem);
}

// The two products of one Linear's backward -- weight gradient (both operands mn-major, 64-row tiles) and data gradient
// (A k-major, B mn-major) -- in ONE launch: they are independent (both read dy), each is a fraction of a round of
// workgroups, and between dependent launches of a few microseconds the launch itself is what costs.
template <typename E, int TM2>
__global__ __launch_bounds__(256) void gemm_small_pair_kernel(const GemmParams pw, const GemmParams pd, const int nblk_w) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x < nblk_w) gemm_small_tile<E, false, false, 64>(pw, blockIdx.x, nblk_w, smem);
  else gemm_small_tile<E, true, false, TM2>(pd, blockIdx.x - nblk_w, gridDim.x - nblk_w, smem);
}

template <typename E, int TM2>
int launch_small_pair(const GemmParams& pw_in, const GemmParams& pd_in, hipStream_t st) {
  constexpr int kSmem = 2 * (64 + STN) * SK * 2;
  GemmParams pw = pw_in, pd = pd_in;
  pw.tiles_n = (int)dvt_cdiv(pw.N, STN);
  pd.tiles_n = (int)dvt_cdiv(pd.N, STN);
  static DvtLdsAttr attr_set;
    dvt_lds_attr(attr_set, (const void*)gemm_small_pair_kernel<E, TM2>, kSmem);
  const int nw = (int)(dvt_cdiv(pw.M, 64) * pw.tiles_n), nd = (int)(dvt_cdiv(pd.M, TM2) * pd.tiles_n);
  hipLaunchKernelGGL((gemm_small_pair_kernel<E, TM2>), dim3((unsigned)(nw + nd)), dim3(256), kSmem, st, pw, pd, nw);
  DVT_LAUNCH_CHECK("dvt_gemm_pair(small)");
  return DVT_OK;
}

template <typename E, bool AK, bool BK, int TM>
int launch_small(const GemmParams& pin, hipStream_t st) {
  constexpr int kSmem = 2 * (TM + STN) * SK * 2;
  static_assert(kSmem <= 160 * 1024, "LDS budget");
  GemmParams p = pin;
  p.tiles_n = (int)dvt_cdiv(p.N, STN);
  static DvtLdsAttr attr_set;
    dvt_lds_attr(attr_set, (const void*)gemm_small_kernel<E, AK, BK, TM>, kSmem);
  const dim3 grid((unsigned)(dvt_cdiv(p.M, TM) * p.tiles_n)), block(256);
  hipLaunchKernelGGL((gemm_small_kernel<E, AK, BK, TM>), grid, block, kSmem, st, p);
  DVT_LAUNCH_CHECK("dvt_gemm(small)");
  return DVT_OK;
}

template <typename E>
int launch_small_any(const GemmParams& p, bool ak, bool bk, int tm, hipStream_t st) {
  if (ak && bk) return tm == 32 ? launch_small<E, true, true, 32>(p, st) : launch_small<E, true, true, 64>(p, st);
  if (ak && !bk) return tm == 32 ? launch_small<E, true, false, 32>(p, st) : launch_small<E, true, false, 64>(p, st);
  if (!ak && !bk) return launch_small<E, false, false, 64>(p, st);
  return 1;
}

}  // namespace

// Tile height for a launch-bound shape: 32-row tiles when 64-row ones would leave most CUs without a workgroup.
// Returns 0 when this kernel does not take the shape (layouts: A k-major with either B, or both mn-major).
int dvt_gemm_small_tile(int64_t M, int64_t N, bool a_kmajor, bool b_kmajor) {
  if (!a_kmajor && b_kmajor) return 0;
  const int64_t t64 = dvt_cdiv(M, 64) * dvt_cdiv(N, STN);
  if (!a_kmajor) return 64;
  return t64 >= 128 ? 64 : 32;
}

// p.colsum_slab (weight-gradient form): the FINAL bias gradient vector here (this kernel does not split K);
// p.accumulate_colsum selects += .  Returns DVT_OK, a negative status, or 1 when there is no instantiation.
int dvt_gemm_small_launch(const GemmParams& p, bool a_kmajor, bool b_kmajor, hipStream_t st) {
  const int tm = dvt_gemm_small_tile(p.M, p.N, a_kmajor, b_kmajor);
  if (tm == 0) return 1;
  if (p.elem == DVT_F16) return launch_small_any<f16>(p, a_kmajor, b_kmajor, tm, st);
  return launch_small_any<bf16>(p, a_kmajor, b_kmajor, tm, st);
}

// Weight gradient (A, B mn-major) + data gradient (A k-major, B mn-major) of one Linear in one launch; 1 when the pair
// has no instantiation (the caller then launches them one after the other).
int dvt_gemm_small_launch_pair(const GemmParams& pw, const GemmParams& pd, hipStream_t st) {
  if (pw.elem != pd.elem) return 1;
  const int tm2 = dvt_gemm_small_tile(pd.M, pd.N, true, false);
  if (tm2 == 0 || dvt_gemm_small_tile(pw.M, pw.N, false, false) != 64) return 1;
  if (pw.elem == DVT_F16) return tm2 == 32 ? launch_small_pair<f16, 32>(pw, pd, st) : launch_small_pair<f16, 64>(pw, pd, st);
  return tm2 == 32 ? launch_small_pair<bf16, 32>(pw, pd, st) : launch_small_pair<bf16, 64>(pw, pd, st);
}
