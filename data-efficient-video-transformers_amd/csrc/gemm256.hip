// gemm256.hip -- 256x256x64 bf16 MFMA GEMM with LDS-DMA staging (large shapes).
//
// One workgroup = 8 waves (2 x 4), one 256x256 output tile, fp32 accumulate:
//   * both operand tiles go HBM/L2 -> LDS directly (global_load_lds_dwordx4, 1 KiB
//     per wave-instruction, no VGPR staging); LDS is lane-linear for the DMA, so
//     the bank-conflict swizzles are applied on the per-lane SOURCE address and
//     again on the fragment read (both are the same XOR involution).  The DMA is
//     issued from inline asm (dvt_dma16): seen by the compiler it is a pending LDS
//     write, and the wait-count pass then puts vmcnt(0) in front of the next ds_read,
//     i.e. right behind the issue;
//   * two 64 KiB stages: the DMA of k-tile t+1 is in flight during all 64 MFMAs
//     per wave of k-tile t; hand-placed s_waitcnt vmcnt + raw s_barrier, one per k-tile;
//   * fragment reads software-pipelined one 16-MFMA block ahead (double register buffer);
//   * each wave owns 128 x 64 of the tile (8 x 4 accumulators of
//     v_mfma_f32_16x16x32_bf16); k-major operands are read with ds_read_b128,
//     mn-major operands (dgrad's W, wgrad's dY and x) with ds_read_b64_tr_b16;
//   * epilogue: every wave stages its own accumulators through a private LDS
//     region (no workgroup barrier) and stores whole 128-byte row segments with
//     bias / GELU / GELU' / ReLU / residual fused; outputs of >= 180 MB are written
//     with streaming (nt) stores so that they do not evict the operand panels from L2;
//   * split-K over blockIdx.z writes fp32 slabs (summed by splitk_reduce_kernel).
// Requirements (else gemm.hip's 128x128 register-staged kernel is used):
//   K and every split a multiple of 64, all leading dimensions multiples of 8.
#include "gemm_common.h"
#include <cstdlib>

// Translation units of this file: 0 (gemm256.hip itself) = configurations 0, 1, 3, 4 and every entry point; 1
// (gemm256_pp.hip includes this file) = the antiphase configuration 5 alone.  Instantiated beside the others it changed
// THEIR register allocation (spills in the 16-wave and implicit weight-gradient kernels), so it gets its own module.
#ifndef DVT_GEMM256_UNIT
#define DVT_GEMM256_UNIT 0
#endif

namespace {

// Configurations of one kernel (0 is the default; 1 serves narrow convolutions; 3 the GELU / GELU' epilogues):
//   CFG 0  256x256 tile, BK=64, 2 LDS stages (128 KiB), 8 waves (2x4): 1 workgroup / CU.
//          Highest arithmetic intensity per L2 byte; used when K is long enough that the
//          un-overlapped epilogue burst does not matter.
//   CFG 1  256x128 tile, BK=32, 3 LDS stages (72 KiB), 4 waves (2x2): 2 workgroups / CU,
//          so one workgroup's epilogue (stores) overlaps the other's main loop.  Used for
//          convolutions with <= 128 output channels (measured slower than CFG 0 on every Linear shape).
//   CFG 3  256x256 tile, BK=64, 16 waves of 64x64 (4 per SIMD, 128 VGPRs, no fragment double buffer): twice the
//          memory-operation concurrency in the epilogue, a slower main loop.
//   CFG 6  CFG 4's tile with BK=64 and 2 stages (80 KiB): a pixel's 64 channels of one filter tap are one 128-byte line per
//          gather instead of two half lines (convolutions with <= 64 output channels and C % 64 == 0).
//   CFG 8  CFG 5 with 224-row tiles (7 m sub-tiles per wave row): N = 512 launches at M = 50,432 are 394 tiles of 256 rows =
//          two rounds on 256 CUs for 1.54 rounds of work; 452 tiles of 224 rows are two rounds of 7/8 the work each.
//   CFG 5  CFG 0's tile with the two wave rows in antiphase (one reads a 32-deep slice's fragments while its SIMD partner
//          runs the previous slice's MFMAs out of registers; four phase barriers per k-tile): +3..6 % on most shapes.
//   CFG 4  256x64 tile, BK=32, 3 LDS stages (60 KiB), 4 waves (4x1) of 64x64: convolutions with <= 64 output channels
//          (the stem and layer 1 of the ResNets: a 128-wide tile would spend half its MFMAs on padding columns).
template <int CFG> struct Cfg;
template <> struct Cfg<0> { enum { TM = 256, TN = 256, TK = 64, NW = 8, WN = 4, NSTG = 2, PP = 0 }; };
template <> struct Cfg<1> { enum { TM = 256, TN = 128, TK = 32, NW = 4, WN = 2, NSTG = 3, PP = 0 }; };
template <> struct Cfg<3> { enum { TM = 256, TN = 256, TK = 64, NW = 16, WN = 4, NSTG = 2, PP = 0 }; };  // 16 waves of 64 x 64: 4 per SIMD
template <> struct Cfg<4> { enum { TM = 256, TN = 64, TK = 32, NW = 4, WN = 1, NSTG = 3, PP = 0 }; };    // 4 waves of 64 x 64, 60 KiB: 2 workgroups / CU
template <> struct Cfg<6> { enum { TM = 256, TN = 64, TK = 64, NW = 4, WN = 1, NSTG = 2, PP = 0 }; };    // CFG 4 with 64-deep k-tiles, 2 x 40 KiB: 2 workgroups / CU
template <> struct Cfg<7> { enum { TM = 256, TN = 128, TK = 64, NW = 8, WN = 2, NSTG = 2, PP = 0 }; };   // CFG 1's tile with 64-deep k-tiles: 8 waves of 64 x 64, 96 KiB
template <> struct Cfg<8> { enum { TM = 224, TN = 256, TK = 64, NW = 8, WN = 4, NSTG = 2, PP = 1 }; };   // CFG 5 with 7 sub-tiles per wave row
template <> struct Cfg<9> { enum { TM = 128, TN = 128, TK = 64, NW = 4, WN = 2, NSTG = 2, PP = 0 }; };   // 4 waves of 64 x 64, 64 KiB: 2 workgroups / CU; convolutions with few output rows
template <> struct Cfg<10> { enum { TM = 128, TN = 128, TK = 64, NW = 4, WN = 2, NSTG = 4, PP = 0 }; };  // CFG 9 with three k-tiles in flight, 128 KiB: grids of at most one workgroup per CU (deep K, operand latency exposed)
template <> struct Cfg<5> { enum { TM = 256, TN = 256, TK = 64, NW = 8, WN = 4, NSTG = 2, PP = 1 }; };   // CFG 0 with the two wave rows in antiphase

constexpr int kEpiStride = 64 + 4;               // floats per staged row
constexpr int kEpiBytes = 32 * kEpiStride * 4;   // 8,704 B per wave

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// 16-byte-chunk swizzle of a k-major image row (XOR involution, also used on the read side)
template <int TK> __device__ __forceinline__ int swz_k(int row) { return TK == 64 ? (row & 7) : ((row >> 1) & 3); }

// 16 zero bytes in global memory: the DMA source of every padded (out-of-image) tap of the implicit convolution
static __device__ __attribute__((aligned(16))) unsigned int dvt_zero16[4] = {0u, 0u, 0u, 0u};

// DMA one operand tile into LDS in 1 KiB pieces (one wave-instruction each).
//   k-major : image [ROWS][TK],  row = TK*2 bytes
//   mn-major: image [TK][ROWS],  row = ROWS*2 bytes, 32-byte units XOR-swizzled by swz_mn(k)
// KLIM (mn-major images only): k rows >= k_lim read the zero page -- the weight gradient of a convolution whose pixel
// count is not a multiple of the k-tile (the last k-tile of the last slice is ragged).
template <bool KMAJOR, int ROWS, int TK, int NW, bool KLIM = false>
__device__ __forceinline__ void dma_tile(const bf16* __restrict__ base, int64_t ld, int mn0, int mn_lim,
                                         int k0, char* tile, int wid, int lane, int k_lim = 0) {
  constexpr int PIECES = ROWS * TK * 2 / 1024;
  constexpr int PPW = (PIECES + NW - 1) / NW;        // (224-row tiles: 28 pieces, 4 each for waves 0 .. 6)
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int piece = wid * PPW + i;
    if (PIECES % NW != 0 && piece >= PIECES) continue;   // wave-uniform
    const bf16* src;
    if (KMAJOR) {
      constexpr int CPR = TK / 8;               // 16-byte chunks per row
      constexpr int RPP = 64 / CPR;             // rows per piece
      const int row = piece * RPP + lane / CPR;
      const int c = (lane % CPR) ^ swz_k<TK>(row);
      int grow = mn0 + row;
      grow = grow < mn_lim ? grow : mn_lim - 1;
      src = base + (int64_t)grow * ld + k0 + c * 8;
    } else {
      constexpr int CPR = ROWS / 8;
      constexpr int RPP = 64 / CPR;
      const int k = piece * RPP + lane / CPR;
      const int cp = lane % CPR;
      const int c = ((((cp >> 1) ^ swz_mn_r<ROWS>(k)) << 1) | (cp & 1));
      int gmn = mn0 + c * 8;
      gmn = gmn < mn_lim ? gmn : mn_lim - 8;
      src = base + (int64_t)(k0 + k) * ld + gmn;
      if (KLIM && k0 + k >= k_lim) src = reinterpret_cast<const bf16*>(dvt_zero16);
    }
    dvt_dma16(src, tile + piece * 1024);
  }
}


// Implicit-GEMM A operand: row = output pixel (n, ho, wo), k = (ki, kj, c) of an NHWC map x[N, H, W, C].
// C % TK == 0, so a whole k-tile lies inside one filter tap: per k-tile a lane only adds the tap's (ki, kj) to the
// pixel coordinates it pre-computed once for its PPW rows, checks the bounds and points padded taps at dvt_zero16.
template <int ROWS, int TK, int NW>
struct ConvRows {
  enum { PIECES = ROWS * TK * 2 / 1024, PPW = PIECES / NW, CPR = TK / 8, RPP = 64 / CPR };
  int pix[PPW], h0[PPW], w0[PPW], coff[PPW];
  // C % TK == 0: the k-tiles of a workgroup are requested in order, TK channels apart, and C / TK of them share a filter tap --
  // the lane's source addresses are carried and advanced by TK; the tap's bounds checks, the pixel arithmetic and the 64-bit
  // multiply run once per TAP instead of once per k-tile (per k-tile they were ~12 vector instructions per 1-KiB piece, one
  // of them quarter-rate: as many issue cycles as half the k-tile's MFMAs at one wave per SIMD, profiles/r06_implicit_l34_counters.md)
  const bf16* src_[PPW];
  unsigned okm_;
  int c0_ = -1, ki_ = 0, kj_ = 0;
  __device__ __forceinline__ void init(const GemmParams& p, int m0, int wid, int lane) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int row = (wid * PPW + i) * RPP + lane / CPR;
      int grow = m0 + row;
      grow = grow < p.M ? grow : p.M - 1;
      const int hw = p.cHo * p.cWo;
      const int n = grow / hw, r = grow - n * hw;
      const int ho = r / p.cWo, wo = r - ho * p.cWo;
      pix[i] = n * p.cH * p.cW;
      h0[i] = ho * p.csh - p.cph;
      w0[i] = wo * p.csw - p.cpw;
      coff[i] = ((lane % CPR) ^ swz_k<TK>(row)) * 8;
    }
  }
  __device__ __forceinline__ void dma(const GemmParams& p, int k0, char* tile, int wid) {
    if (p.cC == 8) {
      // Stem form (the 3 image channels zero-extended to 8, custom_resnet.py:100): a 16-byte chunk is one (pixel, tap), so
      // the chunks of a k-tile are consecutive taps and every lane derives its own tap; k >= kh*kw*8 (K is rounded up
      // to the k-tile) and padded taps read the zero page.  tap / kw by multiplication (taps < 2^10, kw < 2^6).
      const int ntap = p.ckh * p.ckw, magic = (65536 + p.ckw - 1) / p.ckw;
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const int tap = (k0 + coff[i]) >> 3;
        const int ki = (tap * magic) >> 16, kj = tap - ki * p.ckw;
        const int hi = h0[i] + ki, wi = w0[i] + kj;
        const bool ok = tap < ntap && (unsigned)hi < (unsigned)p.cH && (unsigned)wi < (unsigned)p.cW;
        const bf16* src = ok ? p.A + (int64_t)(pix[i] + hi * p.cW + wi) * 8 : reinterpret_cast<const bf16*>(dvt_zero16);
        dvt_dma16(src, tile + (wid * PPW + i) * 1024);
      }
      return;
    }
    if (p.cC % TK) {
      // C % 8 == 0 only (R(2+1)D-18's 144 mid planes, video_resnet.py:69): a k-tile may straddle filter taps, so every lane
      // derives the tap of its own 16-byte chunk (chunk index / chunks per tap, by multiplication); k >= kh*kw*C (K is
      // rounded up to the k-tile) reads the zero page
      const unsigned cpc = (unsigned)p.cC >> 3, ntap = (unsigned)(p.ckh * p.ckw);
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const unsigned kq = (unsigned)(k0 + coff[i]) >> 3;
        const unsigned tap = __umulhi(kq, p.cmc);
        const int c = (int)(kq - tap * cpc) * 8;
        const unsigned ki = p.ckw == 1 ? tap : __umulhi(tap, p.cmk);
        const int hi = h0[i] + (int)ki, wi = w0[i] + (int)(tap - ki * p.ckw);
        const bool ok = tap < ntap && (unsigned)hi < (unsigned)p.cH && (unsigned)wi < (unsigned)p.cW;
        const bf16* src = ok ? p.A + ((int64_t)(pix[i] + hi * p.cW + wi) * p.cC + c) : reinterpret_cast<const bf16*>(dvt_zero16);
        dvt_dma16(src, tile + (wid * PPW + i) * 1024);
      }
      return;
    }
#ifdef DVT_CONV_ROWS_PER_TILE
    const int tap = k0 / p.cC, c0 = k0 - tap * p.cC;
    const int ki = tap / p.ckw, kj = tap - ki * p.ckw;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int hi = h0[i] + ki, wi = w0[i] + kj;
      const bool ok = (unsigned)hi < (unsigned)p.cH && (unsigned)wi < (unsigned)p.cW;
      const bf16* src = ok ? p.A + ((int64_t)(pix[i] + hi * p.cW + wi) * p.cC + c0 + coff[i])
                           : reinterpret_cast<const bf16*>(dvt_zero16);
      dvt_dma16(src, tile + (wid * PPW + i) * 1024);
    }
#else
    if (c0_ < 0 || c0_ >= p.cC) {                 // (wave-uniform) the first request, or the first k-tile of the next tap
      if (c0_ < 0) {
        const int tap = k0 / p.cC;
        c0_ = k0 - tap * p.cC;
        ki_ = tap / p.ckw;
        kj_ = tap - ki_ * p.ckw;
      } else {
        c0_ = 0;
        if (++kj_ == p.ckw) { kj_ = 0; ++ki_; }
      }
      okm_ = 0u;
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const int hi = h0[i] + ki_, wi = w0[i] + kj_;
        const bool ok = (unsigned)hi < (unsigned)p.cH && (unsigned)wi < (unsigned)p.cW;
        src_[i] = p.A + ((int64_t)(pix[i] + hi * p.cW + wi) * p.cC + c0_ + coff[i]);     // (formed, not read, when !ok)
        okm_ |= ok ? 1u << i : 0u;
      }
    }
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      dvt_dma16((okm_ >> i) & 1u ? src_[i] : reinterpret_cast<const bf16*>(dvt_zero16), tile + (wid * PPW + i) * 1024);
      src_[i] += TK;
    }
    c0_ += TK;
#endif
  }
};

// Implicit-GEMM mn-major operand (weight gradient): image [TK k-rows][ROWS m-columns], k = output pixel row of the
// layer, m = (ki, kj, c) column of the never-materialised column matrix.  A lane's 16-byte chunk is 8 channels of one
// tap of one pixel; its (tap, channel) is fixed for the whole K loop, only the pixel changes per k-tile.  The k-tiles of
// a workgroup are requested in order, TK pixels apart, so the pixel coordinates (image, ho, wo) are carried and advanced
// by TK = (dqq * Ho + dqr) * Wo + dr per request instead of being divided out of the row index every time (two integer
// divisions per chunk and k-tile were more vector work than the k-tile's MFMAs at 64 output channels).
template <int ROWS, int TK, int NW>
struct ConvColsMN {
  enum { PIECES = ROWS * TK * 2 / 1024, PPW = PIECES / NW, CPR = ROWS / 8, RPP = 64 / CPR };
  int wo[PPW], ho[PPW], nb[PPW], tc[PPW];   // tc = channel | kj << 16 | ki << 24
  int krow[PPW];                            // the pixel row this lane's chunk of the NEXT k-tile names (rows >= p.K: zeros)
  int dr, dqr, dqq;
  __device__ __forceinline__ void init(const GemmParams& p, int m0, int kbeg, int wid, int lane) {
    const int dq = TK / p.cWo;
    dr = TK - dq * p.cWo;
    dqq = dq / p.cHo;
    dqr = dq - dqq * p.cHo;
    const int hw = p.cHo * p.cWo;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int k = (wid * PPW + i) * RPP + lane / CPR;
      const int cp = lane % CPR;
      const int c = ((((cp >> 1) ^ swz_mn_r<ROWS>(k)) << 1) | (cp & 1));
      int gmn = m0 + c * 8;
      gmn = gmn < p.M ? gmn : p.M - 8;
      const int tap = gmn / p.cC;
      const int ki = tap / p.ckw, kj = tap - ki * p.ckw;
      tc[i] = (gmn - tap * p.cC) | (kj << 16) | (ki << 24);
      const int row = kbeg + k;
      krow[i] = row;
      const int n = row / hw, r = row - n * hw;
      ho[i] = r / p.cWo;
      wo[i] = r - ho[i] * p.cWo;
      nb[i] = n * p.cH * p.cW;
    }
  }
  // the NEXT k-tile of this workgroup (call order = k order)
  __device__ __forceinline__ void dma(const GemmParams& p, char* tile, int wid) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int hi = ho[i] * p.csh - p.cph + (tc[i] >> 24), wi = wo[i] * p.csw - p.cpw + ((tc[i] >> 16) & 0xff);
      const bool ok = (unsigned)hi < (unsigned)p.cH && (unsigned)wi < (unsigned)p.cW && krow[i] < p.K;
      const bf16* src = ok ? p.A + ((int64_t)(nb[i] + hi * p.cW + wi) * p.cC + (tc[i] & 0xffff))
                           : reinterpret_cast<const bf16*>(dvt_zero16);
      dvt_dma16(src, tile + (wid * PPW + i) * 1024);
      krow[i] += TK;
      wo[i] += dr;
      ho[i] += dqr;
      nb[i] += dqq * p.cH * p.cW;
      if (wo[i] >= p.cWo) { wo[i] -= p.cWo; ho[i] += 1; }
      if (ho[i] >= p.cHo) { ho[i] -= p.cHo; nb[i] += p.cH * p.cW; }
    }
  }
};

// MFMA operand fragment of 16 rows (k-major) / 16 columns (mn-major) starting at `base`,
// k-step kk (32 k each): lane (g, li) gets element j <-> (base + li, kk*32 + 8g + j).
template <typename E, bool KMAJOR, int ROWS, int TK>
__device__ __forceinline__ typename Elem16<E>::v8 frag(const char* tile, int base, int kk, int g, int li) {
  typedef typename Elem16<E>::v8 V8;
  if (KMAJOR) {
    const int row = base + li;
    const int c = kk * 4 + g;
    return *reinterpret_cast<const V8*>(tile + row * (TK * 2) + ((c ^ swz_k<TK>(row)) << 4));
  } else {
    const int q = li >> 2, pp = li & 3;
    const int u = base >> 4;
    typename Elem16<E>::v4 half[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int k = kk * 32 + 8 * g + 4 * hf + q;
      const char* a = tile + k * (ROWS * 2) + ((u ^ swz_mn_r<ROWS>(k)) << 5) + 8 * pp;
      half[hf] = Elem16<E>::tr_read(a);
    }
    // concatenation, not element inserts: the two 64-bit reads land in adjacent VGPR pairs
    return __builtin_shufflevector(half[0], half[1], 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

__device__ __forceinline__ void wave_lds_fence() {
  // LDS operations of one wave complete in order; only the compiler must not reorder
  // across this point.  (A wavefront-scope fence would also emit s_waitcnt vmcnt(0)
  // and serialise the wave behind its outstanding global stores.)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

template <int N> __device__ __forceinline__ void wait_vm() {
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#ifndef DVT_GEMM_NT
#define DVT_GEMM_NT 1
#endif
#ifndef DVT_C_STORE          // (a dev harness that includes this file may define its own: tools/dev/gemm_storewave_price.hip)
#if DVT_GEMM_NT
#define DVT_C_STORE(ptr, vals) do { if (p.stream_out) store8_nt<E>(ptr, vals); else store8<E>(ptr, vals); } while (0)
#else
#define DVT_C_STORE(ptr, vals) store8<E>(ptr, vals)
#endif
#endif

// OUT: 0 = C in bf16 with the fused epilogue EPI, 1 = C in fp32 (optionally accumulated),
//      2 = raw fp32 split-K slab.  EPI and OUT are compile-time so that the unrolled
//      epilogue stays a few hundred instructions (a runtime switch replicated over the
//      4 x 4 unrolled passes was ~19k ISA lines per kernel and thrashed the I-cache).
enum { OUT_BF16 = 0, OUT_F32 = 1, OUT_SLAB = 2 };

template <typename E, bool A_KMAJOR, bool B_KMAJOR, int CFG, int EPI, int OUT, bool A_CONV = false>
__global__ __launch_bounds__(Cfg<CFG>::NW * 64, Cfg<CFG>::NW == 16 ? 4 : 2) void gemm_dma_kernel(const GemmParams p) {
  // A_CONV: the A operand is gathered from an NHWC map (k-major: forward / data gradient; mn-major: weight gradient)
  typedef Cfg<CFG> C;
  typedef typename Elem16<E>::v8 V8;
  constexpr int TM = C::TM, TN = C::TN, TK = C::TK, NW = C::NW, WN = C::WN, NSTG = C::NSTG;
  constexpr int kATile = TM * TK * 2, kBTile = TN * TK * 2, kStage = kATile + kBTile;
  constexpr int kPPT = (kATile + kBTile) / 1024 / NW;   // DMA instructions per thread per k-tile
  constexpr bool kKLim = A_CONV && !A_KMAJOR && !B_KMAJOR;   // convolution weight gradient: K = output pixels, any count
  constexpr int WROWS = TM / (NW / WN);                 // rows of the tile owned by one wave (x 64 columns)
  constexpr int MT = WROWS / 16, NTH = MT / 4;          // m sub-tiles per wave, 64-row halves per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;      // every wave owns WROWS x 64
  const int g = lane >> 4, li = lane & 15;

  // XCD-aware order over the combined (K-slice, tile) space.  Blocks b and b+8 share an XCD
  // (round-robin dispatch); each XCD gets one contiguous run of the linear index
  // id = slice * ntiles + tile, bijective for any count.  Forward/dgrad (one slice):
  // consecutive n-tiles of an A row-panel hit the same L2.  Split-K weight gradients: an XCD
  // owns whole K-slices, so every byte of dY and x is fetched by exactly one XCD (PMC: the
  // per-plane remap re-read the x slice once per XCD, 2.3x the algorithmic bytes).
  // Workgroups past the tiles carry a split-K reduce of the previous launch (dvt_splitk_pending): they are dispatched
  // last, i.e. into the CUs the partial last round of tiles leaves idle.  (Launches with such a tail have one K slice.)
  const int ntiles = gridDim.x - p.pig_blocks;
  if (p.pig_blocks > 0 && (int)blockIdx.x >= ntiles) {
    splitk_reduce_f32_part(p.pig, (int64_t)(blockIdx.x - ntiles) * blockDim.x + threadIdx.x, (int64_t)p.pig_blocks * blockDim.x);
    return;
  }
  int tile, zsl;
  {
    const int total = ntiles * gridDim.z;
    const int b = blockIdx.z * ntiles + blockIdx.x, xcd = b & 7, q = total >> 3, r = total & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    zsl = id / ntiles;
    tile = id - zsl * ntiles;
  }
  const int m0 = (tile / p.tiles_n) * TM;
  const int n0 = (tile % p.tiles_n) * TN;
  const int kbeg = zsl * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nk = (kend - kbeg + TK - 1) / TK;      // (a multiple of TK except in the ragged last slice of a convolution weight gradient)

  f32x4 acc[4][MT];  // [u: n sub-tile][t: m sub-tile]
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fused bias gradient (weight-gradient launches only): each wave of the n-tile-0 workgroups
  // also accumulates sum_k A(m,k) for two of its eight m sub-tiles (wave wn takes t = 2wn, 2wn+1)
  constexpr bool kCanColsum = !A_KMAJOR && OUT == OUT_SLAB && WN == 4 && MT == 8;   // (a 16-wave build spills with it)
  constexpr int CS = MT / 4;                          // m sub-tiles per wave whose column sums this wave takes (wave wn: CS*wn ..)
  const bool do_cs = kCanColsum && p.colsum_slab != nullptr && n0 == 0;
  f32x4 csum[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  V8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (E)1.0f;

  ConvRows<TM, TK, NW> cv;
  ConvColsMN<TM, TK, NW> cvm;
  if (A_CONV && A_KMAJOR) cv.init(p, m0, wid, lane);
  if (A_CONV && !A_KMAJOR) cvm.init(p, m0, kbeg, wid, lane);

  if constexpr (C::PP) {
    // ---- antiphase main loop ("ping-pong").  Waves w and w + 4 share a SIMD (a workgroup's waves go to the SIMDs
    // cyclically), i.e. wave row 0 and wave row 1.  Each wave alternates a READ phase (the 12 fragments of one 32-deep
    // slice into registers, plus its share of the DMA for the k-tile after next) and an MFMA phase (the slice's 32 MFMAs,
    // operands already in registers), one phase out of step with its SIMD partner: after every phase barrier one of the two
    // waves of a SIMD starts MFMAs whose operands are there, so the matrix pipe does not wait for a barrier plus an LDS
    // round trip per k-tile as it does when both waves reach the k-tile barrier together.  Phase ph (4 per k-tile):
    //   row 0:  R(kt,0) M(kt,0) R(kt,1) M(kt,1)           row 1:  M(kt-1,1) R(kt,0) M(kt,0) R(kt,1)   [+ M(nk-1,1) at the end]
    // k-tile kt+1 is requested at the start of the wave's R(kt,0) (row 1 requests k-tile 1 in its idle phase 0; k-tile 0 is the
    // prologue) into the stage that held kt-1 (last read in phase 4kt-1), and must have landed by the barrier that ends
    // phase 4kt+3.
    static_assert(NSTG == 2 && TK == 64 && NW == 8 && (MT == 8 || MT == 7), "antiphase loop: 256x256x64 or 224x256x64, 8 waves");
    auto issue = [&](int kt) {
      char* st = smem + (kt & 1) * kStage;
      if (A_CONV && A_KMAJOR) cv.dma(p, kbeg + kt * TK, st, wid);
      else if (A_CONV) cvm.dma(p, st, wid);
      else dma_tile<A_KMAJOR, TM, TK, NW>(p.A, p.lda, m0, p.M, kbeg + kt * TK, st, wid, lane);
      dma_tile<B_KMAJOR, TN, TK, NW, kKLim>(p.B, p.ldb, n0, p.N, kbeg + kt * TK, st + kATile, wid, lane, p.K);
    };
    if (nk > 0) issue(0);
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    V8 fa[MT], fb[4];
    auto rd = [&](int kt, int kk) {
      const char* sa = smem + (kt & 1) * kStage;
      const char* sb = sa + kATile;
#pragma unroll
      for (int u = 0; u < 4; ++u) fb[u] = frag<E, B_KMAJOR, TN, TK>(sb, wn * 64 + u * 16, kk, g, li);
#pragma unroll
      for (int t = 0; t < MT; ++t) fa[t] = frag<E, A_KMAJOR, TM, TK>(sa, wm * WROWS + t * 16, kk, g, li);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto mm = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[u][t] = Elem16<E>::mma(fb[u], fa[t], acc[u][t]);
      if (kCanColsum && do_cs) {                     // static fragment indices (a runtime index would move fa to scratch)
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4)
          if (wn == w4) {
#pragma unroll
            for (int tt = 0; tt < CS; ++tt) csum[tt] = Elem16<E>::mma(ones, fa[CS * w4 + tt], csum[tt]);
          }
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    };
    auto bar = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    if (wm == 0) {
      for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) issue(kt + 1);
        rd(kt, 0); bar();
        mm(); bar();
        rd(kt, 1); bar();
        mm(); wait_vm<0>(); bar();
      }
    } else {
      for (int kt = 0; kt < nk; ++kt) {
        if (kt >= 1) mm();
        else if (nk > 1) issue(1);                   // phase 0 has no MFMAs for this row: its share of k-tile 1 goes out now
        bar();
        if (kt >= 1 && kt + 1 < nk) issue(kt + 1);
        rd(kt, 0); bar();
        mm(); bar();
        rd(kt, 1); wait_vm<0>(); bar();
      }
      if (nk > 0) mm();
    }
    // no barrier here: the last LDS read (row 1's R(nk-1,1)) completed before the last phase barrier, so row 0 starts its
    // epilogue (per-wave staging over the stage ring) while row 1 still runs its last 32 MFMAs out of registers
  }
  const int nko = C::PP ? 0 : nk;                 // the in-phase loop below (dead code in the antiphase build)
  // prologue: NSTG-1 k-tiles in flight
#pragma unroll
  for (int s = 0; s < NSTG - 1; ++s)
    if (s < nko) {
      if (A_CONV && A_KMAJOR) cv.dma(p, kbeg + s * TK, smem + s * kStage, wid);
      else if (A_CONV) cvm.dma(p, smem + s * kStage, wid);
      else dma_tile<A_KMAJOR, TM, TK, NW>(p.A, p.lda, m0, p.M, kbeg + s * TK, smem + s * kStage, wid, lane);
      dma_tile<B_KMAJOR, TN, TK, NW, kKLim>(p.B, p.ldb, n0, p.N, kbeg + s * TK, smem + s * kStage + kATile, wid, lane, p.K);
    }
  int st_cur = 0, st_nxt = NSTG - 1;           // ring positions of k-tile kt and kt+NSTG-1
  constexpr int NB = NTH * (TK / 32);
  constexpr bool kPipe = NW <= 8;                // 16 waves (4 per SIMD, 128 VGPRs) hide the read latency by occupancy instead
  V8 bfr[kPipe ? 2 : 1][4], af[kPipe ? 2 : 1][4];
  for (int kt = 0; kt < nko; ++kt) {
    // (1) this wave's pieces of k-tile kt have landed (younger k-tiles may stay in flight)
    {
      const int issued = min(nko - 1, kt + NSTG - 2);                 // youngest k-tile in flight
      const int young = issued - kt;                                 // k-tiles that may stay in flight
      if (NSTG >= 4 && young >= 2) wait_vm<2 * kPPT>();
      else if (NSTG >= 3 && young >= 1) wait_vm<kPPT>();
      else wait_vm<0>();
    }
    // (2) one barrier: everybody's pieces of kt landed AND everybody finished reading the
    //     stage that the DMA below overwrites (it was consumed in iteration kt-1)
    __builtin_amdgcn_s_barrier();
    const char* sa = smem + st_cur * kStage;
    const char* sb = sa + kATile;
    // Software-pipelined over the 2 * TK/32 blocks of 16 MFMAs (block = one 32-deep k-slice x one 64-row half of the
    // wave's 128 rows): the fragment reads of block b+1 are issued BEFORE the MFMAs of block b, into the other half of
    // a double register buffer, so their LDS latency runs under 16 MFMAs instead of behind them.
    if (NB > 0 && kPipe) {
#pragma unroll
      for (int u = 0; u < 4; ++u) bfr[0][u] = frag<E, B_KMAJOR, TN, TK>(sb, wn * 64 + u * 16, 0, g, li);
#pragma unroll
      for (int t = 0; t < 4; ++t) af[0][t] = frag<E, A_KMAJOR, TM, TK>(sa, wm * WROWS + t * 16, 0, g, li);
    }
    // the next k-tile's DMA is issued behind the first fragment reads: its address arithmetic runs under their latency
    __builtin_amdgcn_sched_barrier(0);
    if (kt + NSTG - 1 < nko) {
      const int k0 = kbeg + (kt + NSTG - 1) * TK;
      if (A_CONV && A_KMAJOR) cv.dma(p, k0, smem + st_nxt * kStage, wid);
      else if (A_CONV) cvm.dma(p, smem + st_nxt * kStage, wid);
      else dma_tile<A_KMAJOR, TM, TK, NW>(p.A, p.lda, m0, p.M, k0, smem + st_nxt * kStage, wid, lane);
      dma_tile<B_KMAJOR, TN, TK, NW, kKLim>(p.B, p.ldb, n0, p.N, k0, smem + st_nxt * kStage + kATile, wid, lane, p.K);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int kk = b / NTH, th = b % NTH;
      if (!kPipe) {
        if (th == 0) {
#pragma unroll
          for (int u = 0; u < 4; ++u) bfr[0][u] = frag<E, B_KMAJOR, TN, TK>(sb, wn * 64 + u * 16, kk, g, li);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) af[0][t] = frag<E, A_KMAJOR, TM, TK>(sa, wm * WROWS + (th * 4 + t) * 16, kk, g, li);
      }
      if (kPipe && b + 1 < NB) {
        const int kk1 = (b + 1) / NTH, th1 = (b + 1) % NTH;
        if (th1 == 0) {
#pragma unroll
          for (int u = 0; u < 4; ++u) bfr[kk1 & 1][u] = frag<E, B_KMAJOR, TN, TK>(sb, wn * 64 + u * 16, kk1, g, li);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
          af[(b + 1) & 1][t] = frag<E, A_KMAJOR, TM, TK>(sa, wm * WROWS + (th1 * 4 + t) * 16, kk1, g, li);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[u][th * 4 + t] = Elem16<E>::mma(bfr[kPipe ? (kk & 1) : 0][u], af[kPipe ? (b & 1) : 0][t], acc[u][th * 4 + t]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (kCanColsum && do_cs && th == NTH - 1) {
#pragma unroll
        for (int tt = 0; tt < CS; ++tt)
          csum[tt] = Elem16<E>::mma(ones, frag<E, A_KMAJOR, TM, TK>(sa, wm * WROWS + (CS * wn + tt) * 16, kk, g, li),
                                    csum[tt]);
      }
    }
    st_cur = st_cur + 1 == NSTG ? 0 : st_cur + 1;
    st_nxt = st_nxt + 1 == NSTG ? 0 : st_nxt + 1;
  }
  if (!C::PP) __builtin_amdgcn_s_barrier();    // all LDS reads done before the staging overlay
  if (kCanColsum && do_cs && g == 0) {         // every row of the ones-product is the column sum: take row 0
#pragma unroll
    for (int tt = 0; tt < CS; ++tt) {
      const int m = m0 + wm * WROWS + (CS * wn + tt) * 16 + li;
      if (m < p.M) p.colsum_slab[(int64_t)zsl * p.M + m] = csum[tt][0];
    }
  }

  // ---- epilogue: per-wave staging region, 4 passes of 32 rows x 64 cols.
  // vmcnt counts loads and stores together in issue order, so a load issued after a
  // store cannot be consumed before that store has retired.  Hence: the bias is
  // loaded once, and the residual / aux rows of pass ps+1 are requested BEFORE the
  // stores of pass ps (the compiler then waits with a counted vmcnt, not vmcnt(0)).
  float* es = reinterpret_cast<float*>(smem + wid * kEpiBytes);
  const int wrow0 = m0 + wm * WROWS, wcol0 = n0 + wn * 64;
  const int c = (lane & 7) << 3;
  const int n = wcol0 + c;
  const bool n_ok = n < p.N;
  constexpr bool kNeedLd = OUT == OUT_BF16 && (EPI == DVT_EPI_RESIDUAL || EPI == DVT_EPI_DGELU || EPI == DVT_EPI_DRELU);
  const E* ldp = EPI == DVT_EPI_RESIDUAL ? (const E*)p.residual : (const E*)p.aux;
  const int64_t ldl = EPI == DVT_EPI_RESIDUAL ? p.ldr : p.ldaux;
  float bias[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (OUT != OUT_SLAB && p.bias && n_ok) load8<float>(p.bias + n, bias);
  // Consume the bias registers once, here: the loads sit behind a branch, so the wait-count pass would otherwise keep them
  // "possibly pending" through every row-bounds branch below and put vmcnt(0) -- which also drains every store issued so
  // far -- in front of each group of stores.
#pragma unroll
  for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(bias[k]));
  // all sixteen residual / aux rows of the wave's four passes are requested up front (the fragment registers are dead
  // here): one memory round trip for the whole epilogue instead of one per pass, and every load precedes every store
  constexpr int NPS = (WROWS + 31) / 32;               // passes of 32 rows (the last one half empty when WROWS = 112)
  // scattered output rows (implicit convolution only: the parity classes of a strided data gradient, p.orow)
  constexpr bool kCanMap = A_CONV && A_KMAJOR && OUT == OUT_BF16;
  const bool mapped = kCanMap && p.orow != nullptr;
  int orow[kCanMap ? NPS : 1][4];
  if (kCanMap && mapped) {
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int m = wrow0 + ps * 32 + (lane >> 3) + 8 * j;
        m = m < p.M ? m : p.M - 1;
        orow[kCanMap ? ps : 0][j] = p.orow[m];
      }
  }
  V8 rs[kNeedLd ? NPS : 1][4];
  if (kNeedLd) {
    // (addresses as for the stores below: the lane's first row once, then wave-uniform offsets; a row past M reads the last
    //  row instead -- loaded unconditionally, never stored)
    const int nn = n_ok ? n : 0;
    const int rl = wrow0 + (lane >> 3);
    const E* const lfirst = ldp + (int64_t)(rl < p.M ? rl : p.M - 1) * ldl + nn;
    const E* const llast = ldp + (int64_t)(p.M - 1) * ldl + nn;
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const E* src = rl + ps * 32 + 8 * j < p.M ? lfirst + (int64_t)(ps * 32 + 8 * j) * ldl : llast;
        if (kCanMap && mapped && !p.res_compact) src = ldp + (int64_t)orow[kCanMap ? ps : 0][j] * ldl + nn;
        rs[ps][j] = *reinterpret_cast<const V8*>(src);
      }
  }
  // Output addresses: the lane's first row once per tile (one 64-bit multiply), every further row a wave-uniform offset
  // (a constant times the leading dimension: scalar unit).  Per row -- m * ldc with a per-lane m -- it was two v_mul_lo_u32
  // and a v_mad_u64_u32 (quarter-rate) per 16-byte store, twice that where the pre-activation is stored too: a sixth of the
  // vector work of the GELU epilogue, which is not hidden under anything.
  const int r0 = wrow0 + (lane >> 3);
  const int64_t cofs = (int64_t)r0 * p.ldc + n, aofs = (int64_t)r0 * p.ldaux + n;
  // fused BatchNorm statistics (convolution outputs): per-lane sums of its 8 columns over its rows of the tile.
  // {sum y, sum y^2} of the STORED output as the BatchNorm behind the convolution needs them.
  constexpr bool kCanBn = OUT == OUT_BF16 && EPI == DVT_EPI_NONE;
  const bool do_bn = kCanBn && p.bn_partial != nullptr;
  float bsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bsq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int ps = 0; ps < NPS; ++ps) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (ps * 2 + tt < MT)
          *reinterpret_cast<f32x4*>(es + (tt * 16 + li) * kEpiStride + u * 16 + 4 * g) = acc[u][ps * 2 + tt] * p.alpha;
    wave_lds_fence();
    float v[4][8];
    V8 cur[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (lane >> 3) + 8 * j;
      const f32x4 a = *reinterpret_cast<const f32x4*>(es + row * kEpiStride + c);
      const f32x4 b = *reinterpret_cast<const f32x4*>(es + row * kEpiStride + c + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[j][k] = a[k]; v[j][4 + k] = b[k]; }
      cur[j] = kNeedLd ? rs[kNeedLd ? ps : 0][j] : V8{0, 0, 0, 0, 0, 0, 0, 0};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = wrow0 + ps * 32 + (lane >> 3) + 8 * j;
      const bool mine = WROWS % 32 == 0 || ps * 32 + (lane >> 3) + 8 * j < WROWS;   // rows past the wave's 112 are the next wave row's
      if (mine && m < p.M && n_ok) {
        if (OUT == OUT_SLAB) {
          // (re-read from cache by the reduce: streaming stores cost 10 %.  Stored through an explicit global pointer: the
          //  compiler had lost p.slab's address space and emitted flat_store, which also counts on lgkmcnt)
          typedef __attribute__((address_space(1))) f32x4 gf4;
          gf4* o = (gf4*)(p.slab + ((int64_t)zsl * p.M + r0) * p.N + n + (int64_t)(ps * 32 + 8 * j) * p.N);
          o[0] = f32x4{v[j][0], v[j][1], v[j][2], v[j][3]};
          o[1] = f32x4{v[j][4], v[j][5], v[j][6], v[j][7]};
        } else if (OUT == OUT_F32) {
          float* o = (float*)p.C + cofs + (int64_t)(ps * 32 + 8 * j) * p.ldc;
#pragma unroll
          for (int k = 0; k < 8; ++k) v[j][k] += bias[k];
          if (p.accumulate) {
            float old[8];
            load8<float>(o, old);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[j][k] += old[k];
          }
          store8<float>(o, v[j]);
        } else {
          float pre[8];
          // (the GELU / GELU' arithmetic as packed fp32 pairs -- v_pk_fma_f32, half the vector instructions, bit-identical --
          //  was built and measured in round 6: 165.3 -> 165.9 us per FF1 launch in bf16, 637 -> 655 us at the long-clip fp16
          //  shape; the epilogue is bound by its stores, not its arithmetic: profiles/r06_gemm_epilogue.md)
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float ld = (float)cur[j][k];
            v[j][k] = epi_apply(EPI, v[j][k], bias[k], ld, ld, pre[k]);
          }
          if (kCanBn && do_bn) {                     // of the values as STORED (rounded to E): what the BatchNorm behind the
#pragma unroll                                        // layer normalises, and what every other route sums (halo / streamed-weight
            for (int k = 0; k < 8; ++k) {            // kernels, the split-K reduce, dvt_bn_stats on z): one convention
              const float f = (float)(E)v[j][k];
              bsum[k] += f;
              bsq[k] = fmaf(f, f, bsq[k]);
            }
          }
          if (EPI == DVT_EPI_GELU && p.aux) DVT_C_STORE((E*)p.aux + aofs + (int64_t)(ps * 32 + 8 * j) * p.ldaux, pre);
          if (kCanMap && mapped) {
            DVT_C_STORE((E*)p.C + (int64_t)orow[kCanMap ? ps : 0][j] * p.ldc + n, v[j]);
          } else {
            DVT_C_STORE((E*)p.C + cofs + (int64_t)(ps * 32 + 8 * j) * p.ldc, v[j]);
          }
        }
      }
    }
    wave_lds_fence();
  }
  if (kCanBn && do_bn) {
    // lanes with equal (lane & 7) hold the same 8 columns for different rows: fixed-order tree over the 8 row lanes
#pragma unroll
    for (int o = 8; o < 64; o <<= 1)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        bsum[k] += __shfl_xor(bsum[k], o, 64);
        bsq[k] += __shfl_xor(bsq[k], o, 64);
      }
    if ((lane >> 3) == 0 && n_ok) {
      const int part = (tile / p.tiles_n) * (NW / WN) + wm;
      store8<float>(p.bn_partial + ((int64_t)part * 2 + 0) * p.N + n, bsum);
      store8<float>(p.bn_partial + ((int64_t)part * 2 + 1) * p.N + n, bsq);
    }
  }
}


// ---------------------------------------------------------------- host side
// dynamic LDS of a configuration: the stage ring, or the epilogue staging that overlays it
template <int CFG> constexpr int smem_bytes() {
  typedef Cfg<CFG> C;
  constexpr int stages = C::NSTG * (C::TM + C::TN) * C::TK * 2;
  constexpr int epi = C::NW * kEpiBytes;
  static_assert((stages >= epi ? stages : epi) <= 160 * 1024, "LDS budget");
  return stages >= epi ? stages : epi;
}

template <typename E, bool AK, bool BK, int CFG, int EPI, int OUT>
int launch_one(const GemmParams& p, dim3 grid, dim3 block, int smem_bytes, hipStream_t st) {
  static DvtLdsAttr attr_set;
    dvt_lds_attr(attr_set, (const void*)gemm_dma_kernel<E, AK, BK, CFG, EPI, OUT>, smem_bytes);
  hipLaunchKernelGGL((gemm_dma_kernel<E, AK, BK, CFG, EPI, OUT>), grid, block, smem_bytes, st, p);
  DVT_LAUNCH_CHECK("dvt_gemm(dma)");
  return DVT_OK;
}

template <typename E, int CFG>
int launch_cfg(const GemmParams& pin, bool ak, bool bk, int split, hipStream_t st) {
  typedef Cfg<CFG> C;
  constexpr int kSmem = smem_bytes<CFG>();
  GemmParams p = pin;
  // Outputs that the next kernel can still find in the 256 MB Infinity Cache stay cacheable; larger ones (>= 180 MB) are
  // streamed so that they do not evict the operand panels from L2
  p.stream_out = (int64_t)p.M * p.N * 2 * (p.epilogue == DVT_EPI_GELU && p.aux ? 2 : 1) >= (int64_t)180 * 1000000;
  const int tiles_m = (int)dvt_cdiv(p.M, C::TM);
  p.tiles_n = (int)dvt_cdiv(p.N, C::TN);
  if (split != 1) p.pig_blocks = 0;        // (the caller only hands a carried reduce to one-slice launches)
  const dim3 grid((unsigned)(tiles_m * p.tiles_n + p.pig_blocks), 1, (unsigned)split), block(C::NW * 64);
  const int e = p.epilogue;
  if (p.slab) {
    if (!ak && !bk) return launch_one<E, false, false, CFG, DVT_EPI_NONE, OUT_SLAB>(p, grid, block, kSmem, st);
    if (ak && bk) return launch_one<E, true, true, CFG, DVT_EPI_NONE, OUT_SLAB>(p, grid, block, kSmem, st);
    if (ak && !bk) return launch_one<E, true, false, CFG, DVT_EPI_NONE, OUT_SLAB>(p, grid, block, kSmem, st);
  } else if (p.out_f32) {
    if (!ak && !bk && e == DVT_EPI_NONE) return launch_one<E, false, false, CFG, DVT_EPI_NONE, OUT_F32>(p, grid, block, kSmem, st);
  } else if (ak && bk) {
    if (e == DVT_EPI_NONE) return launch_one<E, true, true, CFG, DVT_EPI_NONE, OUT_BF16>(p, grid, block, kSmem, st);
    if (e == DVT_EPI_GELU) return launch_one<E, true, true, CFG, DVT_EPI_GELU, OUT_BF16>(p, grid, block, kSmem, st);
    if (e == DVT_EPI_RELU) return launch_one<E, true, true, CFG, DVT_EPI_RELU, OUT_BF16>(p, grid, block, kSmem, st);
    if (e == DVT_EPI_RESIDUAL) return launch_one<E, true, true, CFG, DVT_EPI_RESIDUAL, OUT_BF16>(p, grid, block, kSmem, st);
  } else if (ak && !bk) {
    if (e == DVT_EPI_NONE) return launch_one<E, true, false, CFG, DVT_EPI_NONE, OUT_BF16>(p, grid, block, kSmem, st);
    if (e == DVT_EPI_DGELU) return launch_one<E, true, false, CFG, DVT_EPI_DGELU, OUT_BF16>(p, grid, block, kSmem, st);
    if (e == DVT_EPI_DRELU) return launch_one<E, true, false, CFG, DVT_EPI_DRELU, OUT_BF16>(p, grid, block, kSmem, st);
  }
  return 1;   // combination not instantiated: caller falls back to the 128x128 kernel
}

template <typename E, int CFG>
int launch_conv(const GemmParams& pin, hipStream_t st) {
  typedef Cfg<CFG> C;
  constexpr int kSmem = smem_bytes<CFG>();
  GemmParams p = pin;
  const int tiles_m = (int)dvt_cdiv(p.M, C::TM);
  p.tiles_n = (int)dvt_cdiv(p.N, C::TN);
  const dim3 grid((unsigned)(tiles_m * p.tiles_n + p.pig_blocks), 1, 1), block(C::NW * 64);   // + a carried split-K reduce
  if (p.epilogue == DVT_EPI_RESIDUAL) {             // data gradient joined by the shortcut's gradient (dvt_conv_desc.residual)
    static DvtLdsAttr attr_set_r;
    dvt_lds_attr(attr_set_r, (const void*)gemm_dma_kernel<E, true, true, CFG, DVT_EPI_RESIDUAL, OUT_BF16, true>, kSmem);
    hipLaunchKernelGGL((gemm_dma_kernel<E, true, true, CFG, DVT_EPI_RESIDUAL, OUT_BF16, true>), grid, block, kSmem, st, p);
    DVT_LAUNCH_CHECK("dvt_conv2d_implicit(dma, residual)");
    return DVT_OK;
  }
  static DvtLdsAttr attr_set;
    dvt_lds_attr(attr_set, (const void*)gemm_dma_kernel<E, true, true, CFG, DVT_EPI_NONE, OUT_BF16, true>, kSmem);
  hipLaunchKernelGGL((gemm_dma_kernel<E, true, true, CFG, DVT_EPI_NONE, OUT_BF16, true>), grid, block, kSmem, st, p);
  DVT_LAUNCH_CHECK("dvt_conv2d_implicit(dma)");
  return DVT_OK;
}

template <typename E, int CFG>
int launch_conv_wgrad(const GemmParams& pin, int split, hipStream_t st) {
  typedef Cfg<CFG> C;
  constexpr int kSmem = smem_bytes<CFG>();
  GemmParams p = pin;
  p.pig_blocks = 0;
  const int tiles_m = (int)dvt_cdiv(p.M, C::TM);
  p.tiles_n = (int)dvt_cdiv(p.N, C::TN);
  const dim3 grid((unsigned)(tiles_m * p.tiles_n), 1, (unsigned)split), block(C::NW * 64);
  static DvtLdsAttr attr_set;
    dvt_lds_attr(attr_set, (const void*)gemm_dma_kernel<E, false, false, CFG, DVT_EPI_NONE, OUT_SLAB, true>, kSmem);
  hipLaunchKernelGGL((gemm_dma_kernel<E, false, false, CFG, DVT_EPI_NONE, OUT_SLAB, true>), grid, block, kSmem, st, p);
  DVT_LAUNCH_CHECK("dvt_conv2d_implicit_wgrad(dma)");
  return DVT_OK;
}

}  // namespace

#if DVT_GEMM256_UNIT == 1
// configuration 6 (256x64x64 convolution tiles), kept out of the main module like configuration 5
int dvt_conv_dma_launch_c6(const GemmParams& p, int cfg, hipStream_t st) {
  if (cfg == 9) return p.elem == DVT_F16 ? launch_conv<f16, 9>(p, st) : launch_conv<bf16, 9>(p, st);
  if (cfg == 10) return p.elem == DVT_F16 ? launch_conv<f16, 10>(p, st) : launch_conv<bf16, 10>(p, st);
  if (cfg == 7) return p.elem == DVT_F16 ? launch_conv<f16, 7>(p, st) : launch_conv<bf16, 7>(p, st);
  return p.elem == DVT_F16 ? launch_conv<f16, 6>(p, st) : launch_conv<bf16, 6>(p, st);
}
// forward / data gradient with the reduction split over blockIdx.z into fp32 slabs (few output rows x deep K: R(2+1)D layers
// 3 - 4): 128 x 128 tiles, two workgroups per CU; the caller's reduce sums the slabs (gemm.hip: conv_split_reduce_kernel)
template <typename E, int CFG>
static int launch_conv_split(const GemmParams& pin, int split, hipStream_t st) {
  typedef Cfg<CFG> C;
  constexpr int kSmem = smem_bytes<CFG>();
  GemmParams p = pin;
  p.pig_blocks = 0;
  p.tiles_n = (int)dvt_cdiv(p.N, C::TN);
  const dim3 grid((unsigned)(dvt_cdiv(p.M, C::TM) * p.tiles_n), 1, (unsigned)split), block(C::NW * 64);
  static DvtLdsAttr attr_set;
  dvt_lds_attr(attr_set, (const void*)gemm_dma_kernel<E, true, true, CFG, DVT_EPI_NONE, OUT_SLAB, true>, kSmem);
  hipLaunchKernelGGL((gemm_dma_kernel<E, true, true, CFG, DVT_EPI_NONE, OUT_SLAB, true>), grid, block, kSmem, st, p);
  DVT_LAUNCH_CHECK("dvt_conv2d_implicit(dma, split K)");
  return DVT_OK;
}
// (at most one workgroup per CU in all: the four-deep ring of configuration 10; else two 64 KiB workgroups per CU)
int dvt_conv_dma_launch_split(const GemmParams& p, int split, hipStream_t st) {
  const bool one = dvt_cdiv(p.M, 128) * dvt_cdiv(p.N, 128) * split <= dvt_num_cus();
  if (p.elem == DVT_F16) return one ? launch_conv_split<f16, 10>(p, split, st) : launch_conv_split<f16, 9>(p, split, st);
  return one ? launch_conv_split<bf16, 10>(p, split, st) : launch_conv_split<bf16, 9>(p, split, st);
}
int dvt_conv_wgrad_dma_launch_c6(const GemmParams& p, int split, int cfg, hipStream_t st) {
  if (cfg == 7) return p.elem == DVT_F16 ? launch_conv_wgrad<f16, 7>(p, split, st) : launch_conv_wgrad<bf16, 7>(p, split, st);
  return p.elem == DVT_F16 ? launch_conv_wgrad<f16, 6>(p, split, st) : launch_conv_wgrad<bf16, 6>(p, split, st);
}
// configuration 8 (224-row tiles of the antiphase loop): A k-major only, the epilogues N = 512 launches use
template <typename E>
int launch_224(const GemmParams& pin, bool bk, hipStream_t st) {
  typedef Cfg<8> C;
  constexpr int kSmem = smem_bytes<8>();
  GemmParams p = pin;
  p.stream_out = (int64_t)p.M * p.N * 2 >= (int64_t)180 * 1000000;
  p.tiles_n = (int)dvt_cdiv(p.N, C::TN);
  const dim3 grid((unsigned)(dvt_cdiv(p.M, C::TM) * p.tiles_n + p.pig_blocks), 1, 1), block(C::NW * 64);
  if (p.slab || p.out_f32) return 1;
  if (bk && p.epilogue == DVT_EPI_NONE) return launch_one<E, true, true, 8, DVT_EPI_NONE, OUT_BF16>(p, grid, block, kSmem, st);
  if (bk && p.epilogue == DVT_EPI_RESIDUAL) return launch_one<E, true, true, 8, DVT_EPI_RESIDUAL, OUT_BF16>(p, grid, block, kSmem, st);
  if (!bk && p.epilogue == DVT_EPI_NONE) return launch_one<E, true, false, 8, DVT_EPI_NONE, OUT_BF16>(p, grid, block, kSmem, st);
  return 1;
}
int dvt_gemm_dma_launch_224(const GemmParams& p, bool b_kmajor, hipStream_t st) {
  return p.elem == DVT_F16 ? launch_224<f16>(p, b_kmajor, st) : launch_224<bf16>(p, b_kmajor, st);
}
// configuration 5 (antiphase main loop), compiled as a module of its own
int dvt_gemm_dma_launch_pp(const GemmParams& p, bool a_kmajor, bool b_kmajor, int split, hipStream_t st) {
  if (p.elem == DVT_F16) return launch_cfg<f16, 5>(p, a_kmajor, b_kmajor, split, st);
  return launch_cfg<bf16, 5>(p, a_kmajor, b_kmajor, split, st);
}
#else
// Weight gradient with the column matrix gathered on the fly: slab[z][M = kh*kw*C][N = Cout] partial sums.
int dvt_conv_wgrad_dma_launch(const GemmParams& p, int split, int cfg, hipStream_t st) {
  if (cfg == 6 || cfg == 7) return dvt_conv_wgrad_dma_launch_c6(p, split, cfg, st);
  if (p.elem == DVT_F16)
    return cfg == 0 ? launch_conv_wgrad<f16, 0>(p, split, st) : cfg == 4 ? launch_conv_wgrad<f16, 4>(p, split, st) : launch_conv_wgrad<f16, 1>(p, split, st);
  return cfg == 0 ? launch_conv_wgrad<bf16, 0>(p, split, st) : cfg == 4 ? launch_conv_wgrad<bf16, 4>(p, split, st) : launch_conv_wgrad<bf16, 1>(p, split, st);
}

// Implicit-GEMM convolution forward / data gradient: C[M = N*Ho*Wo, Cout] = gather(x) * Wp^T with the gather
// fused into the A-operand DMA.  cfg 0 = 256x256x64 (Cout > 128), cfg 1 = 256x128x32, cfg 4 = 256x64x32 (Cout <= 64).
int dvt_conv_dma_launch(const GemmParams& p, int cfg, hipStream_t st) {
  if (cfg == 6 || cfg == 7 || cfg == 9 || cfg == 10) return dvt_conv_dma_launch_c6(p, cfg, st);
  if (p.elem == DVT_F16) return cfg == 0 ? launch_conv<f16, 0>(p, st) : cfg == 4 ? launch_conv<f16, 4>(p, st) : launch_conv<f16, 1>(p, st);
  return cfg == 0 ? launch_conv<bf16, 0>(p, st) : cfg == 4 ? launch_conv<bf16, 4>(p, st) : launch_conv<bf16, 1>(p, st);
}

// Returns DVT_OK, a negative dvt_status, or 1 when this (layout, epilogue, output)
// combination has no LDS-DMA instantiation.
int dvt_gemm_dma_launch(const GemmParams& p, bool a_kmajor, bool b_kmajor, int split, int cfg, hipStream_t st) {
  if (cfg == 5) return dvt_gemm_dma_launch_pp(p, a_kmajor, b_kmajor, split, st);
  if (cfg == 8) {
    const int rc = a_kmajor && split == 1 ? dvt_gemm_dma_launch_224(p, b_kmajor, st) : 1;
    return rc == 1 ? dvt_gemm_dma_launch_pp(p, a_kmajor, b_kmajor, split, st) : rc;   // no 224-row instantiation: 256 rows
  }
  if (p.elem == DVT_F16)
    return cfg == 3 ? launch_cfg<f16, 3>(p, a_kmajor, b_kmajor, split, st) : launch_cfg<f16, 0>(p, a_kmajor, b_kmajor, split, st);
  if (cfg == 3) return launch_cfg<bf16, 3>(p, a_kmajor, b_kmajor, split, st);
  return cfg == 0 ? launch_cfg<bf16, 0>(p, a_kmajor, b_kmajor, split, st)
                  : launch_cfg<bf16, 1>(p, a_kmajor, b_kmajor, split, st);
}
#endif  // DVT_GEMM256_UNIT
