// conv3x1_wgrad.hip -- weight gradient of the temporal half of R(2+1)D-18's layer-1 Conv2Plus1D (video_resnet.py:30-31: a
// (3, 1, 1) convolution, 144 mid planes -> 64 planes, stride 1, pad 1) from LDS-resident sliding windows.
//
//   dW[co][ci][kt] = sum over clips n, frames t and pixels p of dz[n, t, p, co] * x[n, t + kt - 1, p, ci]
//
// The implicit form (gemm256.hip, mn-major operands) gathers x once per temporal tap through the CU's vector-memory path:
// 1,220 MB fetched per launch for 438 MB of operands (profiles/r04_frametransformer_pmc_traffic.md, traffic ratio 2.37),
// 205 us at 28 clips of 12 x 56^2.  Here a workgroup owns a SEGMENT of S pixels of one clip over ALL its T frames: it stages
// the (T + 2) x S x 144 window of x (frames -1 and T are zero rows) and the T x S x 64 tile of dz ONCE each, and the three
// taps are three constant position offsets (0, S, 2 S) into the same window -- conv3x3_wgrad.hip's scheme with a halo in the
// frame direction only, so no input element is fetched twice, not even by a neighbouring tile.
//
//   * dz image: [position][64 channels], 128-byte rows, the 32-byte-unit swizzle of conv3x3_wgrad.hip.
//   * x image: [position][144 channels] in rows of TEN 32-byte units (320 bytes): nine channel blocks of 16 plus one unit of
//     padding, block cb of position k stored at unit cb + ((k >> 3) & 1).  A 32-lane half of ds_read_b64_tr_b16 touches the
//     positions {q, 8 + q} + 4 hf: 320-byte rows put positions q = 0 .. 3 on banks 0 / 16 / 32 / 48 (+ 8 each), the shift by
//     one unit moves positions 8 + q to banks 8 / 24 / 40 / 56 -- the eight 8-bank groups of one LDS cycle, conflict-free.
//     (No linear row pitch does that: positions k and k + 8 are 8 rows apart and 8 x pitch is a multiple of 256 bytes for
//     every pitch that is a multiple of 32.)  The DMA is lane-linear, so the layout lives on the per-lane source address:
//     slot = 16 bytes, 20 slots per position, two of them zero padding.
//   * Output [co 64][tap * 144 + ci] = 4 x 27 MFMA blocks; wave w owns column blocks w, w + 8, w + 16 (and w + 24 for
//     w < 3) for all four co blocks: 12 - 16 MFMAs per 32 positions from 4 + 3 (4) fragment reads; accumulators in registers
//     over the workgroup's whole tile sequence; next tile's images stream into second buffers under the current tile's MFMAs.
//   * Persistent grid; ONE fp32 partial [432][64] per workgroup, summed and scattered into the parameter's own
//     [co][ci][kt] layout by the family's split-K reduce (dvt_splitk_pending with conv_taps = 3, conv_cin = 144).
#include "conv3x1_window.h"

namespace {

using namespace dvt_window;
constexpr int kM = 3 * kCI;                  // 432 slab rows: tap * 144 + ci
constexpr int kNB = 3 * kCB;                 // 27 column blocks
constexpr int kMaxZP = 2;                    // dz DMA pieces (1 KiB) per wave: tile <= 16 KiB

struct TwParams {
  const void* x;        // [N, T, L, 144]: the normalised mid activation, or (aff.mean != nullptr) the convolution output z in front of it
  const void* dz;       // [N, T, L, 64]
  float* slab;          // [grid][432][64]
  Window w;
  Affine aff;
  int ntiles, z_bytes;
};

__device__ __forceinline__ int tw_swz(int k) { return ((k >> 1) & 1) | (((k >> 3) & 1) << 1); }   // conv3x3_wgrad.hip's cw_swz

template <int N> struct IntC { static constexpr int value = N; };

template <typename E>
__global__ __launch_bounds__(kNW * 64) void conv3x1_wgrad_kernel(const TwParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using V8 = typename Elem16<E>::v8;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const Window& w = p.w;
  const int S = w.S;
  // two windows, then two dz tiles, addressed as smem + byte offset (a pointer picked at run time by it & 1 turned every
  // fragment address into a 64-bit generic pointer: add, null check, select and cast per transposing read)
  const int zbase = 2 * w.x_bytes;
  const E* xg = (const E*)p.x;
  const E* zg = (const E*)p.dz;
  const int zp = p.z_bytes >> 10;
  const bool affine = p.aff.mean != nullptr;                                       // (kernel-uniform)
  AffineRegs st{};
  if (affine) window_affine_regs(p.aff, st);

  // ---- per-lane coordinates of this wave's DMA pieces (fixed for the launch); dz: frame << 20 | pixel << 8 | source chunk << 1 | valid
  unsigned xq[kMaxXP];
  int zq[kMaxZP];
  window_coords(w, wid, lane, xq);
  const int c16 = lane & 7;
#pragma unroll
  for (int i = 0; i < kMaxZP; ++i) {
    const int piece = wid + kNW * i;
    const int pos = (piece * 64 + lane) >> 3;
    const int t = pos / S, sx = pos - t * S;
    const int ok = (piece < zp && pos < w.KP) ? 1 : 0;
    zq[i] = (t << 20) | (sx << 8) | (((((c16 >> 1) ^ tw_swz(pos)) << 1) | (c16 & 1)) << 1) | ok;
  }
  auto load_tile = [&](int tile, int b) {
    const int n = tile / w.segs, sg = tile - n * w.segs;
    const int64_t pix0 = (int64_t)n * w.T * w.L + (int64_t)sg * S;          // pixel (frame 0, first pixel of the segment)
    window_load<E>(w, xg, pix0, xq, wid, smem + b * w.x_bytes);
#pragma unroll
    for (int i = 0; i < kMaxZP; ++i) {
      const int piece = wid + kNW * i;
      if (piece < zp) {
        const E* src = (zq[i] & 1) ? zg + (pix0 + (int64_t)(zq[i] >> 20) * w.L + ((zq[i] >> 8) & 0xFFF)) * kCO + ((zq[i] >> 1) & 7) * 8
                                   : reinterpret_cast<const E*>(window_zero16);
        dvt_dma16(src, smem + zbase + b * p.z_bytes + piece * 1024);
      }
    }
  };

  // this wave's column blocks nb = wid + 8 j: tap = nb / 9 (position offset tap * S), ci block = nb % 9.  A k-step adds 32
  // positions, which changes neither bit 1 nor bit 3 of a position: the swizzle terms are fixed for the whole launch.
  const int cnt = wid < kNB - 3 * kNW ? 4 : 3;            // wave-uniform (waves 0 .. 2: four blocks)
  int zo[4][2], xo[4][2];
  {
    const int q = li >> 2, pp = li & 3;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int k = 8 * g + 4 * hf + q;
#pragma unroll
      for (int m = 0; m < 4; ++m) zo[m][hf] = k * 128 + ((m ^ tw_swz(k)) << 5) + 8 * pp;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nb = min(wid + kNW * j, kNB - 1), tap = nb / kCB, cb = nb - tap * kCB;
        const int kx = k + tap * S;
        xo[j][hf] = kx * kXRow + ((cb + ((kx >> 3) & 1)) << 5) + 8 * pp;
      }
    }
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int tile = blockIdx.x;
  if (tile < p.ntiles) load_tile(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (affine) {                                  // the first window: z -> relu(z * s + t) in place
    window_transform<E>(w, smem, st, p.aff.relu);
    __syncthreads();
  }

  const int nks = w.KP >> 5;
  // the tile sequence with the wave's count of column blocks as a compile-time constant (instantiated for 4 and 3, picked per
  // wave; both forms pass the same barriers) -- as a run-time test it put a branch around every row's fourth MFMA
  auto run = [&](auto CNT) {
  constexpr int NJ = decltype(CNT)::value;
  for (int it = 0; tile < p.ntiles; ++it, tile += gridDim.x) {
    const char* cx = smem + (it & 1) * w.x_bytes;
    const char* cz = smem + zbase + (it & 1) * p.z_bytes;
    if (tile + (int)gridDim.x < p.ntiles) load_tile(tile + gridDim.x, (it + 1) & 1);
    V8 zf[2][4], xf[2][4];
    auto rd = [&](int ks, V8* zv, V8* xv) {
      const char* bz = cz + ks * (32 * 128);
      const char* bx = cx + ks * (32 * kXRow);
#pragma unroll
      for (int m = 0; m < 4; ++m)
        zv[m] = __builtin_shufflevector(Elem16<E>::tr_read(bz + zo[m][0]), Elem16<E>::tr_read(bz + zo[m][1]), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
          xv[j] = __builtin_shufflevector(Elem16<E>::tr_read(bx + xo[j][0]), Elem16<E>::tr_read(bx + xo[j][1]), 0, 1, 2, 3, 4, 5, 6, 7);
    };
    rd(0, zf[0], xf[0]);
    for (int ks = 0; ks < nks; ks += 2) {        // two steps per trip: the fragment buffers alternate without indexing
      if (ks + 1 < nks) rd(ks + 1, zf[1], xf[1]);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[m][j] = Elem16<E>::mma(zf[0][m], xf[0][j], acc[m][j]);
      if (ks + 1 < nks) {
        if (ks + 2 < nks) rd(ks + 2, zf[0], xf[0]);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[m][j] = Elem16<E>::mma(zf[1][m], xf[1][j], acc[m][j]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the next tile's images have landed
    __builtin_amdgcn_s_barrier();                                  // and everybody is done with this tile's
    if (affine && tile + (int)gridDim.x < p.ntiles) {
      window_transform<E>(w, smem + ((it + 1) & 1) * w.x_bytes, st, p.aff.relu);
      __syncthreads();
    }
  }
  };
  if (cnt == 4) run(IntC<4>{});
  else run(IntC<3>{});

  // ---- this workgroup's partial: slab[blockIdx][m = tap * 144 + 16 cb + li][n = 16 mb + 4 g .. + 3]
  // (mma(dz fragment, x fragment): lane (g, li) holds C[co = 16 mb + 4 g + r][ci = 16 cb + li])
  float* out = p.slab + (int64_t)blockIdx.x * kM * kCO;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j >= cnt) break;
    const int nb = wid + kNW * j, tap = nb / kCB, cb = nb - tap * kCB;
    float* row = out + (int64_t)(tap * kCI + cb * 16 + li) * kCO + 4 * g;
#pragma unroll
    for (int m = 0; m < 4; ++m) *reinterpret_cast<f32x4*>(row + 16 * m) = acc[m][j];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same weight gradient with the tile's phases overlapped by construction (round 6; conv3x1_fwd.hip has the forward's
// sibling and the counters behind it): sixteen waves, ONE barrier per tile.  Waves 0 - 7 run tile i's MFMAs from buffer
// i % 3 exactly as above; waves 8 - 15 meanwhile request window + gradient tile i + 2 into the buffer tile i - 1 has left and
// apply the virtual BatchNorm to window i + 1, which landed before the barrier that opened the interval.  Three (window,
// gradient tile) pairs: 3 x (35 + 12) KiB at 12 frames of 8-pixel segments.
template <typename E>
__global__ __launch_bounds__(2 * kNW * 64) void conv3x1_wgrad_pipe_kernel(const TwParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using V8 = typename Elem16<E>::v8;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Window& w = p.w;
  const int S = w.S;
  const int zbase = 3 * w.x_bytes;
  const int n_my = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // tiles of this workgroup (>= 1)

  if (wid >= kNW) {
    // ================================================================ helper waves
    const int hw = wid - kNW, htid = threadIdx.x - kNW * 64;
    const E* xg = (const E*)p.x;
    const E* zg = (const E*)p.dz;
    const int zp = p.z_bytes >> 10;
    const bool affine = p.aff.mean != nullptr;
    AffineRegs st{};
    if (affine) window_affine_regs(p.aff, st, htid);
    unsigned xq[kMaxXP];
    int zq[kMaxZP];
    window_coords(w, hw, lane, xq);
    const int c16 = lane & 7;
#pragma unroll
    for (int i = 0; i < kMaxZP; ++i) {
      const int piece = hw + kNW * i;
      const int pos = (piece * 64 + lane) >> 3;
      const int t = pos / S, sx = pos - t * S;
      const int ok = (piece < zp && pos < w.KP) ? 1 : 0;
      zq[i] = (t << 20) | (sx << 8) | (((((c16 >> 1) ^ tw_swz(pos)) << 1) | (c16 & 1)) << 1) | ok;
    }
    auto load_tile = [&](int j) {
      const int tile = blockIdx.x + j * gridDim.x, b = j % 3;
      const int n = tile / w.segs, sg = tile - n * w.segs;
      const int64_t pix0 = (int64_t)n * w.T * w.L + (int64_t)sg * S;
      window_load<E>(w, xg, pix0, xq, hw, smem + b * w.x_bytes);
#pragma unroll
      for (int i = 0; i < kMaxZP; ++i) {
        const int piece = hw + kNW * i;
        if (piece < zp) {
          const E* src = (zq[i] & 1) ? zg + (pix0 + (int64_t)(zq[i] >> 20) * w.L + ((zq[i] >> 8) & 0xFFF)) * kCO + ((zq[i] >> 1) & 7) * 8
                                     : reinterpret_cast<const E*>(window_zero16);
          dvt_dma16(src, smem + zbase + b * p.z_bytes + piece * 1024);
        }
      }
    };
    load_tile(0);
    if (n_my > 1) load_tile(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                              // P0: pairs 0 and 1 have landed
    if (affine) window_transform<E>(w, smem, st, p.aff.relu);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                              // P1: window 0 is ready
    for (int i = 0; i < n_my; ++i) {
      if (i + 2 < n_my) load_tile(i + 2);                         // into the buffers tile i - 1 has left
      if (affine && i + 1 < n_my) window_transform<E>(w, smem + ((i + 1) % 3) * w.x_bytes, st, p.aff.relu);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // pair i + 2 has landed (needed from the next interval on)
      __syncthreads();                                            // B_{i+1}
    }
    return;
  }

  // ================================================================== compute waves (the MFMA phase of the kernel above)
  const int g = lane >> 4, li = lane & 15;
  const int cnt = wid < kNB - 3 * kNW ? 4 : 3;
  int zo[4][2], xo[4][2];
  {
    const int q = li >> 2, pp = li & 3;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int k = 8 * g + 4 * hf + q;
#pragma unroll
      for (int m = 0; m < 4; ++m) zo[m][hf] = k * 128 + ((m ^ tw_swz(k)) << 5) + 8 * pp;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nb = min(wid + kNW * j, kNB - 1), tap = nb / kCB, cb = nb - tap * kCB;
        const int kx = k + tap * S;
        xo[j][hf] = kx * kXRow + ((cb + ((kx >> 3) & 1)) << 5) + 8 * pp;
      }
    }
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();                                                // P0
  __syncthreads();                                                // P1
  const int nks = w.KP >> 5;
  auto run = [&](auto CNT) {
  constexpr int NJ = decltype(CNT)::value;
  for (int i = 0; i < n_my; ++i) {
    const char* cx = smem + (i % 3) * w.x_bytes;
    const char* cz = smem + zbase + (i % 3) * p.z_bytes;
    // (sixteen waves leave 128 registers per lane: ONE set of fragments -- the second compute wave of the SIMD and the helpers
    //  cover its LDS round trips; two sets spilled 95 registers)
    V8 zf[4], xf[4];
    for (int ks = 0; ks < nks; ++ks) {
      const char* bz = cz + ks * (32 * 128);
      const char* bx = cx + ks * (32 * kXRow);
#pragma unroll
      for (int m = 0; m < 4; ++m)
        zf[m] = __builtin_shufflevector(Elem16<E>::tr_read(bz + zo[m][0]), Elem16<E>::tr_read(bz + zo[m][1]), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        xf[j] = __builtin_shufflevector(Elem16<E>::tr_read(bx + xo[j][0]), Elem16<E>::tr_read(bx + xo[j][1]), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[m][j] = Elem16<E>::mma(zf[m], xf[j], acc[m][j]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                 // B_{i+1}
  }
  };
  if (cnt == 4) run(IntC<4>{});
  else run(IntC<3>{});
  float* out = p.slab + (int64_t)blockIdx.x * kM * kCO;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j >= cnt) break;
    const int nb = wid + kNW * j, tap = nb / kCB, cb = nb - tap * kCB;
    float* row = out + (int64_t)(tap * kCI + cb * 16 + li) * kCO + 4 * g;
#pragma unroll
    for (int m = 0; m < 4; ++m) *reinterpret_cast<f32x4*>(row + 16 * m) = acc[m][j];
  }
}

// the pipelined form's geometry: three (window, gradient tile) pairs; 0 = not taken (the kernel above then)
int twp_plan(int T, int L, Window* q) {
#ifdef DVT_TW_NO_PIPE
  return 0;
#endif
  if (!window_plan(T, L, q, 128, 0, 3)) return 0;
  return ((q->KP * 128) >> 10) <= kNW * kMaxZP;
}

// pixels per segment and the image sizes for T frames of L pixels (two windows + two dz tiles + the affine table in LDS)
int tw_plan_two(int T, int L, Window* q) {
  if (!window_plan(T, L, q, 128, 0, 2)) return 0;
  return ((q->KP * 128) >> 10) <= kNW * kMaxZP;
}

// the launcher's choice: the pipelined form where its buffers fit (*pipe = 1), else the two-buffer kernel
int tw_plan(int T, int L, Window* q, int* pipe = nullptr) {
  int dummy;
  if (!pipe) pipe = &dummy;
  *pipe = twp_plan(T, L, q);
  if (*pipe) return 1;
  return tw_plan_two(T, L, q);
}

int tw_grid(int64_t N, const Window& q) {
  const int64_t ntiles = N * q.segs;
  return (int)(ntiles < dvt_num_cus() ? ntiles : dvt_num_cus());
}

}  // namespace

extern "C" {

int dvt_conv3x1_wgrad_supported(int64_t N, int T, int L, int Cin, int Cout, int dtype) {
  Window q;
  return N > 0 && Cin == kCI && Cout == kCO && dvt_is_16bit(dtype) && tw_plan(T, L, &q) && N * q.segs < ((int64_t)1 << 31) &&
                 N * T * L < ((int64_t)1 << 31) ? 1 : 0;
}

size_t dvt_conv3x1_wgrad_workspace_bytes(int64_t N, int T, int L) {
  Window q;
  if (N <= 0 || !tw_plan(T, L, &q)) return 0;
  return (size_t)tw_grid(N, q) * kM * kCO * sizeof(float);
}

int dvt_conv3x1_wgrad(const void* x, const dvt_bn_affine* x_affine, const void* dz, float* dw, void* workspace, int64_t N, int T,
                      int L, int accumulate, int defer_reduce, dvt_splitk_pending* pending, int dtype, dvt_stream_t stream) {
  DVT_REQUIRE(x && dz && dw && workspace && N > 0 && T > 0 && L > 0, "dvt_conv3x1_wgrad: bad arguments");
  DVT_REQUIRE(dvt_aligned16(x) && dvt_aligned16(dz) && dvt_aligned16(dw) && dvt_aligned16(workspace),
              "dvt_conv3x1_wgrad: buffers must be 16-byte aligned");
  DVT_REQUIRE(!defer_reduce || pending, "dvt_conv3x1_wgrad: defer_reduce needs a pending descriptor to fill");
  if (!dvt_conv3x1_wgrad_supported(N, T, L, kCI, kCO, dtype))
    DVT_UNSUPPORTED("dvt_conv3x1_wgrad: needs a 16-bit dtype, 144 -> 64 channels and a segment length S <= 16 with L %% S == 0, "
                    "(T * S) %% 32 == 0 and two (window + gradient tile) pairs in 160 KiB of LDS");
  TwParams p{};
  int pipe = 0;
  tw_plan(T, L, &p.w, &pipe);
  p.x = x; p.dz = dz; p.slab = (float*)workspace;
  p.z_bytes = p.w.KP * 128;
  p.ntiles = (int)(N * p.w.segs);
  if (x_affine && x_affine->mean) {
    DVT_REQUIRE(x_affine->invstd && x_affine->gamma && x_affine->beta && x_affine->c_valid >= 0 && x_affine->c_valid <= kCI,
                "dvt_conv3x1_wgrad: x_affine needs mean, invstd, gamma, beta and 0 <= c_valid <= 144");
    p.aff = Affine{x_affine->mean, x_affine->invstd, x_affine->gamma, x_affine->beta,
                   x_affine->c_valid > 0 ? x_affine->c_valid : kCI, x_affine->relu};
  }
  const int grid = tw_grid(N, p.w);
  hipStream_t st = (hipStream_t)stream;
  if (pipe) {
    const int lds3 = 3 * (p.w.x_bytes + p.z_bytes);
    if (dtype == DVT_BF16) {
      static DvtLdsAttr set;
      dvt_lds_attr(set, (const void*)conv3x1_wgrad_pipe_kernel<bf16>, 160 * 1024);
      hipLaunchKernelGGL((conv3x1_wgrad_pipe_kernel<bf16>), dim3(grid), dim3(2 * kNW * 64), lds3, st, p);
    } else {
      static DvtLdsAttr set;
      dvt_lds_attr(set, (const void*)conv3x1_wgrad_pipe_kernel<f16>, 160 * 1024);
      hipLaunchKernelGGL((conv3x1_wgrad_pipe_kernel<f16>), dim3(grid), dim3(2 * kNW * 64), lds3, st, p);
    }
  } else
  if (dtype == DVT_BF16) {
    static DvtLdsAttr set;
    dvt_lds_attr(set, (const void*)conv3x1_wgrad_kernel<bf16>, 160 * 1024);
    hipLaunchKernelGGL((conv3x1_wgrad_kernel<bf16>), dim3(grid), dim3(kNW * 64), 2 * (p.w.x_bytes + p.z_bytes), st, p);
  } else {
    static DvtLdsAttr set;
    dvt_lds_attr(set, (const void*)conv3x1_wgrad_kernel<f16>, 160 * 1024);
    hipLaunchKernelGGL((conv3x1_wgrad_kernel<f16>), dim3(grid), dim3(kNW * 64), 2 * (p.w.x_bytes + p.z_bytes), st, p);
  }
  DVT_LAUNCH_CHECK("dvt_conv3x1_wgrad");
  // the slabs are summed by the family's split-K reduce, which scatters [tap * 144 + ci][co] into the parameter's [co][ci][3]
  dvt_splitk_pending q{};
  q.slab = p.slab; q.splits = grid; q.valid = 1; q.M = kM; q.N = kCO; q.C = dw; q.ldc = kCO;
  q.accumulate = accumulate; q.cs_accumulate = 0; q.cs_slab = nullptr; q.cs_out = nullptr;
  q.conv_cin = kCI; q.conv_taps = 3; q.conv_cin_l = 0; q.conv_cout_l = 0;
  if (defer_reduce) {
    *pending = q;
    return DVT_OK;
  }
  return dvt_splitk_reduce_pending(&q, stream);
}

}  // extern "C"
