// conv3x1_fwd.hip -- forward of the temporal half of R(2+1)D-18's layer-1 Conv2Plus1D (video_resnet.py:30-31: a (3, 1, 1)
// convolution, 144 mid planes -> 64 planes, stride 1, pad 1) from the LDS window of conv3x1_window.h.
//
//   z[n, t, p, co] = sum over kt, ci of x[n, t + kt - 1, p, ci] * W[co][kt * 144 + ci]
//
// The implicit GEMM (gemm256.hip, per-lane taps) gathers every pixel of the 144-plane map once per temporal tap (152 - 160 us
// per layer at 28 clips of 12 x 56^2, 303 MB of input).  Here a workgroup owns a segment of S pixels over all T frames,
// stages the (T + 2) x S x 144 window once and reads the three taps at the position offsets 0 / S / 2S:
//   * eight waves = 4 output-channel blocks of 16 x 2 halves of the tile's 16-position blocks; a wave's share of the weights
//     (16 output channels x 432: 12 x 16-byte + 3 x 8-byte fragments, 54 VGPRs) is loaded ONCE per launch and stays in
//     registers, so the only LDS traffic of the main loop is one x row fragment per MFMA (ds_read_b128 / ds_read_b64 on the
//     window image, conflict-free);
//   * 144 channels per tap = four 16x16x32 steps + one 16x16x16 step;
//   * optional virtual BatchNorm: the map is the spatial convolution's output z and relu(z * s + t) is formed in the staged
//     window (window_transform), once per tile -- the normalised 144-plane activation then never exists in HBM;
//   * epilogue: the tile's outputs go through an LDS staging image -- laid over the tile's own, by then consumed, window, so
//     that two windows of 16-pixel segments fit (2 x 70 KiB at 12 frames; with a staging area of its own the segment was 8
//     pixels: twice the tiles, 170 -> 159 us isolated) -- and leave as whole 128-byte pixel rows; the column sums /
//     sums of squares of the STORED values for the BatchNorm behind the layer are carried per thread over the workgroup's
//     whole tile sequence (a thread always stores the same 8 channels) and reduced once per launch.
#include "conv3x1_window.h"

namespace {

using namespace dvt_window;

struct TfParams {
  const void* x;        // [N, T, L, 144]
  const void* w;        // [64][ldw] k-major, k = tap * 144 + ci
  void* y;              // [N, T, L, 64]
  float* bn_partial;    // [grid][2][64] or nullptr
  Window w_;
  Affine aff;
  int ntiles, ldw;
  int nwin, xs;         // pipelined form: window buffers (3 or 4) and the bytes between them
};

template <typename E> struct Mma16f;
template <> struct Mma16f<bf16> {
  static __device__ __forceinline__ f32x4 mma(bf16x4 a, bf16x4 b, f32x4 c) {
    typedef __attribute__((ext_vector_type(4))) short s4;
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s4, a), __builtin_bit_cast(s4, b), c, 0, 0, 0);
  }
};
template <> struct Mma16f<f16> {
  static __device__ __forceinline__ f32x4 mma(f16x4 a, f16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
};

// A 16x16x16 MFMA whose SrcC is the vDst of the 16x16x32 MFMA issued right before it (or the reverse) is not interlocked:
// the listing has the two back to back with no s_nop (hipcc 7.2 counts "same vDst as SrcC" as the hardware's back-to-back
// accumulate case whatever the two opcodes are), and the second one then read two of its four accumulator registers before
// the first had written them (conv3x1_fwd<.., 1>: `v_mfma_f32_16x16x32 v[4:7] .. ; s_waitcnt lgkmcnt(0) ; v_mfma_f32_16x16x16
// v[4:7], .., v[4:7]`).  With NPB position blocks per wave the k-step-major order puts NPB - 1 independent MFMAs between the
// two; below four blocks that distance is made up with wait states: 16 of them (`s_nop 15` = 64 cycles) outlast the 16-cycle
// issue + write-back of either shape.  tests: test_temporal_forward_with_one_two_and_five_position_blocks_per_wave (the
// unfenced build, -DDVT_NO_MFMA_SHAPE_FENCE, fails it at NPB = 1: profiles/r06_mfma_shape_hazard.md).
template <int NPB>
__device__ __forceinline__ void mfma_shape_fence() {
#ifndef DVT_NO_MFMA_SHAPE_FENCE
  if constexpr (NPB < 4) {
    asm volatile("s_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
#endif
}

constexpr int kMaxPB = 6;                    // 16-position blocks per wave (tile <= 192 positions: 16 pixels x 12 frames)
constexpr int kFMaxXP = 9;                   // window DMA pieces per wave: window <= 72 KiB

// NPB = 16-position blocks per wave (KP / 32), a template parameter: with a run-time count every block of the unrolled k
// loop became a branch and the accumulators were copied around them (1,100 v_mov_b64 in the listing)
template <typename E, int NPB>
__global__ __launch_bounds__(kNW * 64) void conv3x1_fwd_kernel(const TfParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using V8 = typename Elem16<E>::v8;
  using V4 = typename Elem16<E>::v4;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const Window& w = p.w_;
  const int S = w.S;
  // (the two window buffers are addressed as smem + offset, never through a pointer picked at run time: a `char* xb[2]`
  //  selected by it & 1 lost its LDS address space and every fragment read became a flat_load)
  const E* xg = (const E*)p.x;
  const bool affine = p.aff.mean != nullptr;
  AffineRegs st{};
  if (affine) window_affine_regs(p.aff, st);
  unsigned xq[kFMaxXP];
  window_coords<kFMaxXP>(w, wid, lane, xq);
  auto load_tile = [&](int tile, int b) {
    const int n = tile / w.segs, sg = tile - n * w.segs;
    window_load<E, kFMaxXP>(w, xg, (int64_t)n * w.T * w.L + (int64_t)sg * S, xq, wid, smem + b * w.x_bytes);
  };

  // ---- this wave: output channels [16 u, 16 u + 16), position blocks pb0 .. pb0 + npb - 1
  const int u = wid & 3, half = wid >> 2;
  constexpr int npb = NPB;
  const int pb0 = half * npb;                                                    // (KP == 32 NPB)
  // weights of the wave's 16 output channels, in registers for the whole launch: lane (g, li) <-> row 16 u + li, k = .. + 8 g
  V8 wf[3][4];
  V4 wr[3];
  {
    const E* wrow = (const E*)p.w + (int64_t)(16 * u + li) * p.ldw;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) wf[kt][kk] = *reinterpret_cast<const V8*>(wrow + kt * kCI + kk * 32 + g * 8);
      wr[kt] = *reinterpret_cast<const V4*>(wrow + kt * kCI + 128 + g * 4);
    }
  }
  // byte offset of this lane's x fragments inside a window, per tap: row = position 16 pb0 + li + S kt, 16-byte fragment kk =
  // channels 32 kk + 8 g -> block 2 kk + (g >> 1) (+ 1 where bit 3 of the position is set), half g & 1; the 8-byte fragment =
  // channels 128 + 4 g.  Position block b and k-step kk add the compile-time constants 16 b x 320 and 64 kk (16 positions
  // leave bit 3 alone), which the reads carry in their offset field: three registers per lane instead of sixty hoisted ones.
  int xo[3], xo4[3];
#pragma unroll
  for (int kt = 0; kt < 3; ++kt) {
    const int pos = pb0 * 16 + li + kt * S, b3 = (pos >> 3) & 1;
    xo[kt] = pos * kXRow + (((g >> 1) + b3) << 5) + ((g & 1) << 4);
    xo4[kt] = pos * kXRow + ((8 + b3) << 5) + (g << 3);
  }

  float bs[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int nst = wid * 8 < w.KP ? (w.KP - wid * 8 + 63) >> 6 : 0;        // output rows (= stores) per thread and tile, wave-uniform
  int tile = blockIdx.x;
  if (tile < p.ntiles) load_tile(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (affine) {
    window_transform<E>(w, smem, st, p.aff.relu);
    __syncthreads();
  }
  for (int it = 0; tile < p.ntiles; ++it, tile += gridDim.x) {
    const int cxo = (it & 1) * w.x_bytes;
    if (tile + (int)gridDim.x < p.ntiles) load_tile(tile + gridDim.x, (it + 1) & 1);
    f32x4 acc[NPB];
#pragma unroll
    for (int b = 0; b < NPB; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      // fragments of every position block first, then the MFMAs by k-step across the blocks: a 16x16x16 step that follows the
      // 16x16x32 step on the SAME accumulator back to back read two of its four result registers early (tap 0, channels
      // 96 .. 127 of the middle block came out partly missing) -- dependent MFMAs of different shapes are kept apart
      V8 xf[2][NPB];                             // (double buffer over the k-steps: the next step's reads under this step's MFMAs)
      V4 x4[NPB];
      auto rdx = [&](int kk, V8* dst) {
#pragma unroll
        for (int b = 0; b < NPB; ++b) dst[b] = *reinterpret_cast<const V8*>(smem + cxo + xo[kt] + b * (16 * kXRow) + kk * 64);
      };
      rdx(0, xf[0]);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk + 1 < 4) {
          rdx(kk + 1, xf[(kk + 1) & 1]);
        } else {
#pragma unroll
          for (int b = 0; b < NPB; ++b) x4[b] = *reinterpret_cast<const V4*>(smem + cxo + xo4[kt] + b * (16 * kXRow));
        }
#pragma unroll
        for (int b = 0; b < NPB; ++b) acc[b] = Elem16<E>::mma(wf[kt][kk], xf[kk & 1][b], acc[b]);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_shape_fence<NPB>();
#pragma unroll
      for (int b = 0; b < NPB; ++b) acc[b] = Mma16f<E>::mma(wr[kt], x4[b], acc[b]);
      __builtin_amdgcn_sched_barrier(0);
      mfma_shape_fence<NPB>();                     // (the next tap's first 16x16x32 step takes the same accumulators)
    }
    // ---- the tile's outputs: lane (g, li) holds z[position (pb0 + b) * 16 + li][16 u + 4 g .. + 3].  The staging image
    // [KP][64] (128-byte rows) OVERLAYS the tile's own window, which nobody reads any more behind this barrier: two windows
    // of 16-pixel segments (2 x 70 KiB at 12 frames) then fit, i.e. half as many tiles -- and a tile's cost here is its
    // chain of phases more than its bytes (conv3x1_window.h)
    __syncthreads();
    char* const stage = smem + cxo;
#pragma unroll
    for (int b = 0; b < NPB; ++b) {
      V4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (E)acc[b][r];
      const int pos = (pb0 + b) * 16 + li;
      *reinterpret_cast<V4*>(stage + pos * 128 + (((2 * u + (g >> 1)) ^ (pos & 7)) << 4) + ((g & 1) << 3)) = o;
    }
    __syncthreads();                               // staging complete
    {
      const int n = tile / w.segs, sg = tile - n * w.segs;
      const int64_t pix0 = (int64_t)n * w.T * w.L + (int64_t)sg * S;
      E* yg = (E*)p.y;
      const int c = threadIdx.x & 7;               // this thread's 8 channels, the same for every tile
      for (int r = threadIdx.x >> 3; r < w.KP; r += (kNW * 64) >> 3) {
        const V8 v = *reinterpret_cast<const V8*>(stage + r * 128 + ((c ^ (r & 7)) << 4));
        const int t = r / S, sx = r - t * S;
        *reinterpret_cast<V8*>(yg + (pix0 + (int64_t)t * w.L + sx) * kCO + c * 8) = v;
        if (p.bn_partial) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float f = (float)v[k];
            bs[k] += f;
            bq[k] = fmaf(f, f, bq[k]);
          }
        }
      }
    }
    // the next window has landed; this tile's stores (issued behind its requests, at most two per thread: vmcnt counts in
    // issue order) stay in flight -- waiting for them as well cost ~3 us per tile, 130 us per launch
    if (nst == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (nst == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (nst == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                              // everybody is done with this tile's window and staging
    if (affine && tile + (int)gridDim.x < p.ntiles) {
      window_transform<E>(w, smem + ((it + 1) & 1) * w.x_bytes, st, p.aff.relu);
      __syncthreads();
    }
  }
  if (p.bn_partial) {                              // threads with equal (tid & 7) hold the same 8 channels: fixed-order sum
    float* red = reinterpret_cast<float*>(smem);   // [2][512][8] = 32 KiB over the window buffers (all reads of them are done)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      red[(0 * kNW * 64 + threadIdx.x) * 8 + k] = bs[k];
      red[(1 * kNW * 64 + threadIdx.x) * 8 + k] = bq[k];
    }
    __syncthreads();
    if (threadIdx.x < 2 * kCO) {
      const int stat = threadIdx.x >> 6, ch = threadIdx.x & 63, c = ch >> 3, k = ch & 7;
      float t = 0.f;
      for (int j = 0; j < (kNW * 64) >> 3; ++j) t += red[(stat * kNW * 64 + j * 8 + c) * 8 + k];
      p.bn_partial[((int64_t)blockIdx.x * 2 + stat) * kCO + ch] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same convolution with a tile's PHASES overlapped by construction (round 6).  SQ counters of the kernel above
// (profiles/r06_conv3x1_bound.md): MFMA pipe 18 % busy, VALU 31 %, LDS 39 %, the waves parked at s_waitcnt / s_barrier for
// 45 % of their cycles -- a chain of phases (window lands -> transform -> fragment reads + MFMAs -> staging -> stores) that
// one workgroup per CU runs one after the other, and that identical co-resident workgroups run in lockstep.  Here sixteen
// waves share ONE barrier per tile and different work:
//   waves 0 - 7   compute tile i from window buffer i % 3 (weights in registers, as above) and leave its outputs in staging
//                 buffer i % 2;
//   waves 8 - 15  in the same interval request window i + 2 (LDS-DMA into the buffer tile i - 1 has just left), store tile
//                 i - 1 from its staging buffer as whole 128-byte rows (carrying the BatchNorm partial sums), and apply the
//                 virtual BatchNorm to window i + 1, which landed before the barrier that opened the interval.
// Three windows + two staging images fit for 8-pixel segments (3 x 35 + 2 x 12 KiB at 12 frames); a window has a whole
// interval to land before anybody waits for it.
__device__ __forceinline__ void wait_vm_n(int n) {      // n is wave-uniform
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
  }
}

template <typename E, int NPB>
__global__ __launch_bounds__(2 * kNW * 64) void conv3x1_fwd_pipe_kernel(const TfParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using V8 = typename Elem16<E>::v8;
  using V4 = typename Elem16<E>::v4;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Window& w = p.w_;
  const int S = w.S;
  // window buffers `xs` bytes apart: with four of them a window's trailing zero frame IS the next buffer's leading one (both are
  // always zero; the request that rewrites one writes zeros over zeros), which is what lets four fit beside two staging images
  const int xs = p.xs, nwin = p.nwin, gb = w.KP * 128, goff = (nwin - 1) * p.xs + w.x_bytes;
  const int n_my = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // tiles of this workgroup (>= 1)

  if (wid >= kNW) {
    // ================================================================ helper waves
    const int hw = wid - kNW, htid = threadIdx.x - kNW * 64;
    const E* xg = (const E*)p.x;
    E* yg = (E*)p.y;
    const bool affine = p.aff.mean != nullptr;
    AffineRegs st{};
    if (affine) window_affine_regs(p.aff, st, htid);
    unsigned xq[kFMaxXP];
    window_coords<kFMaxXP>(w, hw, lane, xq);
    auto load_tile = [&](int j) {
      const int tile = blockIdx.x + j * gridDim.x;
      const int n = tile / w.segs, sg = tile - n * w.segs;
      window_load<E, kFMaxXP>(w, xg, (int64_t)n * w.T * w.L + (int64_t)sg * S, xq, hw, smem + (j % nwin) * xs);
    };
    int np = 0;                                   // window pieces THIS wave requests per tile
#pragma unroll
    for (int i = 0; i < kFMaxXP; ++i) np += hw + kNW * i < (w.x_bytes >> 10) ? 1 : 0;
    const int ahead = nwin - 1;                   // windows requested ahead of the one being computed
    float bs[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int c = htid & 7;                       // this thread's 8 channels, the same for every tile
    // the (at most three) output rows this thread stores of every tile: element offset of row r = frame t, pixel sx of the segment
    int roff[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int r = q * ((kNW * 64) >> 3) + (htid >> 3);
      const int t = r / S, sx = r - t * S;
      roff[q] = (t * w.L + sx) * kCO + c * 8;
    }
    // stores of tile j from its staging buffer; -> the number of store instructions this WAVE has certainly issued
    auto store_tile = [&](int j) -> int {
      const int tile = blockIdx.x + j * gridDim.x;
      const int n = tile / w.segs, sg = tile - n * w.segs;
      E* yt = yg + ((int64_t)n * w.T * w.L + (int64_t)sg * S) * kCO;
      const char* stage = smem + goff + (j & 1) * gb;
      int issued = 0;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int r0 = q * ((kNW * 64) >> 3);
        const int r = r0 + (htid >> 3);
        issued += r0 + hw * 8 < w.KP ? 1 : 0;     // (some lane of this wave is in range: the store is issued)
        if (r < w.KP) {
          const V8 v = *reinterpret_cast<const V8*>(stage + r * 128 + ((c ^ (r & 7)) << 4));
          *reinterpret_cast<V8*>(yt + roff[q]) = v;
          if (p.bn_partial) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const float f = (float)v[k];
              bs[k] += f;
              bq[k] = fmaf(f, f, bq[k]);
            }
          }
        }
      }
      return issued;
    };
    load_tile(0);
    if (n_my > 1) load_tile(1);
    if (ahead > 2 && n_my > 2) load_tile(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                              // P0: the first windows have landed
    if (affine) window_transform<E>(w, smem, st, p.aff.relu);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                              // P1: window 0 is ready
    for (int i = 0; i < n_my; ++i) {
      const bool req = i + ahead < n_my;
      if (req) load_tile(i + ahead);                              // into the buffer tile i - 1 has left
      int nst = 0;
      if (i >= 1) nst = store_tile(i - 1);
      if (affine && i + 1 < n_my) window_transform<E>(w, smem + ((i + 1) % nwin) * xs, st, p.aff.relu);
      // Window i + 2 must have landed before the next interval transforms it.  vmcnt retires in issue order: behind its
      // requests this wave has issued (three windows ahead: the requests of window i + 3 and) this interval's stores, which
      // may all stay in flight.
      wait_vm_n((ahead > 2 && req ? np : 0) + nst);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();                                            // B_{i+1}
    }
    store_tile(n_my - 1);
    float* red = reinterpret_cast<float*>(smem);                  // [2][512][8] over the window buffers (all reads are done)
    if (p.bn_partial) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        red[(0 * kNW * 64 + htid) * 8 + k] = bs[k];
        red[(1 * kNW * 64 + htid) * 8 + k] = bq[k];
      }
    }
    __syncthreads();                                              // (the compute waves join this one too)
    if (p.bn_partial && htid < 2 * kCO) {
      const int stat = htid >> 6, ch = htid & 63, cc = ch >> 3, k = ch & 7;
      float t = 0.f;
      for (int j = 0; j < (kNW * 64) >> 3; ++j) t += red[(stat * kNW * 64 + j * 8 + cc) * 8 + k];
      p.bn_partial[((int64_t)blockIdx.x * 2 + stat) * kCO + ch] = t;
    }
    return;
  }

  // ================================================================== compute waves
  const int g = lane >> 4, li = lane & 15;
  const int u = wid & 3, half = wid >> 2;
  const int pb0 = half * NPB;
  V8 wf[3][4];
  V4 wr[3];
  {
    const E* wrow = (const E*)p.w + (int64_t)(16 * u + li) * p.ldw;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) wf[kt][kk] = *reinterpret_cast<const V8*>(wrow + kt * kCI + kk * 32 + g * 8);
      wr[kt] = *reinterpret_cast<const V4*>(wrow + kt * kCI + 128 + g * 4);
    }
  }
  int xo[3], xo4[3];
#pragma unroll
  for (int kt = 0; kt < 3; ++kt) {
    const int pos = pb0 * 16 + li + kt * S, b3 = (pos >> 3) & 1;
    xo[kt] = pos * kXRow + (((g >> 1) + b3) << 5) + ((g & 1) << 4);
    xo4[kt] = pos * kXRow + ((8 + b3) << 5) + (g << 3);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                                // P0
  __syncthreads();                                                // P1
  for (int i = 0; i < n_my; ++i) {
    const int cxo = (i % nwin) * xs;
    f32x4 acc[NPB];
#pragma unroll
    for (int b = 0; b < NPB; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      V8 xf[2][NPB];
      V4 x4[NPB];
      auto rdx = [&](int kk, V8* dst) {
#pragma unroll
        for (int b = 0; b < NPB; ++b) dst[b] = *reinterpret_cast<const V8*>(smem + cxo + xo[kt] + b * (16 * kXRow) + kk * 64);
      };
      rdx(0, xf[0]);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk + 1 < 4) {
          rdx(kk + 1, xf[(kk + 1) & 1]);
        } else {
#pragma unroll
          for (int b = 0; b < NPB; ++b) x4[b] = *reinterpret_cast<const V4*>(smem + cxo + xo4[kt] + b * (16 * kXRow));
        }
#pragma unroll
        for (int b = 0; b < NPB; ++b) acc[b] = Elem16<E>::mma(wf[kt][kk], xf[kk & 1][b], acc[b]);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_shape_fence<NPB>();
#pragma unroll
      for (int b = 0; b < NPB; ++b) acc[b] = Mma16f<E>::mma(wr[kt], x4[b], acc[b]);
      __builtin_amdgcn_sched_barrier(0);
      mfma_shape_fence<NPB>();
    }
    // the tile's outputs: lane (g, li) holds z[position (pb0 + b) * 16 + li][16 u + 4 g .. + 3] -> staging buffer i % 2 (the
    // helpers stored its previous content, tile i - 2, before the barrier that opened this interval)
    char* const stage = smem + goff + (i & 1) * gb;
#pragma unroll
    for (int b = 0; b < NPB; ++b) {
      V4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (E)acc[b][r];
      const int pos = (pb0 + b) * 16 + li;
      *reinterpret_cast<V4*>(stage + pos * 128 + (((2 * u + (g >> 1)) ^ (pos & 7)) << 4) + ((g & 1) << 3)) = o;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                              // B_{i+1}
  }
  __syncthreads();                                                // (the helpers' statistics scratch)
}

// geometry of the pipelined form: three windows + two staging images; 0 = not taken (the kernel above then)
// -> number of window buffers (3, or 4 with shared zero frames), 0 = not taken; *xs = bytes between the buffers
int tfp_plan(int T, int L, Window* q, int* xs) {
#ifdef DVT_TF_NO_PIPE
  return 0;
#endif
  if (!window_plan(T, L, q, 0, 2 * 128, 3, kNW * kFMaxXP)) return 0;
  if ((q->KP >> 5) < 1 || (q->KP >> 5) > kMaxPB || 3 * q->x_bytes + 2 * q->KP * 128 < 2 * kNW * 64 * 8 * 4) return 0;
  *xs = q->x_bytes;
  // Three buffers.  (Measured and dropped, tools/dev/tf_probe.py on one box: FOUR buffers whose shared zero frames overlap --
  // the kernel takes nwin / xs for it -- so that a window has two intervals to land: 147 - 150 us against 142 - 145 us; the
  // requests were never the wait.  Transform before the stores: 149 - 155 us.  The ablations say why: without the requests
  // 116 us, without the stores 121, without the transform 123 of 141 -- the helpers' three jobs each cost what they issue.)
  return 3;
}

template <typename E, int NPB>
void tfp_launch(const TfParams& p, int grid, int lds, hipStream_t st) {
  static DvtLdsAttr set;
  dvt_lds_attr(set, (const void*)conv3x1_fwd_pipe_kernel<E, NPB>, 160 * 1024);
  hipLaunchKernelGGL((conv3x1_fwd_pipe_kernel<E, NPB>), dim3(grid), dim3(2 * kNW * 64), lds, st, p);
}

int tf_plan(int T, int L, Window* q) {
  if (!window_plan(T, L, q, 0, 0, 2, kNW * kFMaxXP)) return 0;   // (the output staging overlays a window)
  return (q->KP >> 5) <= kMaxPB && 2 * q->x_bytes >= 2 * kNW * 64 * 8 * 4;      // (the statistics scratch overlays the windows)
}

int tf_grid(int64_t N, const Window& q) {
  const int64_t ntiles = N * q.segs;
  return (int)(ntiles < dvt_num_cus() ? ntiles : dvt_num_cus());
}

template <typename E, int NPB>
void tf_launch(const TfParams& p, int grid, int lds, hipStream_t st) {
  static DvtLdsAttr set;
  dvt_lds_attr(set, (const void*)conv3x1_fwd_kernel<E, NPB>, 160 * 1024);
  hipLaunchKernelGGL((conv3x1_fwd_kernel<E, NPB>), dim3(grid), dim3(kNW * 64), lds, st, p);
}

}  // namespace

extern "C" {

int dvt_conv3x1_fwd_supported(int64_t N, int T, int L, int Cin, int Cout, int dtype) {
  if (Cin == 64 && Cout == 64) return dvt_internal::conv3x1_c64_supported(N, T, L, dtype);
  Window q;
  return N > 0 && Cin == kCI && Cout == kCO && dvt_is_16bit(dtype) && tf_plan(T, L, &q) && N * q.segs < ((int64_t)1 << 31) &&
                 N * T * L < ((int64_t)1 << 31) ? 1 : 0;
}

int64_t dvt_conv3x1_fwd_stats_parts(int64_t N, int T, int L, int Cin) {
  if (Cin == 64) return dvt_internal::conv3x1_c64_stats_parts(N, T, L);
  Window q;
  if (N <= 0 || !tf_plan(T, L, &q)) return 0;
  Window qp;
  int xs_ = 0;
  if (tfp_plan(T, L, &qp, &xs_)) q = qp;           // (the launcher's choice: the pipelined form where its buffers fit)
  return tf_grid(N, q);                            // one partial row per workgroup of the persistent grid
}

int dvt_conv3x1_fwd(const void* x, const dvt_bn_affine* x_affine, const void* w, int64_t ldw, void* y, float* stats_partial,
                    int64_t N, int T, int L, int Cin, int dtype, dvt_stream_t stream) {
  DVT_REQUIRE(x && w && y && N >= 0 && T > 0 && L > 0 && (Cin == kCI || Cin == 64) && ldw >= 3 * Cin && ldw % 8 == 0,
              "dvt_conv3x1_fwd: bad arguments");
  DVT_REQUIRE(dvt_aligned16(x) && dvt_aligned16(w) && dvt_aligned16(y) && dvt_aligned16(stats_partial),
              "dvt_conv3x1_fwd: buffers must be 16-byte aligned");
  if (N == 0) return DVT_OK;
  if (Cin == 64) {                                 // the stem's temporal half (and its data gradient): conv3x1_c64.hip
    DVT_REQUIRE(!(x_affine && x_affine->mean), "dvt_conv3x1_fwd: x_affine is for the 144-channel form");
    if (!dvt_internal::conv3x1_c64_supported(N, T, L, dtype))
      DVT_UNSUPPORTED("dvt_conv3x1_fwd (64 channels): needs a 16-bit dtype and a segment length S <= 16 with L %% S == 0, "
                      "(T * S) %% 32 == 0, T * S <= 192 and two windows + the output staging in 160 KiB of LDS");
    const int rc = dvt_internal::conv3x1_c64_fwd(x, w, ldw, y, stats_partial, N, T, L, dtype, (hipStream_t)stream);
    if (rc != DVT_OK) return rc;
    DVT_LAUNCH_CHECK("dvt_conv3x1_fwd(64 channels)");
    return DVT_OK;
  }
  if (!dvt_conv3x1_fwd_supported(N, T, L, kCI, kCO, dtype))
    DVT_UNSUPPORTED("dvt_conv3x1_fwd: needs a 16-bit dtype, 144 -> 64 channels and a segment length S <= 16 with L %% S == 0, "
                    "(T * S) %% 32 == 0, T * S <= 192 and two windows in 160 KiB of LDS");
  TfParams p{};
  tf_plan(T, L, &p.w_);
  Window qp;
  int xs_ = 0;
  const int nwin = tfp_plan(T, L, &qp, &xs_);
  const bool pipe = nwin != 0;
  if (pipe) { p.w_ = qp; p.nwin = nwin; p.xs = xs_; }
  p.x = x; p.w = w; p.y = y; p.bn_partial = stats_partial; p.ldw = (int)ldw;
  p.ntiles = (int)(N * p.w_.segs);
  if (x_affine && x_affine->mean) {
    DVT_REQUIRE(x_affine->invstd && x_affine->gamma && x_affine->beta && x_affine->c_valid >= 0 && x_affine->c_valid <= kCI,
                "dvt_conv3x1_fwd: x_affine needs mean, invstd, gamma, beta and 0 <= c_valid <= 144");
    p.aff = Affine{x_affine->mean, x_affine->invstd, x_affine->gamma, x_affine->beta,
                   x_affine->c_valid > 0 ? x_affine->c_valid : kCI, x_affine->relu};
  }
  const int grid = tf_grid(N, p.w_);
  hipStream_t st = (hipStream_t)stream;
  const bool h = dtype == DVT_F16;
  if (pipe) {
    const int lds = (nwin - 1) * p.xs + p.w_.x_bytes + 2 * p.w_.KP * 128;
    switch (p.w_.KP >> 5) {
      case 1: h ? tfp_launch<f16, 1>(p, grid, lds, st) : tfp_launch<bf16, 1>(p, grid, lds, st); break;
      case 2: h ? tfp_launch<f16, 2>(p, grid, lds, st) : tfp_launch<bf16, 2>(p, grid, lds, st); break;
      case 3: h ? tfp_launch<f16, 3>(p, grid, lds, st) : tfp_launch<bf16, 3>(p, grid, lds, st); break;
      case 4: h ? tfp_launch<f16, 4>(p, grid, lds, st) : tfp_launch<bf16, 4>(p, grid, lds, st); break;
      case 5: h ? tfp_launch<f16, 5>(p, grid, lds, st) : tfp_launch<bf16, 5>(p, grid, lds, st); break;
      default: h ? tfp_launch<f16, 6>(p, grid, lds, st) : tfp_launch<bf16, 6>(p, grid, lds, st); break;
    }
    DVT_LAUNCH_CHECK("dvt_conv3x1_fwd(pipelined)");
    return DVT_OK;
  }
  const int lds = 2 * p.w_.x_bytes;
  switch (p.w_.KP >> 5) {
    case 1: h ? tf_launch<f16, 1>(p, grid, lds, st) : tf_launch<bf16, 1>(p, grid, lds, st); break;
    case 2: h ? tf_launch<f16, 2>(p, grid, lds, st) : tf_launch<bf16, 2>(p, grid, lds, st); break;
    case 3: h ? tf_launch<f16, 3>(p, grid, lds, st) : tf_launch<bf16, 3>(p, grid, lds, st); break;
    case 4: h ? tf_launch<f16, 4>(p, grid, lds, st) : tf_launch<bf16, 4>(p, grid, lds, st); break;
    case 5: h ? tf_launch<f16, 5>(p, grid, lds, st) : tf_launch<bf16, 5>(p, grid, lds, st); break;
    default: h ? tf_launch<f16, 6>(p, grid, lds, st) : tf_launch<bf16, 6>(p, grid, lds, st); break;
  }
  DVT_LAUNCH_CHECK("dvt_conv3x1_fwd");
  return DVT_OK;
}

}  // extern "C"
