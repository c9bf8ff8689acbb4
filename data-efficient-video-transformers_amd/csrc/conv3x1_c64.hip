// conv3x1_c64.hip -- the (3, 1, 1) temporal convolution 64 -> 64 of R(2+1)D-18's STEM (torchvision r2plus1d_18 as used by
// frame_transformer.py:64-74: Conv3d(45, 64, (3, 1, 1), pad (1, 0, 0)) behind the (1, 7, 7) spatial half; the 45 mid planes
// are stored zero-padded to 64) and, with the data-gradient pack of its weights, that layer's data gradient.
//
//   z[n, t, p, co] = sum over kt, ci of x[n, t + kt - 1, p, ci] * W[co][kt * 64 + ci]
//
// The implicit GEMM (gemm256.hip) gathers every pixel of the map once per temporal tap: 130 us per launch at 28 clips of
// 12 x 56^2 for 270 MB of input + output (2 TB/s).  Same scheme as conv3x1_fwd.hip (144 -> 64), with what 64-channel pixels
// change:
//   * window image: [position][64 channels] as eight 16-byte slots + one slot of padding = 144-byte rows (9 is odd: the
//     sixteen positions of a ds_read_b128 access group land on sixteen different 16-byte bank groups); a segment of 16
//     pixels over T + 2 frames is 32 KiB at 12 frames, two of them + the output staging fit with room to spare, so a tile
//     is 16 pixels x all frames (192 positions at 12 frames: half the tiles of the 144-channel kernel per pixel);
//   * eight waves = 4 output-channel blocks of 16 x 2 halves of the tile's position blocks (up to 6 each); a wave's weights
//     (16 output channels x 192: six 16-byte fragments) stay in registers for the whole launch;
//   * 64 channels per tap = two 16x16x32 steps, no 16x16x16 tail;
//   * epilogue as in conv3x1_fwd.hip: LDS staging image, whole 128-byte pixel rows stored, the BatchNorm column sums of the
//     STORED values carried per thread over the workgroup's tile sequence (optional: the data gradient has no use for them).
#include "common.h"

namespace {

constexpr int kC = 64, kNW = 8;
constexpr int kXRow = 144;                   // bytes per window position: 8 data slots + 1 padding slot
constexpr int kSlots = 9;
constexpr int kMaxXP = 5;                    // window DMA pieces (1 KiB) per wave: window <= 40 KiB
constexpr int kMaxPB = 6;                    // 16-position blocks per wave (tile <= 192 positions)

struct Win {
  int T, L, S, segs, KP, xpos, x_bytes;
};

struct TcParams {
  const void* x;        // [N, T, L, 64]
  const void* w;        // [64][ldw] k-major, k = tap * 64 + ci
  void* y;              // [N, T, L, 64]
  float* bn_partial;    // [grid][2][64] or nullptr
  Win w_;
  int ntiles, ldw;
};

__device__ __attribute__((aligned(16))) unsigned int tc_zero16[4] = {0u, 0u, 0u, 0u};

int tc_plan(int T, int L, Win* q) {
  if (T < 1 || L < 1 || T + 2 > 2047) return 0;
  for (int S = 16; S >= 2; --S) {
    if (L % S || (T * S) % 32) continue;
    const int KP = T * S, xpos = (T + 2) * S;
    if ((KP >> 5) > kMaxPB) continue;
    const int xbytes = (xpos * kXRow + 1023) & ~1023;
    if (2 * xbytes + 128 * KP + 4096 > 160 * 1024) continue;
    if ((xbytes >> 10) > kNW * kMaxXP) continue;
    q->T = T; q->L = L; q->S = S; q->segs = L / S; q->KP = KP; q->xpos = xpos; q->x_bytes = xbytes;
    return 1;
  }
  return 0;
}

int tc_grid(int64_t N, const Win& q) {
  const int64_t ntiles = N * q.segs;
  return (int)(ntiles < dvt_num_cus() ? ntiles : dvt_num_cus());
}

template <typename E, int NPB>
__global__ __launch_bounds__(kNW * 64) void conv3x1_c64_kernel(const TcParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using V8 = typename Elem16<E>::v8;
  using V4 = typename Elem16<E>::v4;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const Win& w = p.w_;
  const int S = w.S;
  char* stage = smem + 2 * w.x_bytes;                                            // [KP][64] outputs of a tile, 128-byte rows
  const E* xg = (const E*)p.x;
  const int xp = w.x_bytes >> 10;

  // per-lane coordinates of this wave's window DMA pieces (fixed for the launch):
  // frame row << 20 | pixel of the segment << 8 | channel of the 16-byte chunk, bit 31 = never loaded (padding slot / past the window)
  unsigned xq[kMaxXP];
#pragma unroll
  for (int i = 0; i < kMaxXP; ++i) {
    const int piece = wid + kNW * i;
    const int sl = piece * 64 + lane;
    const int pos = sl / kSlots, c = sl - pos * kSlots;
    const int tt = pos / S, sx = pos - tt * S;
    const bool ok = piece < xp && pos < w.xpos && c < 8;
    xq[i] = ok ? ((unsigned)tt << 20) | ((unsigned)sx << 8) | (unsigned)(c * 8) : 0x80000000u;
  }
  auto load_tile = [&](int tile, int b) {
    const int n = tile / w.segs, sg = tile - n * w.segs;
    const int64_t pix0 = (int64_t)n * w.T * w.L + (int64_t)sg * S;
#pragma unroll
    for (int i = 0; i < kMaxXP; ++i) {
      const int piece = wid + kNW * i;
      if (piece < xp) {                            // wave-uniform
        const int frame = (int)((xq[i] >> 20) & 0x7FF) - 1;
        const bool ok = (int)xq[i] >= 0 && (unsigned)frame < (unsigned)w.T;
        const E* src = ok ? xg + (pix0 + (int64_t)frame * w.L + ((xq[i] >> 8) & 0xFFF)) * kC + (xq[i] & 0xFF)
                          : reinterpret_cast<const E*>(tc_zero16);
        dvt_dma16(src, smem + b * w.x_bytes + piece * 1024);
      }
    }
  };

  // ---- this wave: output channels [16 u, 16 u + 16), position blocks pb0 .. pb0 + NPB - 1
  const int u = wid & 3, half = wid >> 2;
  const int pb0 = half * NPB;                                                    // (KP == 32 NPB)
  V8 wf[3][2];                                     // lane (g, li) <-> weight row 16 u + li, k = tap * 64 + 32 kk + 8 g
  {
    const E* wrow = (const E*)p.w + (int64_t)(16 * u + li) * p.ldw;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) wf[kt][kk] = *reinterpret_cast<const V8*>(wrow + kt * kC + kk * 32 + g * 8);
  }
  // byte offset of this lane's x fragments inside a window, per tap: row = position 16 pb0 + li + S kt, slot 4 kk + g
  int xo[3];
#pragma unroll
  for (int kt = 0; kt < 3; ++kt) xo[kt] = (pb0 * 16 + li + kt * S) * kXRow + (g << 4);

  float bs[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int nst = wid * 8 < w.KP ? (w.KP - wid * 8 + 63) >> 6 : 0;        // output rows (= stores) per thread and tile, wave-uniform
  int tile = blockIdx.x;
  if (tile < p.ntiles) load_tile(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int it = 0; tile < p.ntiles; ++it, tile += gridDim.x) {
    const int cxo = (it & 1) * w.x_bytes;
    if (tile + (int)gridDim.x < p.ntiles) load_tile(tile + gridDim.x, (it + 1) & 1);
    f32x4 acc[NPB];
#pragma unroll
    for (int b = 0; b < NPB; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    V8 xf[2][NPB];                                 // (two sets: the next step's reads under this step's MFMAs)
    auto rdx = [&](int st, V8* dst) {              // st = 2 kt + kk
#pragma unroll
      for (int b = 0; b < NPB; ++b)
        dst[b] = *reinterpret_cast<const V8*>(smem + cxo + xo[st >> 1] + b * (16 * kXRow) + (st & 1) * 64);
    };
    rdx(0, xf[0]);
#pragma unroll
    for (int st = 0; st < 6; ++st) {
      if (st + 1 < 6) rdx(st + 1, xf[(st + 1) & 1]);
#pragma unroll
      for (int b = 0; b < NPB; ++b) acc[b] = Elem16<E>::mma(wf[st >> 1][st & 1], xf[st & 1][b], acc[b]);
    }
    // ---- the tile's outputs: lane (g, li) holds z[position (pb0 + b) * 16 + li][16 u + 4 g .. + 3]
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
        *reinterpret_cast<V8*>(yg + (pix0 + (int64_t)t * w.L + sx) * kC + c * 8) = v;
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
    // the next window has landed; this tile's stores (issued behind its requests, at most three per thread: vmcnt counts in
    // issue order) stay in flight
    if (nst == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (nst == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (nst == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                              // everybody is done with this tile's window and staging
  }
  if (p.bn_partial) {                              // threads with equal (tid & 7) hold the same 8 channels: fixed-order sum
    float* red = reinterpret_cast<float*>(smem);   // [2][512][8] = 32 KiB over the images (all reads of them are done)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      red[(0 * kNW * 64 + threadIdx.x) * 8 + k] = bs[k];
      red[(1 * kNW * 64 + threadIdx.x) * 8 + k] = bq[k];
    }
    __syncthreads();
    if (threadIdx.x < 2 * kC) {
      const int stat = threadIdx.x >> 6, ch = threadIdx.x & 63, c = ch >> 3, k = ch & 7;
      float t = 0.f;
      for (int j = 0; j < (kNW * 64) >> 3; ++j) t += red[(stat * kNW * 64 + j * 8 + c) * 8 + k];
      p.bn_partial[((int64_t)blockIdx.x * 2 + stat) * kC + ch] = t;
    }
  }
}

template <typename E, int NPB>
void tc_launch(const TcParams& p, int grid, int lds, hipStream_t st) {
  static DvtLdsAttr set;
  dvt_lds_attr(set, (const void*)conv3x1_c64_kernel<E, NPB>, 160 * 1024);
  hipLaunchKernelGGL((conv3x1_c64_kernel<E, NPB>), dim3(grid), dim3(kNW * 64), lds, st, p);
}

}  // namespace

namespace dvt_internal {

int conv3x1_c64_supported(int64_t N, int T, int L, int dtype) {
  Win q;
  return N > 0 && dvt_is_16bit(dtype) && tc_plan(T, L, &q) && N * q.segs < ((int64_t)1 << 31) && N * T * L < ((int64_t)1 << 31) ? 1 : 0;
}

int64_t conv3x1_c64_stats_parts(int64_t N, int T, int L) {
  Win q;
  if (N <= 0 || !tc_plan(T, L, &q)) return 0;
  return tc_grid(N, q);                            // one partial row per workgroup of the persistent grid
}

// (arguments checked by dvt_conv3x1_fwd)
int conv3x1_c64_fwd(const void* x, const void* w, int64_t ldw, void* y, float* stats_partial, int64_t N, int T, int L, int dtype,
                    hipStream_t st) {
  TcParams p{};
  if (!tc_plan(T, L, &p.w_)) return DVT_ERR_UNSUPPORTED;
  p.x = x; p.w = w; p.y = y; p.bn_partial = stats_partial; p.ldw = (int)ldw;
  p.ntiles = (int)(N * p.w_.segs);
  const int grid = tc_grid(N, p.w_);
  int lds = 2 * p.w_.x_bytes + p.w_.KP * 128;
  if (lds < 2 * kNW * 64 * 8 * 4) lds = 2 * kNW * 64 * 8 * 4;      // (the statistics scratch [2][512][8] overlays the images)
  const bool h = dtype == DVT_F16;
  switch (p.w_.KP >> 5) {
    case 1: h ? tc_launch<f16, 1>(p, grid, lds, st) : tc_launch<bf16, 1>(p, grid, lds, st); break;
    case 2: h ? tc_launch<f16, 2>(p, grid, lds, st) : tc_launch<bf16, 2>(p, grid, lds, st); break;
    case 3: h ? tc_launch<f16, 3>(p, grid, lds, st) : tc_launch<bf16, 3>(p, grid, lds, st); break;
    case 4: h ? tc_launch<f16, 4>(p, grid, lds, st) : tc_launch<bf16, 4>(p, grid, lds, st); break;
    case 5: h ? tc_launch<f16, 5>(p, grid, lds, st) : tc_launch<bf16, 5>(p, grid, lds, st); break;
    default: h ? tc_launch<f16, 6>(p, grid, lds, st) : tc_launch<bf16, 6>(p, grid, lds, st); break;
  }
  return DVT_OK;
}

}  // namespace dvt_internal
