// conv3x3_rw.hip -- the (1, 3, 3) spatial half of R(2+1)D-18's layer-1 Conv2Plus1D, forward 64 -> 144 mid planes (torchvision's
// layout behind frame_transformer.py:64-74), with the weights in REGISTERS and helper waves (round 6) -- behind the unchanged
// dvt_conv3x3_stream for (Cin, Cout) = (64, 144).
//
//   z[n, h, w, co] = sum over ki, kj, c of x[n, h + ki - 1, w + kj - 1, c] * W[co][(ki * 3 + kj) * 64 + c]
//
// conv3x3_stream.hip streams the 162 KiB of weights through LDS once per 224-pixel tile (7 compute waves x 32 pixels x ALL 144
// channels, 230 registers): 182 us per layer at 336 frames of 56^2, its DMA stream and its MFMAs adding up instead of
// overlapping (file header there).  The scheme of conv3x1_dbn.hip carries over: wave u of NINE compute waves owns the output
// channels [16 u, 16 u + 16) -- its 16 x 576 weights are eighteen fragments, 72 registers, loaded once per launch -- and runs
// every 16-pixel block of a tile from the staged halo patch ([pixel][64 channels] in 144-byte rows: nine 16-byte slots, the odd
// count spreads a ds_read_b128 group over all banks); SEVEN helper waves request the next tile's patch (LDS-DMA, zero page for
// the halo) and store the previous tile from one of two staging images as whole 288-byte pixel rows, carrying the BatchNorm
// partial sums of the STORED values (a thread owns one 16-byte channel group for the launch).  One barrier per tile.
// REJECTED (round 6) -- kept out of the build as the record of a measured attempt; nothing links this file.
//   Built behind dvt_conv3x3_stream for (64, 144); the 48 stream tests passed; isolated at 336 frames of 56^2 (tools/dev/win_time.sh):
//   267 us (277 us with a filter row's six fragment reads grouped in front of its MFMAs) against 195-205 us of
//   conv3x3_stream_kernel<64, 144, 9>.  Why: a compute wave that owns 16 output channels reads EVERY patch fragment of a
//   16-pixel block for ONE 16x16x32 MFMA -- 1 KiB of LDS per 16,384 flops.  Nine such waves move 1.1 MiB of LDS per 112-pixel
//   tile = 8.9k cycles at 128 B/clk against 6.0k MFMA cycles on the busiest SIMD: the kernel is LDS-read bound at ~136 us
//   before conflicts, twice the LDS bytes per pixel of the streamed-weight kernel (whose fragment of weights serves two
//   pixel blocks).  Register-resident weights need 4.5 VGPRs per output channel (648 wave-registers for the layer), so a
//   16-wave workgroup cannot give a wave 32 channels (144 VGPRs of weights under the 128-register cap); conv3x1_dbn.hip gets
//   away with the same scheme because its K is 288, not 576.  DESIGN.md 4.5 "Round 6".
#include "common.h"

namespace {

constexpr int kCI = 64, kCO = 144, kNC = 9, kNH = 7, kNWv = kNC + kNH;
constexpr int kXRow = 144;                   // patch bytes per pixel: 8 data slots + 1 padding slot
constexpr int kSlots = 9;
constexpr int kMaxXP = 6;                    // patch DMA pieces (1 KiB) per helper wave: patch <= 42 KiB
constexpr int kSPitch = 296;                 // staging bytes per pixel (conflict-free ds_write_b64 groups)
constexpr int kCH = kCO / 8;                 // 18 chunks of 16 bytes per pixel
constexpr int kHRows = (kNH * 64) / kCH;     // 24 row lanes among the helpers
constexpr int kMaxNB = 7;                    // 16-pixel blocks per tile: 112 pixels
constexpr int kNR = (kMaxNB * 16 + kHRows - 1) / kHRows;      // 5 rows per helper thread and tile

struct RwParams {
  const void* x;        // [N, H, W, 64]
  const void* w;        // [144][576] k-major, k = tap * 64 + c
  void* y;              // [N, H, W, 144]
  float* bn_partial;    // [grid][2][144] or nullptr
  int H, W, R, PW, TP, NB, tiles_per_img, ntiles;
  int patch_slots, npieces, patch_bytes;
  unsigned magic_pw, magic_w;     // ceil(2^32 / (PW * 9)) for slot -> patch row; ceil(2^32 / W)
};

__device__ __attribute__((aligned(16))) unsigned int rw_zero16[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void rw_wait_vm(int n) {      // n is wave-uniform
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
  }
}

template <typename E>
__global__ __launch_bounds__(kNWv * 64) void conv3x3_rw_kernel(const RwParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using V8 = typename Elem16<E>::v8;
  using V4 = typename Elem16<E>::v4;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pb = p.patch_bytes, gb = p.NB * 16 * kSPitch, goff = 2 * p.patch_bytes;
  const int n_my = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // tiles of this workgroup (>= 1)

  if (wid >= kNC) {
    // ================================================================ helper waves
    const int hw = wid - kNC, htid = threadIdx.x - kNC * 64;
    const E* xg = (const E*)p.x;
    E* yg = (E*)p.y;
    // patch DMA pieces of this wave: patch row << 20 | patch column << 8 | channel of the chunk, bit 31 = never loaded
    unsigned xq[kMaxXP];
#pragma unroll
    for (int i = 0; i < kMaxXP; ++i) {
      const int piece = hw + kNH * i;
      const int sl = piece * 64 + lane;
      const int pr = (int)__umulhi((unsigned)sl, p.magic_pw);
      const int within = sl - pr * (p.PW * kSlots);
      const int pc = within / kSlots, c = within - pc * kSlots;
      const bool ok = piece < p.npieces && sl < p.patch_slots && c < 8 && (unsigned)(pc - 1) < (unsigned)p.W;
      xq[i] = ok ? ((unsigned)pr << 20) | ((unsigned)(pc - 1) << 8) | (unsigned)(c * 8) : 0x80000000u;
    }
    auto tile_of = [&](int j, int& n, int& h0) {
      const int tile = blockIdx.x + j * gridDim.x;
      n = tile / p.tiles_per_img;
      h0 = (tile - n * p.tiles_per_img) * p.R;
    };
    auto load_patch = [&](int j) {
      int n, h0;
      tile_of(j, n, h0);
      const E* base = xg + (int64_t)n * p.H * p.W * kCI;
      char* dst = smem + (j & 1) * pb;
#pragma unroll
      for (int i = 0; i < kMaxXP; ++i) {
        const int piece = hw + kNH * i;
        if (piece < p.npieces) {                   // wave-uniform
          const int h = h0 - 1 + (int)((xq[i] >> 20) & 0x7FF);
          const bool ok = (int)xq[i] >= 0 && (unsigned)h < (unsigned)p.H;
          const E* src = ok ? base + ((int64_t)h * p.W + ((xq[i] >> 8) & 0xFFF)) * kCI + (xq[i] & 0xFF)
                            : reinterpret_cast<const E*>(rw_zero16);
          dvt_dma16(src, dst + piece * 1024);
        }
      }
    };
    // this thread: channel group c18 (8 channels) of the tile pixels rr, rr + 24, ...
    const int rr = htid / kCH, c18 = htid - rr * kCH;
    const bool live = rr < kHRows;
    float bs[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // stores of tile j from its staging image; -> stores this WAVE has certainly issued
    auto store_tile = [&](int j) -> int {
      int n, h0;
      tile_of(j, n, h0);
      const int valid = min(p.R, p.H - h0) * p.W;                  // the tile's pixels are consecutive in y
      E* yt = yg + ((int64_t)n * p.H + h0) * p.W * kCO;
      const char* stage = smem + goff + (j & 1) * gb;
      int issued = 0;
#pragma unroll
      for (int q = 0; q < kNR; ++q) {
        const int m = q * kHRows + rr;
        issued += (q * kHRows + (hw * 64) / kCH < valid && hw * 64 < kHRows * kCH) ? 1 : 0;
        if (live && m < valid) {
          const V4 lo = *reinterpret_cast<const V4*>(stage + m * kSPitch + c18 * 16);
          const V4 hi = *reinterpret_cast<const V4*>(stage + m * kSPitch + c18 * 16 + 8);
          const V8 v = V8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          *reinterpret_cast<V8*>(yt + (int64_t)m * kCO + c18 * 8) = v;
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
    load_patch(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                              // P0: patch 0 has landed
    for (int i = 0; i < n_my; ++i) {
      // interval i (the compute waves run tile i): the request first, then the stores of tile i - 1
      const bool req = i + 1 < n_my;
      if (req) load_patch(i + 1);                                 // into the buffer tile i - 1 has left
      int nst = 0;
      if (i >= 1) nst = store_tile(i - 1);
      // patch i + 1 must have landed before the next interval computes from it; this interval's stores (issued behind the
      // requests: vmcnt retires in issue order) may stay in flight
      rw_wait_vm(nst);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();                                            // B_{i+1}
    }
    store_tile(n_my - 1);
    float* red = reinterpret_cast<float*>(smem);                  // [2][448][8] = 28 KiB over the patches (all reads are done)
    if (p.bn_partial) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        red[(0 * kNH * 64 + htid) * 8 + k] = bs[k];
        red[(1 * kNH * 64 + htid) * 8 + k] = bq[k];
      }
    }
    __syncthreads();                                              // (the compute waves join this one too)
    if (p.bn_partial && htid < 2 * kCO) {
      const int stat = htid / kCO, ch = htid - stat * kCO, cc = ch >> 3, k = ch & 7;
      float t = 0.f;
      for (int j = 0; j < kHRows; ++j) t += red[(stat * kNH * 64 + j * kCH + cc) * 8 + k];
      p.bn_partial[((int64_t)blockIdx.x * 2 + stat) * kCO + ch] = t;
    }
    return;
  }

  // ================================================================== compute waves: wave u <-> output channels [16 u, 16 u + 16)
  const int g = lane >> 4, li = lane & 15;
  const int u = wid;
  V8 wf[9][2];                                     // lane (g, li) <-> weight row 16 u + li, k = tap * 64 + 32 kk + 8 g
  {
    const E* wrow = (const E*)p.w + (int64_t)(16 * u + li) * (9 * kCI);
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) wf[tp][kk] = *reinterpret_cast<const V8*>(wrow + tp * kCI + kk * 32 + g * 8);
  }
  // byte offset of this lane's pixel (tap (0, 0)) inside a patch, per block: pixel m = 16 b + li of the tile, row m / W
  int xoff[kMaxNB];
#pragma unroll
  for (int b = 0; b < kMaxNB; ++b) {
    int m = b * 16 + li;
    m = m < p.TP ? m : 0;                          // (padding pixels of the last block: computed on pixel 0, never stored)
    const int r = (int)__umulhi((unsigned)m, p.magic_w);
    xoff[b] = (r * p.PW + (m - r * p.W)) * kXRow + (g << 4);
  }
  const int row_b = p.PW * kXRow;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                                // P0
  for (int i = 0; i < n_my; ++i) {
    const char* px = smem + (i & 1) * pb;
    char* const stage = smem + goff + (i & 1) * gb;
#pragma unroll
    for (int b = 0; b < kMaxNB; ++b) {
      if (b < p.NB) {                              // wave-uniform
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* xb = px + xoff[b];
        // a filter row's six fragments are requested together and consumed together (left to itself the compiler keeps TWO in
        // flight -- 98 registers -- and every pair of MFMAs waits out an LDS round trip: 267 us per layer)
#pragma unroll
        for (int ki = 0; ki < 3; ++ki) {
          V8 xf[3][2];
#pragma unroll
          for (int kj = 0; kj < 3; ++kj)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) xf[kj][kk] = *reinterpret_cast<const V8*>(xb + ki * row_b + kj * kXRow + kk * 64);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int kj = 0; kj < 3; ++kj)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) acc = Elem16<E>::mma(wf[ki * 3 + kj][kk], xf[kj][kk], acc);
          __builtin_amdgcn_sched_barrier(0);
        }
        // lane (g, li) holds z[pixel 16 b + li][16 u + 4 g .. + 3] -> the staging image (the helpers finished with its previous
        // content, tile i - 2, before the barrier that opened this interval)
        V4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (E)acc[r];
        *reinterpret_cast<V4*>(stage + (b * 16 + li) * kSPitch + u * 32 + g * 8) = o;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                              // B_{i+1}
  }
  __syncthreads();                                                // (the helpers' statistics scratch)
}

int rw_plan(int H, int W, RwParams* q) {
  if (H < 1 || W < 2 || W > 112) return 0;
  int R = (kMaxNB * 16) / W;
  if (R > H) R = H;
  if (R < 1) return 0;
  const int PW = W + 2;
  const int slots = (R + 2) * PW * kSlots;
  const int npieces = (slots + 63) / 64;
  if (npieces > kNH * kMaxXP || R + 2 > 2047 || PW > 4095) return 0;
  q->H = H; q->W = W; q->R = R; q->PW = PW; q->TP = R * W; q->NB = (R * W + 15) / 16;
  q->tiles_per_img = (H + R - 1) / R;
  q->patch_slots = slots; q->npieces = npieces; q->patch_bytes = npieces * 1024;
  const unsigned d = (unsigned)(PW * kSlots);
  q->magic_pw = (unsigned)((((uint64_t)1 << 32) + d - 1) / d);
  q->magic_w = (unsigned)((((uint64_t)1 << 32) + (uint64_t)W - 1) / (uint64_t)W);
  const int lds = 2 * q->patch_bytes + 2 * q->NB * 16 * kSPitch;
  return lds <= 160 * 1024 && lds >= 2 * kNH * 64 * 8 * 4;
}

int rw_grid(int64_t N, const RwParams& q) {
  const int64_t ntiles = N * q.tiles_per_img;
  return (int)(ntiles < dvt_num_cus() ? ntiles : dvt_num_cus());
}

template <typename E>
void rw_launch(const RwParams& p, int grid, int lds, hipStream_t st) {
  static DvtLdsAttr set;
  dvt_lds_attr(set, (const void*)conv3x3_rw_kernel<E>, 160 * 1024);
  hipLaunchKernelGGL((conv3x3_rw_kernel<E>), dim3(grid), dim3(kNWv * 64), lds, st, p);
}

}  // namespace

namespace dvt_internal {

int conv3x3_rw_supported(int64_t N, int H, int W, int dtype) {
#ifdef DVT_NO_RW
  return 0;
#endif
  RwParams q;
  return N > 0 && dvt_is_16bit(dtype) && rw_plan(H, W, &q) && N * q.tiles_per_img < ((int64_t)1 << 31) &&
                 N * H * W < ((int64_t)1 << 30) ? 1 : 0;
}

int conv3x3_rw_parts(int64_t N, int H, int W) {
  RwParams q;
  if (N <= 0 || !rw_plan(H, W, &q)) return 0;
  return rw_grid(N, q);
}

// (arguments checked by dvt_conv3x3_stream)
int conv3x3_rw_fwd(const void* x, const void* w, void* y, float* stats_partial, int64_t N, int H, int W, int dtype, hipStream_t st) {
  RwParams p{};
  if (!rw_plan(H, W, &p)) return DVT_ERR_UNSUPPORTED;
  p.x = x; p.w = w; p.y = y; p.bn_partial = stats_partial;
  p.ntiles = (int)(N * p.tiles_per_img);
  const int grid = rw_grid(N, p);
  const int lds = 2 * p.patch_bytes + 2 * p.NB * 16 * kSPitch;
  if (dtype == DVT_F16) rw_launch<f16>(p, grid, lds, st);
  else rw_launch<bf16>(p, grid, lds, st);
  return DVT_OK;
}

}  // namespace dvt_internal
