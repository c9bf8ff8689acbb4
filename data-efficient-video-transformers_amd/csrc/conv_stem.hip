// conv_stem.hip -- the 7x7 / stride 2 / pad 3 stem convolution on 3-channel frames (custom_resnet.py:100 `conv1`, and the
// (1, 7, 7) spatial half of torchvision's R(2+1)D stem behind frame_transformer.py:64-74), 64 output channels, from an LDS
// halo patch with the weights held in registers.
//
// Input: the PIXEL-PAIR map of dvt_nchw_to_nhwc_pad(.., 4): [N, H, Wp = W/2, 8] -- two horizontally adjacent pixels x (3
// channels + one zero) per 16-byte chunk.  In that view the stem is a (7, 4) convolution of stride (2, 1), pad (3, 2) over
// pairs (dvt_conv_weight_pairs re-lays the weights; the superfluous last output column is never computed), and one
// 16x16x32 MFMA k-step is exactly ONE filter row: lane group g <-> pair column kj, 8 k = one pair.
//
//   z[n, oy, ox, co] = sum over ki < 7, kj < 4, c8 < 8 of P[n, 2 oy - 3 + ki, ox - 2 + kj, c8] * W[co][(ki * 4 + kj) * 8 + c8]
//
// The implicit GEMM (gemm256.hip, dvt_conv2d_implicit with the pair geometry) gathers the 28 chunks of an output pixel
// through per-lane DMA addresses: 248 - 253 us for 256 frames of 224^2 against the 84 us its 411 MB of output cost at
// 5 TB/s, each input chunk fetched 14 times from L2.  Here:
//   * a persistent workgroup (8 waves) owns R whole output rows of one frame (R * Wo <= 896 pixels); the (2 R + 5) x
//     (Wp + 3) pair patch is staged ONCE per tile by LDS-DMA (zero page for rows / columns outside the frame) and the next
//     tile's patch lands in the second buffer under this tile's MFMAs;
//   * a wave computes ALL 64 output channels of its 16-pixel blocks: its 4 x 7 weight fragments (112 VGPRs) are loaded once
//     per launch, so the main loop's only LDS traffic is ONE ds_read_b128 per four MFMAs (the x fragment of a filter row:
//     16 consecutive pixels x 4 pair columns = 19 consecutive chunks, conflict-free for the instruction's lane groups);
//   * outputs leave through a wave-private 2 KiB staging image as whole 128-byte pixel rows; the column sums / sums of
//     squares of the STORED values for the BatchNorm behind the layer are carried per lane (a lane always stores the same 8
//     channels) over the workgroup's whole tile sequence: one partial row per workgroup (dvt_bn_stats_from_partials).
#include "common.h"

namespace {

constexpr int kNW = 8;                       // waves per workgroup
constexpr int kMaxPieces = 6;                // 1 KiB DMA pieces per wave and patch: patch <= 48 KiB
constexpr int kStage = 2048;                 // staging bytes per wave: 16 pixels x 128 bytes
constexpr int kTilePix = 896;                // output pixels per tile: 7 blocks of 16 per wave

struct StemParams {
  const void* x;        // [N, H, Wp, 8] pair map
  const void* w;        // [64][ldw], k = (ki * 4 + kj) * 8 + c8
  void* y;              // [N, Ho, Wo, 64]
  float* bn_partial;    // [grid][2][64] or nullptr
  int H, Wp, Ho, Wo, R, tiles_per_img, ntiles, ldw;
  int PPR;              // pair chunks per patch row: Wp + 3 (two left of the frame, one right)
  int prows;            // patch rows: 2 R + 5
  int npieces;          // DMA pieces of a patch
  int patch_bytes;      // npieces KiB
  unsigned magic_ppr, magic_wo;    // ceil(2^32 / PPR), ceil(2^32 / Wo)
};

__device__ __attribute__((aligned(16))) unsigned int stem_zero16[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void wait_vm_upto(int n) {      // n is wave-uniform, 0 .. 14
  switch (n) {
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

template <typename E>
__global__ __launch_bounds__(kNW * 64) void conv_stem_kernel(const StemParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using V8 = typename Elem16<E>::v8;
  using V4 = typename Elem16<E>::v4;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const E* xg = (const E*)p.x;
  // LDS: [2 patch buffers][8 staging images]; buffers are addressed as smem + offset (a pointer picked at run time would
  // lose its address space: tools/check_flat_ops.py)
  char* const stg = smem + 2 * p.patch_bytes + wid * kStage;

  // ---- per-lane constants of this wave's DMA pieces: patch row << 12 | pair column, bit 31 = never loaded
  unsigned pq[kMaxPieces];
#pragma unroll
  for (int i = 0; i < kMaxPieces; ++i) {
    const int piece = wid + kNW * i;
    const int slot = piece * 64 + lane;
    const int r = (int)__umulhi((unsigned)slot, p.magic_ppr);
    const int pc = slot - r * p.PPR;
    const bool ok = piece < p.npieces && r < p.prows && (unsigned)(pc - 2) < (unsigned)p.Wp;
    pq[i] = ok ? ((unsigned)r << 12) | (unsigned)(pc - 2) : 0x80000000u;
  }
  auto load_tile = [&](int tile, int b) {
    const int n = tile / p.tiles_per_img, oy0 = (tile - n * p.tiles_per_img) * p.R;
    const int iy0 = 2 * oy0 - 3;
    const E* base = xg + (int64_t)n * p.H * p.Wp * 8;
    char* dst = smem + b * p.patch_bytes;
#pragma unroll
    for (int i = 0; i < kMaxPieces; ++i) {
      const int piece = wid + kNW * i;
      if (piece < p.npieces) {                        // wave-uniform
        const int iy = iy0 + (int)((pq[i] >> 12) & 0x7FFFF);
        const bool ok = (int)pq[i] >= 0 && (unsigned)iy < (unsigned)p.H;
        const E* src = ok ? base + ((int64_t)iy * p.Wp + (pq[i] & 0xFFF)) * 8 : reinterpret_cast<const E*>(stem_zero16);
        dvt_dma16(src, dst + piece * 1024);
      }
    }
  };

  // ---- the wave's weights: all 64 output channels x 224, in registers for the whole launch
  // lane (g, li) <-> row 16 u + li of the weights, k = 32 ki + 8 g .. + 7 (filter row ki, pair column g)
  V8 wf[4][7];
  {
    const E* wrow = (const E*)p.w + (int64_t)li * p.ldw + g * 8;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int ki = 0; ki < 7; ++ki) wf[u][ki] = *reinterpret_cast<const V8*>(wrow + (int64_t)(16 * u) * p.ldw + ki * 32);
  }

  float bs[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int prow_bytes = p.PPR * 16;
  int tile = blockIdx.x;
  if (tile < p.ntiles) load_tile(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int it = 0; tile < p.ntiles; ++it, tile += gridDim.x) {
    const int cur = (it & 1) * p.patch_bytes;
    const bool more = tile + (int)gridDim.x < p.ntiles;
    if (more) load_tile(tile + gridDim.x, (it + 1) & 1);
    const int n = tile / p.tiles_per_img, oy0 = (tile - n * p.tiles_per_img) * p.R;
    const int rows_ok = min(p.R, p.Ho - oy0);
    const int npix = rows_ok * p.Wo;
    const int nblk = (npix + 15) >> 4;
    E* yt = (E*)p.y + ((int64_t)n * p.Ho + oy0) * p.Wo * 64;
    int nstores = 0;
    for (int blk = wid; blk < nblk; blk += kNW) {
      // lane (g, li): pixel blk * 16 + li of the tile (padding pixels of the last block compute pixel 0, never stored)
      int m = blk * 16 + li;
      m = m < npix ? m : 0;
      const int oyl = (int)__umulhi((unsigned)m, p.magic_wo);
      const int ox = m - oyl * p.Wo;
      const char* xb = smem + cur + (2 * oyl) * prow_bytes + (ox + g) * 16;
      f32x4 acc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      V8 xf[7];
#pragma unroll
      for (int ki = 0; ki < 7; ++ki) xf[ki] = *reinterpret_cast<const V8*>(xb + ki * prow_bytes);
#pragma unroll
      for (int ki = 0; ki < 7; ++ki)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = Elem16<E>::mma(wf[u][ki], xf[ki], acc[u]);
      // ---- the block's outputs: lane (g, li) holds z[pixel li][16 u + 4 g .. + 3] -> wave-private staging image [16][64]
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        V4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (E)acc[u][r];
        *reinterpret_cast<V4*>(stg + li * 128 + (((2 * u + (g >> 1)) ^ (li & 7)) << 4) + ((g & 1) << 3)) = o;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int r = ps * 8 + (lane >> 3), c = lane & 7;
        const V8 v = *reinterpret_cast<const V8*>(stg + r * 128 + ((c ^ (r & 7)) << 4));
        const int mm = blk * 16 + r;
        // (counted only where the store is certainly issued -- some lane is in range: an over-count would let the wait
        //  below pass with a patch request still in flight, an under-count only waits for a store as well)
        nstores += blk * 16 + ps * 8 < npix ? 1 : 0;
        if (mm < npix) {
          *reinterpret_cast<V8*>(yt + (int64_t)mm * 64 + c * 8) = v;
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
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (the staging image is rewritten by the next block)
      __builtin_amdgcn_wave_barrier();
    }
    // the next patch has landed: its requests were issued BEFORE this tile's stores and vmcnt retires in issue order, so the
    // stores (at most 14 per lane) may stay in flight
    if (more) wait_vm_upto(nstores);
    __syncthreads();
  }
  if (p.bn_partial) {                              // lanes with equal (lane & 7) hold the same 8 channels: fixed-order sum
    float* red = reinterpret_cast<float*>(smem);   // [2][512][8] = 32 KiB over the patch buffers (every read of them is done)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      red[(0 * kNW * 64 + threadIdx.x) * 8 + k] = bs[k];
      red[(1 * kNW * 64 + threadIdx.x) * 8 + k] = bq[k];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
      const int stat = threadIdx.x >> 6, ch = threadIdx.x & 63, c = ch >> 3, k = ch & 7;
      float t = 0.f;
      for (int j = 0; j < (kNW * 64) >> 3; ++j) t += red[(stat * kNW * 64 + j * 8 + c) * 8 + k];
      p.bn_partial[((int64_t)blockIdx.x * 2 + stat) * 64 + ch] = t;
    }
  }
}

int stem_plan(int H, int Wp, StemParams* q) {
  if (H < 2 || (H & 1) || Wp < 2 || Wp > 4090) return 0;
  const int Ho = H / 2, Wo = Wp;
  int R = kTilePix / Wo;
  if (R > Ho) R = Ho;
  if (R < 1) return 0;
  const int PPR = Wp + 3;
  int prows = 2 * R + 5;
  while (R > 1 && (prows * PPR + 63) / 64 > kNW * kMaxPieces) { --R; prows = 2 * R + 5; }
  const int npieces = (prows * PPR + 63) / 64;
  if (npieces > kNW * kMaxPieces || ((R * Wo + 15) / 16 + kNW - 1) / kNW > 7) return 0;
  q->H = H; q->Wp = Wp; q->Ho = Ho; q->Wo = Wo; q->R = R; q->PPR = PPR; q->prows = prows; q->npieces = npieces;
  q->patch_bytes = npieces * 1024;
  q->tiles_per_img = (Ho + R - 1) / R;
  q->magic_ppr = (unsigned)((((uint64_t)1 << 32) + (uint64_t)PPR - 1) / (uint64_t)PPR);
  q->magic_wo = (unsigned)((((uint64_t)1 << 32) + (uint64_t)Wo - 1) / (uint64_t)Wo);
  return 1;
}

// LDS of a launch: two patch buffers + the staging images, and at least the 32 KiB of the statistics scratch that overlays them
int stem_lds(const StemParams& q) {
  const int a = 2 * q.patch_bytes + kNW * kStage, b = 2 * kNW * 64 * 8 * 4;
  return a > b ? a : b;
}

int stem_grid(int64_t N, const StemParams& q) {
  const int64_t ntiles = N * q.tiles_per_img;
  return (int)(ntiles < dvt_num_cus() ? ntiles : dvt_num_cus());
}

template <typename E>
void stem_launch(const StemParams& p, int grid, int lds, hipStream_t st) {
  static DvtLdsAttr set;
  dvt_lds_attr(set, (const void*)conv_stem_kernel<E>, 160 * 1024);
  hipLaunchKernelGGL((conv_stem_kernel<E>), dim3(grid), dim3(kNW * 64), lds, st, p);
}

}  // namespace

extern "C" {

int dvt_conv_stem7_supported(int64_t N, int H, int Wp, int dtype) {
  StemParams q;
  return N > 0 && dvt_is_16bit(dtype) && stem_plan(H, Wp, &q) && N * q.tiles_per_img < ((int64_t)1 << 31) &&
                 N * H * Wp < ((int64_t)1 << 31) ? 1 : 0;
}

int64_t dvt_conv_stem7_stats_parts(int64_t N, int H, int Wp) {
  StemParams q;
  if (N <= 0 || !stem_plan(H, Wp, &q)) return 0;
  return stem_grid(N, q);                          // one partial row per workgroup of the persistent grid
}

int dvt_conv_stem7(const void* x_pairs, const void* w, int64_t ldw, void* y, float* stats_partial, int64_t N, int H, int Wp,
                   int dtype, dvt_stream_t stream) {
  DVT_REQUIRE(x_pairs && w && y && N >= 0 && H > 0 && Wp > 0 && ldw >= 224 && ldw % 8 == 0, "dvt_conv_stem7: bad arguments");
  DVT_REQUIRE(dvt_aligned16(x_pairs) && dvt_aligned16(w) && dvt_aligned16(y) && dvt_aligned16(stats_partial),
              "dvt_conv_stem7: buffers must be 16-byte aligned");
  if (N == 0) return DVT_OK;
  if (!dvt_conv_stem7_supported(N, H, Wp, dtype))
    DVT_UNSUPPORTED("dvt_conv_stem7: needs a 16-bit dtype, an even H and a tile of whole output rows whose pair patch fits 48 KiB");
  StemParams p{};
  stem_plan(H, Wp, &p);
  p.x = x_pairs; p.w = w; p.y = y; p.bn_partial = stats_partial; p.ldw = (int)ldw;
  p.ntiles = (int)(N * p.tiles_per_img);
  const int grid = stem_grid(N, p);
  const int lds = stem_lds(p);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DVT_F16) stem_launch<f16>(p, grid, lds, st);
  else stem_launch<bf16>(p, grid, lds, st);
  DVT_LAUNCH_CHECK("dvt_conv_stem7");
  return DVT_OK;
}

}  // extern "C"
