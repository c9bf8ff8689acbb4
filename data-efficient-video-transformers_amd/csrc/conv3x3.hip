// conv3x3.hip -- 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels from an LDS-resident halo patch
// (nn.Conv2d of custom_resnet.py:19-22 in layer 1 of ResNet-18, custom_resnet.py:109: four such layers per frame, forward
// and data gradient).
//
// The implicit GEMM (gemm256.hip) gathers every input element once per filter tap: at 64 output channels it is bound by
// that operand supply (9 x 103 MB L2 -> LDS per launch at 256 frames of 56^2, 450-540 TF/s).  Here a workgroup owns R whole
// output rows of one frame (R * W <= 256 pixels), stages the (R + 2) x (W + 2) x 64-channel input patch ONCE, zero border
// included, and all nine taps read their shifted windows of it from LDS; the 64 x 576 weights stay resident for the
// workgroup's whole tile sequence (persistent grid, one workgroup per CU), and the next tile's patch streams into a second
// buffer under the MFMAs of the current one.  160 KiB of LDS: 72 KiB weights + 2 x 44 KiB patches (W = 56, R = 4).
//
//   * 8 waves; wave w owns pixels [32 w, 32 w + 32) x all 64 output channels: acc[u][t] = W-fragment(u) x X-fragment(t),
//     lane (g, li) holds output channels 16 u + 4 g .. + 3 of pixel 16 t + li (the C^T convention of gemm256.hip);
//   * LDS images are lane-linear for the DMA, the bank swizzle (16-byte slot ^ (patch column or weight row & 7)) sits on the global source address
//     and on the fragment read; a pixel's 128 bytes and a weight row's 1152 bytes both alternate between the two 128-byte
//     halves of the 64 banks, and the XOR spreads the 16 lanes of a ds_read_b128 group over all slots;
//   * fragments of step s + 2 (a step = one tap x 32 channels: 2 + 4 reads, 8 MFMAs) are requested before the MFMAs of s;
//   * epilogue: per-wave 2 KiB staging inside the consumed patch buffer, whole 128-byte rows stored; optional per-wave
//     column sums / sums of squares of the stored values for the BatchNorm that follows (one partial row per wave).
#include "common.h"

namespace {

constexpr int kC = 64;                       // input channels = output channels
constexpr int kK = 9 * kC;                   // 576
constexpr int kWBytes = kC * kK * 2;         // 73,728
constexpr int kMaxPieces = 6;                // patch pieces (1 KiB) per wave: patch <= 44 KiB

struct Conv3Params {
  const void* x;        // [N, H, W, 64]
  const void* w;        // [64][576] k-major, k = tap * 64 + c
  void* y;              // [N, H, W, 64]
  float* bn_partial;    // [grid * 8][2][64] or nullptr
  const void* residual; // [N, H, W, 64] added to the output rows, or nullptr
  int N, H, W, R, tiles_per_img, ntiles, patch_bytes;
};

__device__ __attribute__((aligned(16))) unsigned int conv3_zero16[4] = {0u, 0u, 0u, 0u};

template <typename E>
__global__ __launch_bounds__(512) void conv3x3_c64_kernel(const Conv3Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using V8 = typename Elem16<E>::v8;
  using V4 = typename Elem16<E>::v4;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int PW = p.W + 2, npos = (p.R + 2) * PW;
  const int npieces = p.patch_bytes >> 10;
  char* wsm = smem;
  // the two patch buffers are smem + an offset: a `char* pb[2]` picked by it & 1 lost its LDS address space, and every x
  // fragment read of the main loop was a flat_load (counted on vmcnt AND lgkmcnt: each wait for a fragment also waited for
  // the next patch's DMA)
  auto patch = [&](int b) -> char* { return smem + kWBytes + b * p.patch_bytes; };
  const E* xg = (const E*)p.x;

  // ---- per-lane patch coordinates of this wave's pieces (fixed for the whole launch)
  int ppr[kMaxPieces], ppc[kMaxPieces], psl[kMaxPieces];
#pragma unroll
  for (int i = 0; i < kMaxPieces; ++i) {
    const int piece = wid + 8 * i;
    const int q = piece * 64 + lane, pos = q >> 3;
    ppr[i] = pos / PW;
    ppc[i] = pos - ppr[i] * PW;
    psl[i] = (piece < npieces && pos < npos) ? ((q & 7) ^ (ppc[i] & 7)) : -1;   // global 16-byte chunk of this LDS slot
  }
  auto load_patch = [&](int tile, char* dst) {
    const int n = tile / p.tiles_per_img, h0 = (tile - n * p.tiles_per_img) * p.R;
#pragma unroll
    for (int i = 0; i < kMaxPieces; ++i) {
      const int piece = wid + 8 * i;
      if (piece < npieces) {                     // wave-uniform
        const int h = h0 - 1 + ppr[i], w = ppc[i] - 1;
        const bool ok = psl[i] >= 0 && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
        const E* src = ok ? xg + ((int64_t)(n * p.H + h) * p.W + w) * kC + psl[i] * 8
                          : reinterpret_cast<const E*>(conv3_zero16);
        dvt_dma16(src, dst + piece * 1024);
      }
    }
  };

  // ---- prologue: weights (72 pieces, 9 per wave) and the first patch
  {
    const E* wg = (const E*)p.w;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int piece = wid * 9 + i;
      const int q = piece * 64 + lane, row = q / 72, c = q - row * 72;
      dvt_dma16(wg + (int64_t)row * kK + ((c & ~7) | ((c & 7) ^ (row & 7))) * 8, wsm + piece * 1024);
    }
  }
  int tile = blockIdx.x;
  if (tile < p.ntiles) load_patch(tile, patch(0));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- this lane's two pixels (t = 0, 1) inside a tile: patch position of tap (0, 0)
  const int npix = p.R * p.W;
  // byte offset of (this lane's pixel t, tap column kj, channel half kk) inside a patch, tap row 0; a tap row adds PW * 128.
  // The slot swizzle is by patch COLUMN, so it does not depend on the tap row and 12 offsets serve all 18 steps.
  int xo[2][3][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int m = wid * 32 + t * 16 + li;
    m = m < npix ? m : 0;                        // padding rows of the MFMA tile: computed on pixel 0, never stored
    const int r = m / p.W, c = m - r * p.W;
#pragma unroll
    for (int kj = 0; kj < 3; ++kj)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        xo[t][kj][kk] = (r * PW + c + kj) * 128 + (((kk * 4 + g) ^ ((c + kj) & 7)) << 4);
  }
  const int prow = PW * 128;
  // weight fragment bases: row 16 u + li, slot (kk * 4 + g) ^ (row & 7); a tap adds 128 bytes (an immediate offset)
  const char* wb[2][4];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int u = 0; u < 4; ++u) wb[kk][u] = wsm + (16 * u + li) * (kK * 2) + (((kk * 4 + g) ^ (li & 7)) << 4);

  // BatchNorm partial sums: taken from the staged rows (lane = 8 channels of one pixel, the values as stored), carried over
  // the workgroup's whole tile sequence and reduced over the 8 row lanes once, after the loop (one partial row per wave and
  // workgroup; per-tile trees over the accumulators' 16 pixel lanes cost 47 us of a 128 us launch).
  float bs[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int it = 0; tile < p.ntiles; ++it, tile += gridDim.x) {
    const int cur = kWBytes + (it & 1) * p.patch_bytes;
    if (tile + (int)gridDim.x < p.ntiles) load_patch(tile + gridDim.x, patch((it + 1) & 1));

    f32x4 acc[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    V8 xf[3][2], wf[3][4];
    auto rd = [&](int s, V8* xv, V8* wo) {         // step s = tap * 2 + kk
      const int tap = s >> 1, kk = s & 1, ki = tap / 3, kj = tap - ki * 3;
#pragma unroll
      for (int u = 0; u < 4; ++u) wo[u] = *reinterpret_cast<const V8*>(wb[kk][u] + tap * 128);
#pragma unroll
      for (int t = 0; t < 2; ++t) xv[t] = *reinterpret_cast<const V8*>(smem + cur + ki * prow + xo[t][kj][kk]);
    };
    rd(0, xf[0], wf[0]);
    rd(1, xf[1], wf[1]);
#pragma unroll
    for (int s = 0; s < 18; ++s) {
      // two steps ahead: a step's 8 MFMAs (128 cycles) are shorter than an LDS round trip under load
      if (s + 2 < 18) rd(s + 2, xf[(s + 2) % 3], wf[(s + 2) % 3]);
      __builtin_amdgcn_sched_barrier(0);          // keep the pipeline as written (the scheduler otherwise hoists every read)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[u][t] = Elem16<E>::mma(wf[s % 3][u], xf[s % 3][t], acc[u][t]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // the next patch has landed (requested a whole tile ago; the previous tile's stores are older still) ...
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ... and every wave is done reading the current one: it becomes the staging area of the epilogue
    __builtin_amdgcn_s_barrier();

    const int n = tile / p.tiles_per_img, h0 = (tile - n * p.tiles_per_img) * p.R;
    const int rows_ok = min(p.R, p.H - h0);
    const int valid = rows_ok * p.W;             // pixels of this tile that exist
    E* yt = (E*)p.y + ((int64_t)(n * p.H + h0) * p.W) * kC;
    const E* rt = p.residual ? (const E*)p.residual + ((int64_t)(n * p.H + h0) * p.W) * kC : nullptr;
    char* stg = smem + cur + wid * 2048;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int m0 = wid * 32 + t * 16;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        V4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (E)acc[u][t][r];
        // staged row li = pixel, 16-byte chunk (2 u + (g >> 1)) ^ ((li >> 1) & 7), 8-byte half g & 1
        *reinterpret_cast<V4*>(stg + li * 128 + (((u * 2 + (g >> 1)) ^ ((li >> 1) & 7)) << 4) + (g & 1) * 8) = o;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int r = ps * 8 + (lane >> 3), c = lane & 7;
        V8 v = *reinterpret_cast<const V8*>(stg + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
        if (m0 + r < valid) {
          if (rt) {                               // a second gradient path joining this one (dvt_conv3x3_c64's residual)
            const V8 rv = *reinterpret_cast<const V8*>(rt + (int64_t)(m0 + r) * kC + c * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (E)((float)v[k] + (float)rv[k]);
          }
          *reinterpret_cast<V8*>(yt + (int64_t)(m0 + r) * kC + c * 8) = v;
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
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
    // every wave is done with the staging area before the next iteration's DMA overwrites it
    __builtin_amdgcn_s_barrier();
  }
  if (p.bn_partial) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        bs[k] += __shfl_xor(bs[k], o, 64);
        bq[k] += __shfl_xor(bq[k], o, 64);
      }
    if ((lane >> 3) == 0) {                      // lane c holds channels 8 c .. 8 c + 7
      float* pr = p.bn_partial + ((int64_t)blockIdx.x * 8 + wid) * 2 * kC + (lane & 7) * 8;
      *reinterpret_cast<f32x4*>(pr) = f32x4{bs[0], bs[1], bs[2], bs[3]};
      *reinterpret_cast<f32x4*>(pr + 4) = f32x4{bs[4], bs[5], bs[6], bs[7]};
      *reinterpret_cast<f32x4*>(pr + kC) = f32x4{bq[0], bq[1], bq[2], bq[3]};
      *reinterpret_cast<f32x4*>(pr + kC + 4) = f32x4{bq[4], bq[5], bq[6], bq[7]};
    }
  }
}

// rows per tile and patch size (rounded up to 1 KiB pieces); 0 when two patches do not fit beside the weights
int plan(int H, int W, int* R, int* patch_bytes) {
  if (W <= 0 || H <= 0 || W > 256) return 0;
  int r = 256 / W;
  if (r > H) r = H;
  if (r < 1) return 0;
  int pbytes = (((r + 2) * (W + 2) * 128) + 1023) & ~1023;
  if (pbytes < 8 * 2048) pbytes = 8 * 2048;      // a consumed patch buffer is also the epilogue's staging area (2 KiB per wave)
  if (kWBytes + 2 * pbytes > 160 * 1024 || (pbytes >> 10) > 8 * kMaxPieces) return 0;
  *R = r;
  *patch_bytes = pbytes;
  return 1;
}

}  // namespace

extern "C" {

int dvt_conv3x3_c64_supported(int64_t N, int H, int W, int dtype) {
  int R, pb;
  return N > 0 && dvt_is_16bit(dtype) && plan(H, W, &R, &pb) && N * H * W < ((int64_t)1 << 31) ? 1 : 0;
}

int64_t dvt_conv3x3_c64_stats_parts(int64_t N, int H, int W) {
  int R, pb;
  if (!plan(H, W, &R, &pb)) return 0;
  const int64_t ntiles = N * dvt_cdiv(H, R);
  return (ntiles < dvt_num_cus() ? ntiles : dvt_num_cus()) * 8;      // one partial row per wave of the persistent grid
}

int dvt_conv3x3_c64(const void* x, const void* w, void* y, float* stats_partial, const void* residual, int64_t N, int H, int W,
                    int dtype, dvt_stream_t stream) {
  DVT_REQUIRE(x && w && y && N >= 0 && H > 0 && W > 0, "dvt_conv3x3_c64: bad arguments");
  DVT_REQUIRE(dvt_aligned16(x) && dvt_aligned16(w) && dvt_aligned16(y) && dvt_aligned16(stats_partial) && dvt_aligned16(residual),
              "dvt_conv3x3_c64: buffers must be 16-byte aligned");
  if (N == 0) return DVT_OK;
  Conv3Params p;
  if (!dvt_conv3x3_c64_supported(N, H, W, dtype))
    DVT_UNSUPPORTED("dvt_conv3x3_c64: needs a 16-bit dtype and (R + 2)(W + 2) * 128 B <= 44 KiB with R = 256 / W rows per tile");
  plan(H, W, &p.R, &p.patch_bytes);
  p.x = x; p.w = w; p.y = y; p.bn_partial = stats_partial; p.residual = residual;
  p.N = (int)N; p.H = H; p.W = W;
  p.tiles_per_img = (int)dvt_cdiv(H, p.R);
  p.ntiles = (int)(N * p.tiles_per_img);
  const int lds = kWBytes + 2 * p.patch_bytes;
  const int grid = p.ntiles < dvt_num_cus() ? p.ntiles : dvt_num_cus();
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DVT_BF16) {
    static DvtLdsAttr set;
    dvt_lds_attr(set, (const void*)conv3x3_c64_kernel<bf16>, 160 * 1024);
    hipLaunchKernelGGL((conv3x3_c64_kernel<bf16>), dim3(grid), dim3(512), lds, st, p);
  } else {
    static DvtLdsAttr set;
    dvt_lds_attr(set, (const void*)conv3x3_c64_kernel<f16>, 160 * 1024);
    hipLaunchKernelGGL((conv3x3_c64_kernel<f16>), dim3(grid), dim3(512), lds, st, p);
  }
  DVT_LAUNCH_CHECK("dvt_conv3x3_c64");
  return DVT_OK;
}

}  // extern "C"
