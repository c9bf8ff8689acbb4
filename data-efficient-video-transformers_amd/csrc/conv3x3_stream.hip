// conv3x3_stream.hip -- 3x3 / stride 1 / pad 1 convolutions of the R(2+1)D-18 layer-1 spatial pair (video_resnet.py:
// Conv2Plus1D, 64 -> 144 mid planes forward, 144 -> 64 as the data gradient) from LDS halo patches with STREAMED weights.
//
// conv3x3.hip keeps the 64 x 576 weights resident; at 144 channels on either side they are 162 KiB and cannot be.  The
// implicit GEMM (gemm256.hip) then pays 9 gathers of the input per output pixel plus, at 144 output channels, a 256-wide
// tile column that is 44 % padding (358 us forward, 535 us data gradient at 336 frames of 56^2: both bound by the L2 -> LDS
// operand supply).  Here a workgroup owns R whole output rows of one frame (R * W <= 224 pixels = 7 compute waves of 32),
// the input patch is staged once per tile (per 48-channel chunk at 144 input channels), and the weights stream through a
// ring of three 18 KiB stages (one tap x 144 x 64, or one tap row x 64 x 48) that every tile re-reads from L2:
// 0.94 KiB (forward) / 1.2 KiB (data gradient) of L2 -> LDS traffic per output pixel instead of 2.3 / 3.2.
//
//   * Where the time goes (tools/dev/stream_probe.py, isolated launches at 336 frames of 56^2, parts switched off): the whole
//     kernel 236-248 us forward / 227 us data gradient; without the MFMAs 156 / 167 us; without MFMAs and fragment reads
//     147 / 149 us; that skeleton with half the weight pieces 132 / 136 us; without any steady-state DMA (barriers and the
//     epilogue's stores alone) 57 / 25 us = the output at 5.3 TB/s.  The DMA stream costs 90-125 us whatever its size, and
//     the MFMAs add their own 60-85 us on top instead of hiding under it.
//   * Measured and dropped (in-step times; run-to-run spread +-4 %): (a) the seven compute waves carrying the weight stream
//     (3 or 2 pieces of every stage each, counted vmcnt with the epilogue's stores on the same counter), producer only the
//     patches: 174 -> 182 us forward, 218 -> 228 us data gradient; (b) additionally the patch requested by the compute waves
//     once per tile, no producer at all (a wave's memory operations return in order, so on the producer's counter every
//     stage's wait for its L2-resident weights also waited for the HBM patch pieces in front of them): 187 us forward,
//     99 us the (3, 1) form (102-105); (c) the data gradient's weights through the compute waves' REGISTERS
//     (global_load_dwordx4 behind one barrier, ds_write_b128 before the next; the producer only the patch chunks): 230 us;
//     (d) fragments two steps ahead in the 96-byte form (three register sets): 220 us; (e) a ring of four weight stages
//     there: 241 us.  Neither the loader wave's in-flight window, nor the ordering of its counter, nor the LDS-DMA path as
//     such, nor the LDS round trip, nor the ring depth is what bounds these kernels; the variants are equivalent within the
//     spread and the simplest one stays.
//   * wave 7 is the PRODUCER: it issues every LDS-DMA of the workgroup (weight stage g + 3 and a group of pieces of the next
//     patch chunk after barrier g + 1) and is the only wave that counts vmcnt, with compile-time batch sizes; the compute
//     waves never wait on memory, only on the one s_barrier per stage the producer joins once the stage's bytes have landed;
//   * a compute wave owns 32 pixels x all output channels: acc[u][t] = W-fragment(u) x X-fragment(t) (C^T convention of
//     gemm256.hip); the barrier that opens stage g + 1 sits before the LAST step's MFMAs of stage g, after that wave's
//     last LDS read of stage g has returned, so the first fragments of g + 1 are fetched under those MFMAs;
//   * LDS images are lane-linear for the DMA; the bank swizzle lives on the global source address and on the fragment read:
//     128-byte pixels / weight rows XOR the 16-byte slot with (column or row & 7) as conv3x3.hip does; 96-byte ones
//     (48-channel chunks: 6 slots) keep slots 0 - 3 where they are and swap slots 4 / 5 in every other group of 8 pixels /
//     rows.  Why (round 6, SQ_LDS_BANK_CONFLICT 3.6e7 of 7.7e7 LDS cycles per launch, tools/dev/win_pmc.sh): a ds_read_b128 is
//     served in the lane groups {0-3, 12-15, 20-27} ... (MI355X_MICROARCH.md, LDS), not in runs of 16 lanes; for those groups
//     16 consecutive pixels x 4 slots at a pitch of 6 slots fall on 16 different 16-byte bank slots WITHOUT any swizzle, and
//     the round-4 one (slot bit 0 flipped by slot bit 4, meant for runs of 16 lanes) made every such read two-way conflicted
//     (7.4 / 8 LDS cycles per x / w fragment instead of 4).  The 8-byte reads of slots 4 / 5 (two groups of 32 lanes) do
//     need pixels p and p + 8 apart -- hence the swap of just those two slots; patch rows are padded to a multiple of 32
//     slots so that a tap row keeps the banks;
//   * the 48-channel chunk is one 16x16x32 and one 16x16x16 MFMA step;
//   * epilogue: 16 pixels at a time through LDS, whole pixel rows stored; 144-wide outputs carry the column sums / sums of
//     squares of the stored values for the BatchNorm that follows (one partial row per compute wave and workgroup), 64-wide
//     outputs the optional second gradient path (residual) of the data gradient.
#include "common.h"

namespace {

template <int CI_, int CO_, int NTAP_> struct SC;
// PPG pieces (1 KiB) per patch group, NG groups per chunk; the groups of the NEXT chunk ride the batches at chunk-relative
// stage positions GPOS .. SPC - 1 and the last group the batch of the chunk's own first stage.
template <> struct SC<64, 144, 9> { enum { CK = 64, NCH = 1, TPS = 1, PPG = 7, GPOS = 3, NG = 7, STG_DEDICATED = 0 }; };
template <> struct SC<144, 64, 9> { enum { CK = 48, NCH = 3, TPS = 3, PPG = 17, GPOS = 2, NG = 2, STG_DEDICATED = 1 }; };
// Layer 2 of R(2+1)D-18 (video_resnet.py: Conv2Plus1D 128 -> 288 mid planes -> 128 on 28 x 28 maps; round 6): the same two
// kernels with more input chunks per tile, launched once per 144- / 64-wide group of the output channels (the group's rows
// of the weights, its columns of y / the residual / the statistics rows: StreamParams.ldy, ldp).
template <> struct SC<128, 144, 9> { enum { CK = 64, NCH = 2, TPS = 1, PPG = 7, GPOS = 3, NG = 7, STG_DEDICATED = 0 }; };
template <> struct SC<288, 64, 9> { enum { CK = 48, NCH = 6, TPS = 3, PPG = 17, GPOS = 2, NG = 2, STG_DEDICATED = 1 }; };
// NTAP = 3: a (3, 1) filter over rows `row_pitch` pixels apart (the temporal half of Conv2Plus1D over the [T, H*W] view of a clip,
// as the data gradient 64 -> 144): three stages per tile, no column halo; the whole patch rides the batch of the tile's first
// stage (62 DMA instructions, one below the vmcnt range).
template <> struct SC<64, 144, 3> { enum { CK = 64, NCH = 1, TPS = 1, PPG = 45, GPOS = 3, NG = 1, STG_DEDICATED = 0 }; };

constexpr int kWStage = 18 * 1024;           // one weight stage: TPS taps x CO rows x CK channels
constexpr int stg_pitch(int co) { return co == 64 ? 128 : co * 2 + 8; }      // bytes between the pixels of the epilogue staging
constexpr int kNC = 7;                       // compute waves

struct StreamParams {
  const void* x;         // [N, H, W, CI]
  const void* w;         // [CO][9 * CI] k-major, k = tap * CI + c
  void* y;               // [N, H, W, CO]
  float* bn_partial;     // [grid * 7][2][CO] or nullptr
  const void* residual;  // [N, H, W, CO] added to the output rows, or nullptr
  int N, H, W, R, tiles_per_img, ntiles;   // N images of H rows x W columns (the (3, 1) form: image = one column segment of a clip)
  int PW, SPR;           // patch columns, 16-byte slots per patch row
  unsigned magic;        // ceil(2^32 / PW) (128-byte pixels) or ceil(2^32 / SPR) (96-byte pixels)
  unsigned magic_w;      // ceil(2^32 / W)
  int hb;                // images per outer block: pixel 0 of image n = (n / hb) * pitch_n + (n % hb) * pitch_h
  int64_t pitch_n;
  int pitch_h, row_pitch;   // pixels between the image's rows (W for a dense NHWC frame)
  int ldy, ldp;             // elements between the pixel rows of y / z / the residual (CO, or the full width when this launch
                            // writes one channel group of a wider map); floats between the two statistics rows of a partial
  // BatchNorm backward of the layer in FRONT of this data gradient, fused (MODE 1 / 2 of the kernel, dvt_conv3x1_stream_bn_bwd):
  const void* bz;           // [.., CO] the convolution output z that BatchNorm normalised (same pixel layout as y)
  const float *bmean, *binvstd, *bgamma, *bbeta;
  const float* bloc;        // MODE 2: [2][CO] this launch pair's own sum dz * xhat, sum dz (bn_finalize)
  float inv_rows;
  int brelu, btraining;
};

__device__ __attribute__((aligned(16))) unsigned int conv3s_zero16[4] = {0u, 0u, 0u, 0u};

template <typename E> struct Mma16;
template <> struct Mma16<bf16> {
  static __device__ __forceinline__ f32x4 mma(bf16x4 a, bf16x4 b, f32x4 c) {
    typedef __attribute__((ext_vector_type(4))) short s4;
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s4, a), __builtin_bit_cast(s4, b), c, 0, 0, 0);
  }
};
template <> struct Mma16<f16> {
  static __device__ __forceinline__ f32x4 mma(f16x4 a, f16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
  }
};

__device__ __forceinline__ void wait_vm(int n) {
  switch (n) {
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
    case 35: asm volatile("s_waitcnt vmcnt(35)" ::: "memory"); break;
    case 63: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// MODE 0: the convolution.  MODE 1 / 2 (the 144-wide outputs only): the convolution is the DATA GRADIENT d of a layer whose
// input was relu(BatchNorm(z)) and the BatchNorm backward runs in its epilogue instead of in two passes over a stored d:
//   MODE 1  nothing is stored; per channel sum dz and sum dz * xhat (dz = d under the ReLU mask recomputed from z) leave as
//           partial rows [grid * 7][2][CO] (the layout bn_finalize sums);
//   MODE 2  the same d again (this layer is cheap: 108 MFMAs per tile against 64 KiB of output), and
//           gamma * invstd * (dz - sum dz / rows - xhat * sum dz xhat / rows) is what gets stored.
// The z rows of a tile are requested before its main loop and consumed in its epilogue.
template <typename E, int CI, int CO, int NTAP, int MODE = 0>
__global__ __launch_bounds__(512) void conv3x3_stream_kernel(const StreamParams p) {
  typedef SC<CI, CO, NTAP> C;
  using V8 = typename Elem16<E>::v8;
  using V4 = typename Elem16<E>::v4;
  constexpr int CK = C::CK, NCH = C::NCH, TPS = C::TPS, SPC = NTAP / TPS, PPG = C::PPG, GPOS = C::GPOS, NG = C::NG;
  constexpr int SPP = CK / 8;                  // 16-byte slots per pixel / weight row
  constexpr int NB = CO / 16;                  // output-channel blocks
  constexpr int kPatch = NG * PPG * 1024;      // one patch (chunk) buffer
  constexpr int NSTEP = CK == 64 ? 2 : 6;      // fragment steps per stage
  // epilogue staging per wave: 16 pixels.  The 144-wide rows are 296 bytes apart, not 288: lanes li = 0 .. 15 of a ds_write_b64
  // group then fall on 16 different 8-byte bank slots (288 = 32 mod 128 put four lanes on each: 16 LDS cycles per store
  // instead of 4, 18 stores per wave and tile -- all of the 64 -> 144 kernels' counted bank conflicts); the rows are read
  // back as two 8-byte halves per 16-byte chunk (the same LDS cycles as one ds_read_b128, which needs 16-byte alignment)
  constexpr int kStgPitch = stg_pitch(CO);
  constexpr int kStgWave = 16 * kStgPitch;
  constexpr int NST = NCH * SPC;               // stages per tile
  constexpr int HALO = NTAP == 9 ? 1 : 0;      // zero columns either side of the patch rows
  static_assert(NST % 3 == 0 && NST >= 3 && TPS * CO * CK * 2 == kWStage && (NSTEP & 1) == 0, "stage geometry");
  static_assert(NTAP == 9 || CK == 64, "the (3, 1) form exists for 128-byte pixels");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wring = smem;
  char* const pbuf = smem + 3 * kWStage;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int grid = gridDim.x;

  if (wid == kNC) {
    // ================================================================ producer
    // Per-lane constants of every DMA slot this wave ever fills (the LDS images do not depend on the tile):
    //   woff[i]  element offset of weight piece i's 16 bytes inside (row-major w, tap 0 / chunk 0)
    //   poff[q]  patch piece q: bits 0..23 element offset from (patch row 0, column -HALO) of the image; bits 24..30 the patch
    //            row; bit 31 = never loaded (padding slot, or a column outside the frame)
    // (one wave issues every DMA of the workgroup and a wave64 VALU instruction costs four cycles: at 35 pieces per stage
    // each instruction of the per-piece address arithmetic is ~5 % of a stage)
    const E* xg = (const E*)p.x;
    const E* wg = (const E*)p.w;
    int woff[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      const int sl = i * 64 + lane;
      if constexpr (CK == 64) {
        const int row = sl >> 3, cs = (sl & 7) ^ (row & 7);
        woff[i] = row * (NTAP * CI) + cs * 8;
      } else {
        const int ra = sl / 6, k6 = sl - ra * 6, kj = ra >> 6, co = ra & 63;
        const int cs = k6 < 4 ? k6 : 4 + ((k6 - 4) ^ ((ra >> 3) & 1));     // (slots 4 / 5 swap in every other 8 rows: see rd)
        woff[i] = co * (NTAP * CI) + kj * CI + cs * 8;
      }
    }
    unsigned poff[NG * PPG];
#pragma unroll
    for (int q = 0; q < NG * PPG; ++q) {
      const int sl = q * 64 + lane;
      int pr, pc, cs;
      bool v;
      if constexpr (CK == 64) {
        const int pixel = sl >> 3;
        pr = (int)__umulhi((unsigned)pixel, p.magic);
        pc = pixel - pr * p.PW;
        cs = (sl & 7) ^ (pc & 7);
        v = pr < p.R + 2;
      } else {
        pr = (int)__umulhi((unsigned)sl, p.magic);
        const int within = sl - pr * p.SPR;
        pc = within / 6;
        const int k6 = within - pc * 6;
        cs = k6 < 4 ? k6 : 4 + ((k6 - 4) ^ ((pc >> 3) & 1));
        v = pr < p.R + 2 && pc < p.PW;
      }
      v = v && (unsigned)(pc - HALO) < (unsigned)p.W;
      poff[q] = v ? (unsigned)(pr * p.row_pitch * CI + pc * CI + cs * 8) | ((unsigned)pr << 24) : 0x80000000u;
    }
    auto issue_weights = [&](int sg) __attribute__((always_inline)) {             // stage position sg of a tile
      char* dst = wring + (sg % 3) * kWStage;
      // (CK == 64: stage = one tap of one 64-channel chunk; CK == 48: a tap row of one 48-channel chunk)
      const E* base = CK == 64 ? wg + (sg % SPC) * CI + (sg / SPC) * CK : wg + (sg % 3) * 3 * CI + (sg / 3) * CK;
#pragma unroll
      for (int i = 0; i < 18; ++i) {
        int o = woff[i];
        asm volatile("" : "+v"(o));                 // (keeps the 64-bit sums out of loop-invariant registers: 18 x 9 pairs)
        dvt_dma16(base + o, dst + i * 1024);
      }
    };
    auto issue_group = [&](int tile, int cc, int grp, char* dst) __attribute__((always_inline)) {   // grp: compile-time at every call site
      const int n = tile / p.tiles_per_img, h0 = (tile - n * p.tiles_per_img) * p.R;
      const int nq = n / p.hb;
      // element (row h0 - 1, column -HALO, chunk cc) of the image
      const E* base = xg + (nq * p.pitch_n + (int64_t)(n - nq * p.hb) * p.pitch_h + (int64_t)(h0 - 1) * p.row_pitch - HALO) * CI + cc * CK;
      const bool interior = h0 >= 1 && h0 + p.R + 1 <= p.H;     // every patch row lies inside the image
      if (interior) {
#pragma unroll
        for (int i = 0; i < PPG; ++i) {
          unsigned pk = poff[grp * PPG + i];
          asm volatile("" : "+v"(pk));              // (same: the unpacked fields stay temporaries)
          const E* src = (int)pk >= 0 ? base + (pk & 0xFFFFFF) : reinterpret_cast<const E*>(conv3s_zero16);
          dvt_dma16(src, dst + (grp * PPG + i) * 1024);
        }
      } else {
#pragma unroll
        for (int i = 0; i < PPG; ++i) {
          unsigned pk = poff[grp * PPG + i];
          asm volatile("" : "+v"(pk));
          const int h = h0 - 1 + (int)((pk >> 24) & 0x7F);
          const bool ok = (int)pk >= 0 && (unsigned)h < (unsigned)p.H;
          const E* src = ok ? base + (pk & 0xFFFFFF) : reinterpret_cast<const E*>(conv3s_zero16);
          dvt_dma16(src, dst + (grp * PPG + i) * 1024);
        }
      }
    };
    // the batch of the stage at chunk cc (runtime), chunk-relative position pos (compile-time) of tile ta (tb: the tile after)
    auto issue_batch = [&](int ita, int ta, int tb, int cc, int pos) __attribute__((always_inline)) {
      issue_weights(cc * SPC + pos);
      if (pos == 0) {
        issue_group(ta, cc, NG - 1, pbuf + ((ita * NCH + cc) & 1) * kPatch);
      } else if (pos >= GPOS) {
        const bool same = cc + 1 < NCH;
        issue_group(same ? ta : tb, same ? cc + 1 : 0, pos - GPOS, pbuf + ((ita * NCH + cc + 1) & 1) * kPatch);
      }
    };
    auto batch_size = [](int pos) __attribute__((always_inline)) { return 18 + ((pos == 0 || pos >= GPOS) ? PPG : 0); };

    int tile = blockIdx.x;
    int t1 = tile + grid < p.ntiles ? tile + grid : tile;
#pragma unroll
    for (int gr = 0; gr < NG - 1; ++gr) issue_group(tile, 0, gr, pbuf);
    issue_batch(0, tile, t1, 0, 0);
    issue_batch(0, tile, t1, 1 / SPC, 1 % SPC);
    issue_batch(0, tile, t1, 2 / SPC, 2 % SPC);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                   // B_0
    for (int it = 0; tile < p.ntiles; ++it, tile += grid) {
      t1 = tile + grid < p.ntiles ? tile + grid : tile;
      const int t2 = t1 + grid < p.ntiles ? t1 + grid : t1;
#pragma unroll 1
      for (int cc = 0; cc < NCH; ++cc) {
#pragma unroll
        for (int pos = 0; pos < SPC; ++pos) {
          wait_vm(batch_size((pos + 2) % SPC));                     // batch g + 1 has landed, g + 2 may be in flight
          __builtin_amdgcn_s_barrier();                             // B_{g+1}: stage g + 1 open, stage g's buffers free
          // batch g + 3: chunk cc + (pos + 3) / SPC, position (pos + 3) % SPC -- of the next tile past this one's chunks
          int c3 = cc + (pos + 3) / SPC;
          if (c3 < NCH) issue_batch(it, tile, t1, c3, (pos + 3) % SPC);
          else issue_batch(it + 1, t1, t2, c3 - NCH, (pos + 3) % SPC);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ================================================================== compute waves
  const int PW = p.PW, npix = p.R * p.W;
  const int prow = p.SPR * 16;
  int xo[2][3][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int m = wid * 32 + t * 16 + li;
    m = m < npix ? m : 0;                          // padding rows of the MFMA tile: computed on pixel 0, never stored
    const int r = m / p.W, c = m - r * p.W;
#pragma unroll
    for (int kj = 0; kj < 3; ++kj) {
      if constexpr (CK == 64) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) xo[t][kj][kk] = (r * PW + c + kj) * 128 + (((kk * 4 + g) ^ ((c + kj) & 7)) << 4);
      } else {
        const int pc = c + kj;
        xo[t][kj][0] = (r * p.SPR + pc * 6 + g) << 4;
        xo[t][kj][1] = ((r * p.SPR + pc * 6 + 4 + ((g >> 1) ^ ((pc >> 3) & 1))) << 4) + (g & 1) * 8;
      }
    }
  }
  int wo[2];
  if constexpr (CK == 64) {
    wo[0] = li * 128 + (((0 + g) ^ (li & 7)) << 4);
    wo[1] = li * 128 + (((4 + g) ^ (li & 7)) << 4);
  } else {
    wo[0] = (li * 6 + g) << 4;
    wo[1] = ((li * 6 + 4 + ((g >> 1) ^ ((li >> 3) & 1))) << 4) + (g & 1) * 8;
  }

  V8 wf8[CK == 64 ? 2 : 1][NB], xf8[CK == 64 ? 2 : 1][2];
  V4 wf4[NB], xf4[2];
  (void)wf4; (void)xf4;
  // fragments of step j of stage (patch buffer pb, position sg)
  auto rd = [&](const char* pb, int sg, int j) __attribute__((always_inline)) {
    const char* wb = wring + (sg % 3) * kWStage;
    if constexpr (CK == 64) {
      const int tp = sg % SPC;
      const int ki = NTAP == 9 ? tp / 3 : tp, kj = NTAP == 9 ? tp % 3 : 0;
#pragma unroll
      for (int u = 0; u < NB; ++u) wf8[j][u] = *reinterpret_cast<const V8*>(wb + wo[j] + u * 2048);
#pragma unroll
      for (int t = 0; t < 2; ++t) xf8[j][t] = *reinterpret_cast<const V8*>(pb + ki * prow + xo[t][kj][j]);
    } else {
      const int ki = sg % 3, kj = j >> 1;
      if ((j & 1) == 0) {
#pragma unroll
        for (int u = 0; u < NB; ++u) wf8[0][u] = *reinterpret_cast<const V8*>(wb + wo[0] + (kj * 64 + 16 * u) * 96);
#pragma unroll
        for (int t = 0; t < 2; ++t) xf8[0][t] = *reinterpret_cast<const V8*>(pb + ki * prow + xo[t][kj][0]);
      } else {
#pragma unroll
        for (int u = 0; u < NB; ++u) wf4[u] = *reinterpret_cast<const V4*>(wb + wo[1] + (kj * 64 + 16 * u) * 96);
#pragma unroll
        for (int t = 0; t < 2; ++t) xf4[t] = *reinterpret_cast<const V4*>(pb + ki * prow + xo[t][kj][1]);
      }
    }
  };
  auto patch_of = [&](int it, int sg) __attribute__((always_inline)) -> const char* { return pbuf + ((it * NCH + sg / SPC) & 1) * kPatch; };

  float bs[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  (void)bs; (void)bq;
  // ---- fused BatchNorm backward.  Per channel (a table in LDS behind the patch buffers, filled before B_0):
  //   sc, sh   the folded forward affine (BnAffine::init, conv.hip): relu mask = z * sc + sh > 0, the decision every route takes
  //   MODE 1:  is, mis   xhat = z * is + mis
  //   MODE 2:  A, B, C0  gamma * invstd * (dz - sum dz / rows - xhat * sum dz xhat / rows) = A dz + B z + C0
  // A lane stores the same 16-byte chunk (8 channels) of every pixel row it handles; its constants are read from the table at
  // the start of each tile's epilogue (the main loop has no registers to spare for them).
  constexpr int kCH = CO / 8, kPPP = 64 / kCH, kPS = (16 + kPPP - 1) / kPPP;   // chunks per pixel, pixels per pass, passes per 16 pixels
  constexpr int kNTab = MODE == 2 ? 5 : 4;
  float* const btab = reinterpret_cast<float*>(smem + 3 * kWStage + 2 * kPatch + (C::STG_DEDICATED ? kNC * kStgWave : 0));
  const int r3e = lane / kCH, c18e = lane - r3e * kCH;
  if constexpr (MODE != 0) {
    static_assert(CO != 64, "the fused BatchNorm backward rides the 144-wide epilogue");
    for (int c = threadIdx.x; c < CO; c += kNC * 64) {
      const float mu = p.bmean[c], is = p.binvstd[c], gmm = p.bgamma[c];
      const float sc = is * gmm;
      btab[c] = sc;
      btab[CO + c] = fmaf(-mu, sc, p.bbeta[c]);
      if (MODE == 1) {
        btab[2 * CO + c] = is;
        btab[3 * CO + c] = -mu * is;
      } else {
        const float kb = p.btraining ? p.bloc[CO + c] * p.inv_rows : 0.f;      // sum dz / rows
        const float kg = p.btraining ? p.bloc[c] * p.inv_rows : 0.f;           // sum dz xhat / rows
        const float gi = gmm * is;
        btab[2 * CO + c] = gi;
        btab[3 * CO + c] = -gi * kg * is;
        btab[4 * CO + c] = gi * (kg * is * mu - kb);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // pixel offset (from the tile's first pixel) of the row this lane stores in pass ps of half t; -1: none
  auto out_pix = [&](int t, int ps, int valid) __attribute__((always_inline)) -> int64_t {
    const int px = ps * kPPP + r3e;
    const int mpx0 = wid * 32 + t * 16 + px;
    if (!(r3e < kPPP && px < 16 && mpx0 < valid)) return -1;
    int mpx = mpx0;
    asm volatile("" : "+v"(mpx));
    int64_t po = mpx;
    if (NTAP != 9) {
      const int r = (int)__umulhi((unsigned)mpx, p.magic_w);
      po = (int64_t)r * p.row_pitch + (mpx - r * p.W);
    }
    return po;
  };
  V8 zq[MODE != 0 ? 2 : 1][MODE != 0 ? kPS : 1];
  (void)zq;
  // z rows of half t of a tile, requested at the start of its epilogue (under the staging of half 0; holding them across the
  // main loop -- 24 registers per half -- spilled, and a spill reload in the producer's path waits for every DMA in flight)
  auto request_z = [&](int tile_, int t) __attribute__((always_inline)) {
    const int n = tile_ / p.tiles_per_img, h0 = (tile_ - n * p.tiles_per_img) * p.R;
    const int valid = min(p.R, p.H - h0) * p.W;
    const int nq = n / p.hb;
    const int64_t pix0 = nq * p.pitch_n + (int64_t)(n - nq * p.hb) * p.pitch_h + (int64_t)h0 * p.row_pitch;
    const E* zt = (const E*)p.bz + pix0 * p.ldy;
#pragma unroll
    for (int ps = 0; ps < kPS; ++ps) {
      const int64_t po = out_pix(t, ps, valid);
      zq[t][ps] = *reinterpret_cast<const V8*>(zt + (po >= 0 ? po : 0) * p.ldy + c18e * 8);
    }
  };
  int tile = blockIdx.x;
  __builtin_amdgcn_s_barrier();                                     // B_0
  rd(patch_of(0, 0), 0, 0);
  for (int it = 0; tile < p.ntiles; ++it, tile += grid) {
    f32x4 acc[NB][2];
#pragma unroll
    for (int u = 0; u < NB; ++u)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sg = 0; sg < NST; ++sg) {
      const char* pb = patch_of(it, sg);
#pragma unroll
      for (int j = 0; j < NSTEP; ++j) {
        if (j + 1 < NSTEP) {
          rd(pb, sg, j + 1);
        } else {
          // this wave's last read of stage g has returned: open stage g + 1 and fetch its first fragments under the MFMAs
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();                             // B_{g+1}
          if (sg + 1 < NST) rd(patch_of(it, sg + 1), sg + 1, 0);
          else rd(patch_of(it + 1, 0), 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (CK == 64) {
#pragma unroll
          for (int u = 0; u < NB; ++u)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[u][t] = Elem16<E>::mma(wf8[j][u], xf8[j][t], acc[u][t]);
        } else if ((j & 1) == 0) {
#pragma unroll
          for (int u = 0; u < NB; ++u)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[u][t] = Elem16<E>::mma(wf8[0][u], xf8[0][t], acc[u][t]);
        } else {
#pragma unroll
          for (int u = 0; u < NB; ++u)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[u][t] = Mma16<E>::mma(wf4[u], xf4[t], acc[u][t]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- epilogue.  Staging: the dedicated region, or the patch buffer this tile consumed (every wave is past B of the next
    // tile's first stage, i.e. past its last read of it; the producer refills it only after the barrier inside that stage).
    const int n = tile / p.tiles_per_img, h0 = (tile - n * p.tiles_per_img) * p.R;
    const int rows_ok = min(p.R, p.H - h0);
    const int valid = rows_ok * p.W;
    const int nq = n / p.hb;
    const int64_t pix0 = nq * p.pitch_n + (int64_t)(n - nq * p.hb) * p.pitch_h + (int64_t)h0 * p.row_pitch;   // the tile's first pixel
    E* yt = (E*)p.y + pix0 * p.ldy;
    char* stg = (C::STG_DEDICATED ? pbuf + 2 * kPatch : pbuf + ((it * NCH + NCH - 1) & 1) * kPatch) + wid * kStgWave;
    float bk[MODE != 0 ? kNTab : 1][8];
    (void)bk;
    if constexpr (MODE != 0) {
      request_z(tile, 0);
      request_z(tile, 1);
      const int c = (r3e < kPPP ? c18e : 0) * 8;
#pragma unroll
      for (int a = 0; a < kNTab; ++a) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(btab + a * CO + c), hi = *reinterpret_cast<const f32x4*>(btab + a * CO + c + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { bk[a][k] = lo[k]; bk[a][4 + k] = hi[k]; }
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int m0 = wid * 32 + t * 16;
      if constexpr (CO == 64) {
        const E* rt = p.residual ? (const E*)p.residual + pix0 * p.ldy : nullptr;   // (dense frames: row_pitch == W)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          V4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (E)acc[u][t][r];
          *reinterpret_cast<V4*>(stg + li * 128 + (((u * 2 + (g >> 1)) ^ ((li >> 1) & 7)) << 4) + (g & 1) * 8) = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const int r = ps * 8 + (lane >> 3), c = lane & 7;
          V8 v = *reinterpret_cast<const V8*>(stg + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
          if (m0 + r < valid) {
            if (rt) {
              const V8 rv = *reinterpret_cast<const V8*>(rt + (int64_t)(m0 + r) * p.ldy + c * 8);
#pragma unroll
              for (int k = 0; k < 8; ++k) v[k] = (E)((float)v[k] + (float)rv[k]);
            }
            *reinterpret_cast<V8*>(yt + (int64_t)(m0 + r) * p.ldy + c * 8) = v;
          }
        }
      } else {
        constexpr int CH = CO / 8;                 // 16-byte chunks per pixel row (18)
        constexpr int PPP = 64 / CH;               // pixels per pass (3)
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          V4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (E)acc[u][t][r];
          *reinterpret_cast<V4*>(stg + li * kStgPitch + u * 32 + g * 8) = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int r3 = lane / CH, c18 = lane - r3 * CH;
#pragma unroll
        for (int ps = 0; ps < (16 + PPP - 1) / PPP; ++ps) {
          const int px = ps * PPP + r3;
          if (r3 < PPP && px < 16 && m0 + px < valid) {
            const V4 vlo = *reinterpret_cast<const V4*>(stg + px * kStgPitch + c18 * 16);
            const V4 vhi = *reinterpret_cast<const V4*>(stg + px * kStgPitch + c18 * 16 + 8);
            V8 v = V8{vlo[0], vlo[1], vlo[2], vlo[3], vhi[0], vhi[1], vhi[2], vhi[3]};
            int mpx = m0 + px;
            asm volatile("" : "+v"(mpx));          // (the 12 tile-invariant 64-bit store offsets stay out of loop-carried registers)
            int64_t po = mpx;                      // pixel offset from the tile's first: rows of the tile are row_pitch apart
            if (NTAP != 9) {
              const int r = (int)__umulhi((unsigned)mpx, p.magic_w);
              po = (int64_t)r * p.row_pitch + (mpx - r * p.W);
            }
            if constexpr (MODE != 0) {
              // v = the data gradient as it would have been stored (rounded to E); z decides the mask and gives xhat
              const V8 zv = zq[t][ps];
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                const float xf = (float)zv[k];
                const bool on = !p.brelu || fmaf(xf, bk[0][k], bk[1][k]) > 0.f;
                const float dzv = on ? (float)v[k] : 0.f;
                if (MODE == 1) {
                  bs[k] += dzv;
                  bq[k] = fmaf(dzv, fmaf(xf, bk[2][k], bk[3][k]), bq[k]);
                } else {
                  v[k] = (E)fmaf(bk[2][k], dzv, fmaf(bk[MODE == 2 ? 3 : 0][k], xf, bk[MODE == 2 ? 4 : 0][k]));
                }
              }
              if (MODE == 2) *reinterpret_cast<V8*>(yt + po * p.ldy + c18 * 8) = v;
            } else {
              *reinterpret_cast<V8*>(yt + po * p.ldy + c18 * 8) = v;
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
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  }
  if constexpr (CO != 64) {
    if (p.bn_partial) {
      constexpr int CH = CO / 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        bs[k] += __shfl(bs[k], lane + CH, 64) + __shfl(bs[k], lane + 2 * CH, 64);
        bq[k] += __shfl(bq[k], lane + CH, 64) + __shfl(bq[k], lane + 2 * CH, 64);
      }
      if (lane < CH) {
        float* pr = p.bn_partial + ((int64_t)blockIdx.x * kNC + wid) * 2 * p.ldp + lane * 8;
        *reinterpret_cast<f32x4*>(pr) = f32x4{bs[0], bs[1], bs[2], bs[3]};
        *reinterpret_cast<f32x4*>(pr + 4) = f32x4{bs[4], bs[5], bs[6], bs[7]};
        *reinterpret_cast<f32x4*>(pr + p.ldp) = f32x4{bq[0], bq[1], bq[2], bq[3]};
        *reinterpret_cast<f32x4*>(pr + p.ldp + 4) = f32x4{bq[4], bq[5], bq[6], bq[7]};
      }
    }
  }
}

struct Plan { int R, PW, SPR, lds; unsigned magic; };

template <int CI, int CO, int NTAP>
int plan_for(int H, int W, Plan* o) {
  typedef SC<CI, CO, NTAP> C;
  if (W < 2 || H <= 0 || W > 224) return 0;     // (W == 1: the magic reciprocals ceil(2^32 / d) do not exist for d == 1)
  int r = 224 / W;
  if (r > H) r = H;
  if (r < 1) return 0;
  const int PW = W + (NTAP == 9 ? 2 : 0);
  const int spp = C::CK / 8;
  const int spr = spp == 8 ? PW * 8 : ((PW * spp + 31) & ~31);
  const int slots = (r + 2) * spr;
  if ((slots + 63) / 64 > C::NG * C::PPG || r + 2 > 127) return 0;
  const int kPatch = C::NG * C::PPG * 1024;
  const int stg = kNC * 16 * stg_pitch(CO);
  if (!C::STG_DEDICATED && stg > kPatch) return 0;
  o->R = r; o->PW = PW; o->SPR = spr;
  o->lds = 3 * kWStage + 2 * kPatch + (C::STG_DEDICATED ? stg : 0);
  const unsigned d = spp == 8 ? (unsigned)PW : (unsigned)spr;
  o->magic = (unsigned)(((1ull << 32) + d - 1) / d);
  return o->lds <= 160 * 1024;
}

// -> number of output-channel groups (launches) of the pair, 0 = not taken
int plan_any(int Cin, int Cout, int H, int W, Plan* o) {
  if (Cin == 64 && Cout == 144) return plan_for<64, 144, 9>(H, W, o);
  if (Cin == 144 && Cout == 64) return plan_for<144, 64, 9>(H, W, o);
  if (Cin == 128 && Cout == 288) return plan_for<128, 144, 9>(H, W, o) ? 2 : 0;
  if (Cin == 288 && Cout == 128) return plan_for<288, 64, 9>(H, W, o) ? 2 : 0;
  return 0;
}

// (3, 1) form over the [T, L] view of a clip (L = H * W pixels per frame): images = column segments of S pixels, S the divisor of
// L with the fullest 224-pixel tile (ties: the longer segment -- fewer patch rows per output row)
int plan_t(int T, int L, Plan* o, int* S) {
  int best = 0, bs = 0;
  for (int s = 1; s <= 224 && s <= L; ++s) {
    if (L % s) continue;
    Plan pl;
    if (!plan_for<64, 144, 3>(T, s, &pl)) continue;
    const int use = pl.R * s;
    if (use >= best) { best = use; bs = s; }
  }
  if (!bs || best < 112) return 0;
  *S = bs;
  if (!plan_for<64, 144, 3>(T, bs, o)) return 0;
  return (int64_t)(o->R + 2) * L * 64 + (int64_t)bs * 64 < ((int64_t)1 << 24);   // the producer's 24-bit patch offsets
}

template <typename E, int CI, int CO, int NTAP, int MODE = 0>
void launch(const StreamParams& p, int lds, int grid, hipStream_t st) {
  static DvtLdsAttr set;
  dvt_lds_attr(set, (const void*)conv3x3_stream_kernel<E, CI, CO, NTAP, MODE>, 160 * 1024);
  hipLaunchKernelGGL((conv3x3_stream_kernel<E, CI, CO, NTAP, MODE>), dim3(grid), dim3(512), lds, st, p);
}

// geometry of the (3, 1) form over N clips of T frames x L pixels
int params_t(StreamParams* p, Plan* pl, int64_t N, int T, int L) {
  int S = 0;
  if (!plan_t(T, L, pl, &S)) return 0;
  p->hb = L / S;                                    // images: (clip, column segment)
  p->N = (int)(N * p->hb); p->H = T; p->W = S; p->R = pl->R; p->PW = pl->PW; p->SPR = pl->SPR; p->magic = pl->magic;
  p->magic_w = (unsigned)((((uint64_t)1 << 32) + (uint64_t)S - 1) / (uint64_t)S);
  p->pitch_n = (int64_t)T * L; p->pitch_h = S; p->row_pitch = L;
  p->ldy = 144; p->ldp = 144;
  p->tiles_per_img = (int)dvt_cdiv(T, pl->R);
  p->ntiles = (int)(N * p->hb * p->tiles_per_img);
  return 1;
}

}  // namespace

extern "C" {

int dvt_conv3x3_stream_supported(int64_t N, int H, int W, int Cin, int Cout, int dtype) {
  Plan pl;
  return N > 0 && dvt_is_16bit(dtype) && plan_any(Cin, Cout, H, W, &pl) && N * H * W < ((int64_t)1 << 30) ? 1 : 0;
}

int64_t dvt_conv3x3_stream_stats_parts(int64_t N, int H, int W, int Cin, int Cout) {
  Plan pl;
  if (!plan_any(Cin, Cout, H, W, &pl)) return 0;
  const int64_t ntiles = N * dvt_cdiv(H, pl.R);
  return (ntiles < dvt_num_cus() ? ntiles : dvt_num_cus()) * kNC;    // one partial row per compute wave of the persistent grid
}

int dvt_conv3x3_stream(const void* x, const void* w, void* y, float* stats_partial, const void* residual, int64_t N, int H, int W,
                       int Cin, int Cout, int dtype, dvt_stream_t stream) {
  DVT_REQUIRE(x && w && y && N >= 0 && H > 0 && W > 0, "dvt_conv3x3_stream: bad arguments");
  DVT_REQUIRE(dvt_aligned16(x) && dvt_aligned16(w) && dvt_aligned16(y) && dvt_aligned16(stats_partial) && dvt_aligned16(residual),
              "dvt_conv3x3_stream: buffers must be 16-byte aligned");
  if (N == 0) return DVT_OK;
  if (!dvt_conv3x3_stream_supported(N, H, W, Cin, Cout, dtype))
    DVT_UNSUPPORTED("dvt_conv3x3_stream: needs a 16-bit dtype, (Cin, Cout) = (64, 144), (144, 64), (128, 288) or (288, 128) and a patch of 224 / W rows within its LDS buffer");
  DVT_REQUIRE(!(stats_partial && Cout % 144) && !(residual && Cout % 64),
              "dvt_conv3x3_stream: statistics come with the 144- / 288-wide output, the residual with the 64- / 128-wide one");
  Plan pl;
  const int groups = plan_any(Cin, Cout, H, W, &pl);
  const int cog = Cout / groups;                      // output channels per launch: 144 or 64
  StreamParams p{};
  p.x = x; p.w = w; p.y = y; p.bn_partial = stats_partial; p.residual = residual;
  p.ldy = Cout; p.ldp = Cout;
  p.N = (int)N; p.H = H; p.W = W; p.R = pl.R; p.PW = pl.PW; p.SPR = pl.SPR; p.magic = pl.magic;
  p.magic_w = (unsigned)((((uint64_t)1 << 32) + (uint64_t)W - 1) / (uint64_t)W);
  p.hb = 1; p.pitch_n = (int64_t)H * W; p.pitch_h = 0; p.row_pitch = W;
  p.tiles_per_img = (int)dvt_cdiv(H, pl.R);
  p.ntiles = (int)(N * p.tiles_per_img);
  const int grid = p.ntiles < dvt_num_cus() ? p.ntiles : dvt_num_cus();
  hipStream_t st = (hipStream_t)stream;
  const bool h = dtype == DVT_F16;
  for (int gi = 0; gi < groups; ++gi) {               // one launch per group of output channels: its rows of w, its columns of y
    p.w = (const char*)w + (size_t)gi * cog * 9 * Cin * 2;
    p.y = (char*)y + (size_t)gi * cog * 2;
    p.residual = residual ? (const char*)residual + (size_t)gi * cog * 2 : nullptr;
    p.bn_partial = stats_partial ? stats_partial + gi * cog : nullptr;
    if (Cin == 64) { h ? launch<f16, 64, 144, 9>(p, pl.lds, grid, st) : launch<bf16, 64, 144, 9>(p, pl.lds, grid, st); }
    else if (Cin == 144) { h ? launch<f16, 144, 64, 9>(p, pl.lds, grid, st) : launch<bf16, 144, 64, 9>(p, pl.lds, grid, st); }
    else if (Cin == 128) { h ? launch<f16, 128, 144, 9>(p, pl.lds, grid, st) : launch<bf16, 128, 144, 9>(p, pl.lds, grid, st); }
    else { h ? launch<f16, 288, 64, 9>(p, pl.lds, grid, st) : launch<bf16, 288, 64, 9>(p, pl.lds, grid, st); }
  }
  DVT_LAUNCH_CHECK("dvt_conv3x3_stream");
  return DVT_OK;
}

int dvt_conv3x1_stream_supported(int64_t N, int T, int L, int Cin, int Cout, int dtype) {
  Plan pl;
  int S;
  return N > 0 && T > 0 && L > 0 && Cin == 64 && Cout == 144 && dvt_is_16bit(dtype) && plan_t(T, L, &pl, &S) &&
                 N * T * L < ((int64_t)1 << 30) ? 1 : 0;
}

int dvt_conv3x1_stream(const void* x, const void* w, void* y, int64_t N, int T, int L, int Cin, int Cout, int dtype,
                       dvt_stream_t stream) {
  DVT_REQUIRE(x && w && y && N >= 0 && T > 0 && L > 0, "dvt_conv3x1_stream: bad arguments");
  DVT_REQUIRE(dvt_aligned16(x) && dvt_aligned16(w) && dvt_aligned16(y), "dvt_conv3x1_stream: buffers must be 16-byte aligned");
  if (N == 0) return DVT_OK;
  if (!dvt_conv3x1_stream_supported(N, T, L, Cin, Cout, dtype))
    DVT_UNSUPPORTED("dvt_conv3x1_stream: needs a 16-bit dtype, (Cin, Cout) = (64, 144) and a divisor of L that fills half a 224-pixel tile");
  Plan pl;
  StreamParams p{};
  params_t(&p, &pl, N, T, L);
  p.x = x; p.w = w; p.y = y; p.bn_partial = nullptr; p.residual = nullptr;
  const int grid = p.ntiles < dvt_num_cus() ? p.ntiles : dvt_num_cus();
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DVT_BF16) launch<bf16, 64, 144, 3>(p, pl.lds, grid, st);
  else launch<f16, 64, 144, 3>(p, pl.lds, grid, st);
  DVT_LAUNCH_CHECK("dvt_conv3x1_stream");
  return DVT_OK;
}

size_t dvt_conv3x1_stream_bn_bwd_workspace_bytes(int64_t N, int T, int L) {
  (void)N; (void)T; (void)L;
  return ((size_t)dvt_num_cus() * kNC * 2 + 2) * 144 * sizeof(float) + 64;     // partial rows of the persistent grid + [2][144]
}

int dvt_conv3x1_stream_bn_bwd(const void* dy, const void* w, const void* z, const dvt_bn_affine* bn, void* dz, float* dgamma,
                              float* dbeta, void* workspace, int64_t N, int T, int L, int training, int accumulate, int dtype,
                              dvt_stream_t stream) {
  DVT_REQUIRE(dy && w && z && bn && dz && dgamma && dbeta && workspace && N >= 0 && T > 0 && L > 0,
              "dvt_conv3x1_stream_bn_bwd: bad arguments");
  DVT_REQUIRE(bn->mean && bn->invstd && bn->gamma && bn->beta && (bn->c_valid == 0 || bn->c_valid == 144),
              "dvt_conv3x1_stream_bn_bwd: the BatchNorm needs mean, invstd, gamma, beta over all 144 channels");
  DVT_REQUIRE(dvt_aligned16(dy) && dvt_aligned16(w) && dvt_aligned16(z) && dvt_aligned16(dz) && dvt_aligned16(workspace),
              "dvt_conv3x1_stream_bn_bwd: buffers must be 16-byte aligned");
  if (N == 0) return DVT_OK;
  if (!dvt_conv3x1_stream_supported(N, T, L, 64, 144, dtype))
    DVT_UNSUPPORTED("dvt_conv3x1_stream_bn_bwd: needs a 16-bit dtype and a divisor of L that fills half a 224-pixel tile");
  hipStream_t st0 = (hipStream_t)stream;
  if (dvt_internal::conv3x1_dbn_supported(N, T, L, dtype)) {
    // the window kernel with helper waves (conv3x1_dbn.hip): one partial row per workgroup
    float* part0 = (float*)workspace;
    float* loc0 = part0 + (size_t)dvt_num_cus() * kNC * 2 * 144;
    const int parts = dvt_internal::conv3x1_dbn_parts(N, T, L);
    int rc = dvt_internal::conv3x1_dbn_pass(1, dy, w, 3 * 64, z, bn->mean, bn->invstd, bn->gamma, bn->beta, bn->relu, training, part0,
                                            nullptr, nullptr, N, T, L, dtype, st0);
    if (rc != DVT_OK) return rc;
    DVT_LAUNCH_CHECK("dvt_conv3x1_stream_bn_bwd(window sums)");
    dvt_internal::bn_bwd_finalize(st0, part0, parts, 144, loc0, accumulate, dgamma, dbeta, 144);
    DVT_LAUNCH_CHECK("dvt_conv3x1_stream_bn_bwd(finalize)");
    rc = dvt_internal::conv3x1_dbn_pass(2, dy, w, 3 * 64, z, bn->mean, bn->invstd, bn->gamma, bn->beta, bn->relu, training, nullptr,
                                        loc0, dz, N, T, L, dtype, st0);
    if (rc != DVT_OK) return rc;
    DVT_LAUNCH_CHECK("dvt_conv3x1_stream_bn_bwd(window apply)");
    return DVT_OK;
  }
  Plan pl;
  StreamParams p{};
  params_t(&p, &pl, N, T, L);
  const int grid = p.ntiles < dvt_num_cus() ? p.ntiles : dvt_num_cus();
  float* part = (float*)workspace;
  float* loc = part + (size_t)dvt_num_cus() * kNC * 2 * 144;
  p.x = dy; p.w = w; p.y = dz; p.residual = nullptr; p.bn_partial = part;
  p.bz = z; p.bmean = bn->mean; p.binvstd = bn->invstd; p.bgamma = bn->gamma; p.bbeta = bn->beta; p.bloc = loc;
  p.brelu = bn->relu; p.btraining = training;
  p.inv_rows = 1.0f / (float)(N * T * L);
  hipStream_t st = (hipStream_t)stream;
  // pass 1: the sums (nothing stored); their fixed-order reduction; pass 2: the same data gradient again, corrected and stored
  const int lds = pl.lds + 5 * 144 * (int)sizeof(float);            // + the per-channel constants table
  DVT_REQUIRE(lds <= 160 * 1024, "dvt_conv3x1_stream_bn_bwd: no LDS left for the BatchNorm constants");
  if (dtype == DVT_BF16) launch<bf16, 64, 144, 3, 1>(p, lds, grid, st);
  else launch<f16, 64, 144, 3, 1>(p, lds, grid, st);
  DVT_LAUNCH_CHECK("dvt_conv3x1_stream_bn_bwd(sums)");
  dvt_internal::bn_bwd_finalize(st, part, grid * kNC, 144, loc, accumulate, dgamma, dbeta, 144);
  DVT_LAUNCH_CHECK("dvt_conv3x1_stream_bn_bwd(finalize)");
  p.bn_partial = nullptr;
  if (dtype == DVT_BF16) launch<bf16, 64, 144, 3, 2>(p, lds, grid, st);
  else launch<f16, 64, 144, 3, 2>(p, lds, grid, st);
  DVT_LAUNCH_CHECK("dvt_conv3x1_stream_bn_bwd(apply)");
  return DVT_OK;
}

}  // extern "C"
