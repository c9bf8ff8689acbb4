// conv3x3_wgrad.hip -- weight gradient of the 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels
// (layer 1 of ResNet-18, custom_resnet.py:19-22,109: four such layers per frame) from LDS-resident halo patches.
//
//   dW[co][ci][ki][kj] = sum over frames and pixels (h, w) of dz[h, w, co] * x[h + ki - 1, w + kj - 1, ci]
//
// The implicit form (gemm256.hip, mn-major operands) gathers the input map once per filter tap -- nine shifted copies of
// every pixel through the CU's vector-memory path per k-tile (150 us per layer at 256 frames of 56^2, 0.17 of the MFMA
// peak).  Here a workgroup owns R whole output rows of one frame at a time, stages the (R + 2) x (W + 2) x 64 input patch
// and the R x W x 64 gradient tile ONCE each, and all nine taps read their shifted windows of the patch from LDS.
//
//   * Both images are [position][64 channels] with 128-byte rows (the mn-major layout of the GEMM family: operands are
//     read with ds_read_b64_tr_b16, 32-byte units XOR-swizzled by f(position)).  The gradient tile is laid out in the
//     PATCH's coordinate system -- PW = W + 2 positions per row, the two halo columns zero -- so that tap (ki, kj) of
//     output position p is patch position p + ki * PW + kj for every p: the reduction runs over positions, and a tap is a
//     constant address offset.  KP = R * PW rounded up to 32 positions per tile (zero gradient rows behind the last).
//   * Output [co 64][tap * 64 + ci 576] = 4 x 36 MFMA blocks; 8 waves: waves 0..3 own five (tap, ci-block) column blocks
//     each, waves 4..7 four (wave w and w + 4 share a SIMD: nine blocks, 36 MFMAs per 32 positions, on every SIMD), for
//     all four co blocks; the next step's fragments (4 dz + 5 x transposing reads) are requested under the MFMAs of the
//     current one (a 12-wave form without that prefetch -- 168 VGPRs -- ran at 13 k cycles per tile, 78-88 us per layer).
//     Measured at 256 frames of 56^2 (tools/dev/cw_probe.py, kernel with parts switched off): 96 us in all; the DMA stream
//     and barriers alone 46 us; the LDS reads + MFMAs alone 79 us, i.e. ~13.5 k cycles per tile for 288 MFMAs per SIMD
//     (4.6 k): 136 transposing reads per 32 positions and CU cost ~12.5 cycles each.  ds_read_b64_tr_b16 moves 8 bytes
//     per lane; one such read per MFMA is the floor of any kernel whose two operands are both channel-contiguous.
//   * Persistent grid (one workgroup per CU); the accumulators live in registers over the workgroup's whole tile sequence;
//     the next tile's two images stream into second buffers under the MFMAs of the current one.  At the end every
//     workgroup leaves ONE fp32 partial [576][64] in a slab, and the family's split-K reduce (dvt_splitk_pending with
//     conv_taps: scatter into the parameter's own [co][ci][3][3] layout) sums the slabs -- stand-alone or carried.
#include "common.h"

namespace {

constexpr int kC = 64;
constexpr int kNW = 8;                       // waves per workgroup
constexpr int kM = 9 * kC;                   // 576 slab rows: tap * 64 + ci
constexpr int kMaxXP = 8;                    // patch pieces (1 KiB) per wave: patch <= 64 KiB
constexpr int kMaxZP = 6;                    // gradient-tile pieces per wave: tile <= 48 KiB

struct CwParams {
  const void* x;        // [N, H, W, 64]
  const void* dz;       // [N, H, W, 64]
  float* slab;          // [grid][576][64]
  int N, H, W, R, PW, KP, tiles_per_img, ntiles;
  int patch_pos;        // (R + 2) * PW: positions the patch DMA covers
  int patch_bytes, dz_bytes;
  int ldz, zc0, zcv;    // dz rows are ldz channels long; this launch takes channels [zc0, zc0 + zcv) of them (zcv <= 64; MB = 5: <= 80)
  int zt_off;           // MB = 5: byte offset of the fifth channel block's image inside a gradient-tile buffer (KP * 128)
};

__device__ __attribute__((aligned(16))) unsigned int cw_zero16[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ int cw_swz(int k) { return ((k >> 1) & 1) | (((k >> 3) & 1) << 1); }
// The fifth channel block (MB = 5) has an image of its own behind the 64-channel one: 32 bytes per position, positions in
// groups of 16 with bits 2 and 3 of the position swapped, so that the positions {q, 8 + q} + 4 hf of a 32-lane half of
// ds_read_b64_tr_b16 are eight consecutive 32-byte slots (256 bytes = every bank once).  Self-inverse.
__device__ __forceinline__ int cw_tail_slot(int k) { return (k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1); }

// MB: output-channel blocks of 16 this launch computes (4: a whole 64-channel group; 5: the last 80 of a 144-wide dz -- a
// 16-channel launch of its own re-staged every input patch for a ninth of the work, 70 us against the group's 115; 1: a
// 16-channel tail).  The slab rows are SN = 80 floats for MB = 5, else 64.
template <int N> struct IntC { static constexpr int value = N; };

template <typename E, int MB>
__global__ __launch_bounds__(kNW * 64) void conv3x3_c64_wgrad_kernel(const CwParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using V8 = typename Elem16<E>::v8;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int PW = p.PW;
  // Two patch buffers, then two gradient-tile buffers, addressed as smem + byte offset -- never through a pointer picked at
  // run time: `char* xb[2]` indexed by it & 1 made every fragment address a 64-bit generic pointer (add, null check, select
  // and cast per transposing read; 100 such sequences in the listing and 256 registers)
  const int zbase = 2 * p.patch_bytes;
  const E* xg = (const E*)p.x;
  const E* zg = (const E*)p.dz;
  const int xp = (p.patch_pos * 128 + 1023) >> 10;       // pieces the patch DMA writes
  const int zp = p.dz_bytes >> 10;
  constexpr bool kTail = MB == 5;
  constexpr int MBm = kTail ? 4 : MB;                    // blocks read from the 64-channel image
  constexpr int SN = kTail ? 80 : kC;
  const int zpm = kTail ? p.zt_off >> 10 : zp;           // pieces of the 64-channel image

  // positions of a patch buffer behind the DMA's range are read (times zero gradient rows) but never written: zero once
  for (int b = 0; b < 2; ++b)
    for (int i = xp * 1024 + threadIdx.x * 16; i < p.patch_bytes; i += kNW * 64 * 16)
      *reinterpret_cast<f32x4*>(smem + b * p.patch_bytes + i) = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- per-lane coordinates of this wave's DMA pieces (fixed for the whole launch): slot = 16-byte chunk of a position
  // one register per piece: row << 17 | column << 5 | source chunk (16 bytes = 8 channels; 8, 9: the fifth block) << 1 | valid
  int xq[kMaxXP], zq[kMaxZP];
  const int c16 = lane & 7;
#pragma unroll
  for (int i = 0; i < kMaxXP; ++i) {
    const int piece = wid + kNW * i;
    const int pos = (piece * 64 + lane) >> 3;
    const int r = pos / PW, c = pos - r * PW;
    const int ok = (piece < xp && pos < p.patch_pos) ? 1 : 0;
    xq[i] = (r << 17) | (c << 5) | (((((c16 >> 1) ^ cw_swz(pos)) << 1) | (c16 & 1)) << 1) | ok;
  }
#pragma unroll
  for (int i = 0; i < kMaxZP; ++i) {
    const int piece = wid + kNW * i;
    int pos = (piece * 64 + lane) >> 3;
    int chunk = (((c16 >> 1) ^ cw_swz(pos)) << 1) | (c16 & 1);
    if (kTail && piece >= zpm) {                           // the fifth block's image: two 16-byte chunks per position
      const int sl = (piece - zpm) * 64 + lane;
      pos = cw_tail_slot(sl >> 1);
      chunk = 8 + (sl & 1);
    }
    const int r = pos / PW, c = pos - r * PW;
    const int ok = (piece < zp && r < p.R && c < p.W) ? 1 : 0;
    zq[i] = (r << 17) | (c << 5) | (chunk << 1) | ok;
  }
  auto load_tile = [&](int tile, int b) {
    const int n = tile / p.tiles_per_img, h0 = (tile - n * p.tiles_per_img) * p.R;
#pragma unroll
    for (int i = 0; i < kMaxXP; ++i) {
      const int piece = wid + kNW * i;
      if (piece < xp) {                          // wave-uniform
        const int h = h0 - 1 + (xq[i] >> 17), w = ((xq[i] >> 5) & 0xFFF) - 1;
        const bool ok = (xq[i] & 1) && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
        const E* src = ok ? xg + ((int64_t)(n * p.H + h) * p.W + w) * kC + ((xq[i] >> 1) & 15) * 8
                          : reinterpret_cast<const E*>(cw_zero16);
        dvt_dma16(src, smem + b * p.patch_bytes + piece * 1024);
      }
    }
#pragma unroll
    for (int i = 0; i < kMaxZP; ++i) {
      const int piece = wid + kNW * i;
      if (piece < zp) {
        const int h = h0 + (zq[i] >> 17);
        const int sc = ((zq[i] >> 1) & 15) * 8;                  // channel of this chunk inside the group
        const bool ok = (zq[i] & 1) && h < p.H && sc < p.zcv;
        const E* src = ok ? zg + ((int64_t)(n * p.H + h) * p.W + ((zq[i] >> 5) & 0xFFF)) * p.ldz + p.zc0 + sc
                          : reinterpret_cast<const E*>(cw_zero16);
        dvt_dma16(src, smem + zbase + b * p.dz_bytes + piece * 1024);
      }
    }
  };

  // this wave's column blocks nb0 .. nb0 + cnt - 1: nb -> tap = nb / 4 (offset ki * PW + kj), ci block = nb % 4.
  // Byte offsets of this lane's two transposing reads per fragment inside an image, for k-step 0; a k-step adds 32 positions
  // = 4 KiB, which changes neither bit 1 nor bit 3 of the position: the swizzle term is fixed for the whole launch.
  const int cnt = wid < 4 ? 5 : 4;                        // wave-uniform
  const int nb0 = wid < 4 ? 5 * wid : 20 + 4 * (wid - 4);
  int zo[2], xo[5][2], zt[2];                             // (dz block m: zo ^ (m << 5) -- the unit field of the offset)
  {
    const int q = li >> 2, pp = li & 3;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int k = 8 * g + 4 * hf + q;
      zo[hf] = k * 128 + (cw_swz(k) << 5) + 8 * pp;
      zt[hf] = p.zt_off + cw_tail_slot(k) * 32 + 8 * pp;       // (a k-step adds 32 positions = 1 KiB of this image)
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int nb = min(nb0 + j, 35), tap = nb >> 2, ki = tap / 3, kj = tap - 3 * ki;
        const int kx = k + ki * PW + kj;
        xo[j][hf] = kx * 128 + (((nb & 3) ^ cw_swz(kx)) << 5) + 8 * pp;
      }
    }
  }
  f32x4 acc[MB][5];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int tile = blockIdx.x;
  if (tile < p.ntiles) load_tile(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int nks = p.KP >> 5;
  // The tile sequence with the wave's count of column blocks as a compile-time constant (instantiated for 5 and 4, picked per
  // wave; both forms pass the same barriers): as a run-time test it put a branch around every row's fifth MFMA.
  auto run = [&](auto CNT) {
  constexpr int NJ = decltype(CNT)::value;
  for (int it = 0; tile < p.ntiles; ++it, tile += gridDim.x) {
    const int cx = (it & 1) * p.patch_bytes;
    const int cz = zbase + (it & 1) * p.dz_bytes;
    if (tile + (int)gridDim.x < p.ntiles) load_tile(tile + gridDim.x, (it + 1) & 1);
    // x fragments in two alternating sets (every row of blocks needs all of them); the dz fragment of row m is refilled IN
    // PLACE for the next step as soon as row m's MFMAs are issued (one set: MB = 5 with two sets of both spilled 38 registers)
    V8 zf[MB], xf[2][5];
    auto rdz = [&](int ks, int m) {
      if (kTail && m == MB - 1)
        zf[m] = __builtin_shufflevector(Elem16<E>::tr_read(smem + cz + ks * 1024 + zt[0]), Elem16<E>::tr_read(smem + cz + ks * 1024 + zt[1]),
                                        0, 1, 2, 3, 4, 5, 6, 7);
      else
        zf[m] = __builtin_shufflevector(Elem16<E>::tr_read(smem + cz + ks * 4096 + (zo[0] ^ (m << 5))),
                                        Elem16<E>::tr_read(smem + cz + ks * 4096 + (zo[1] ^ (m << 5))), 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto rdx = [&](int ks, V8* xv) {
      const char* bx = smem + cx + ks * 4096;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
          xv[j] = __builtin_shufflevector(Elem16<E>::tr_read(bx + xo[j][0]), Elem16<E>::tr_read(bx + xo[j][1]), 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // MFMAs of step ks; row m's dz fragment of the next step is requested behind row m -- unconditionally (the last step
    // re-reads its own: a branch per row cut the MFMA sequence into blocks that each waited for every LDS read in flight)
    auto step = [&](int ks, const V8* xv) {
      const int kn = ks + 1 < nks ? ks + 1 : ks;
#pragma unroll
      for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[m][j] = Elem16<E>::mma(zf[m], xv[j], acc[m][j]);
        rdz(kn, m);
      }
    };
#pragma unroll
    for (int m = 0; m < MB; ++m) rdz(0, m);
    rdx(0, xf[0]);
    for (int ks = 0; ks < nks; ks += 2) {        // two steps per trip: the x sets alternate without indexing
      if (ks + 1 < nks) rdx(ks + 1, xf[1]);
      step(ks, xf[0]);
      if (ks + 1 < nks) {
        if (ks + 2 < nks) rdx(ks + 2, xf[0]);
        step(ks + 1, xf[1]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the next tile's images have landed
    __builtin_amdgcn_s_barrier();                                  // and everybody is done with this tile's
  }
  };
  if (cnt == 5) run(IntC<5>{});
  else run(IntC<4>{});

  // ---- this workgroup's partial: slab[blockIdx][m = tap * 64 + 16 cb + li][n = 16 mb + 4 g .. + 3] (rows of SN floats)
  // (mma(dz fragment, x fragment): lane (g, li) holds C[co = 16 mb + 4 g + r][ci = 16 cb + li])
  float* out = p.slab + (int64_t)blockIdx.x * kM * SN;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    if (j >= cnt) break;
    const int nb = nb0 + j, tap = nb >> 2, cb = nb & 3;
    float* row = out + (int64_t)(tap * kC + cb * 16 + li) * SN + 4 * g;
#pragma unroll
    for (int m = 0; m < MB; ++m) *reinterpret_cast<f32x4*>(row + 16 * m) = acc[m][j];   // (MB == 1: columns 16.. stay unwritten, unread)
  }
}

// rows per tile and the two image sizes for a W-wide map; zrow = bytes of a gradient-tile position (128; MB = 5: 160)
int plan(int H, int W, CwParams* q, int zrow = 128) {
  if (H < 1 || W < 1) return 0;
  const int PW = W + 2;
  for (int R = H < 16 ? H : 16; R >= 1; --R) {
    const int KP = (R * PW + 31) & ~31;
    const int patch_pos = (R + 2) * PW;
    const int need = KP + 2 * PW + 2 > patch_pos ? KP + 2 * PW + 2 : patch_pos;
    const int pbytes = ((need * 128) + 1023) & ~1023, zbytes = KP * zrow;
    const int xp = (patch_pos * 128 + 1023) >> 10, zp = zbytes >> 10;
    if (2 * (pbytes + zbytes) > 160 * 1024) continue;
    if (xp > kNW * kMaxXP || zp > kNW * kMaxZP) continue;
    if (KP > 512) continue;                               // (long reductions per tile gain nothing: keep tiles plentiful)
    q->R = R; q->PW = PW; q->KP = KP; q->patch_pos = patch_pos; q->patch_bytes = pbytes; q->dz_bytes = zbytes;
    q->zt_off = KP * 128;
    return 1;
  }
  return 0;
}

}  // namespace

extern "C" int dvt_conv3x3_c64_wgrad_supported(int64_t N, int H, int W, int dtype);

namespace {

int cw_grid(int64_t N, int H, const CwParams& q) {
  const int64_t ntiles = N * dvt_cdiv(H, q.R);
  return (int)(ntiles < dvt_num_cus() ? ntiles : dvt_num_cus());
}

// one launch: channels [c0, c0 + cv) of a dz whose rows are ldz channels long, partials into p.slab
template <typename E>
void cw_launch(const CwParams& p, int grid, int lds, hipStream_t st) {
  if (p.zcv > kC) {
    static DvtLdsAttr set;
    dvt_lds_attr(set, (const void*)conv3x3_c64_wgrad_kernel<E, 5>, 160 * 1024);
    hipLaunchKernelGGL((conv3x3_c64_wgrad_kernel<E, 5>), dim3(grid), dim3(kNW * 64), lds, st, p);
  } else if (p.zcv > 16) {
    static DvtLdsAttr set;
    dvt_lds_attr(set, (const void*)conv3x3_c64_wgrad_kernel<E, 4>, 160 * 1024);
    hipLaunchKernelGGL((conv3x3_c64_wgrad_kernel<E, 4>), dim3(grid), dim3(kNW * 64), lds, st, p);
  } else {
    static DvtLdsAttr set;
    dvt_lds_attr(set, (const void*)conv3x3_c64_wgrad_kernel<E, 1>, 160 * 1024);
    hipLaunchKernelGGL((conv3x3_c64_wgrad_kernel<E, 1>), dim3(grid), dim3(kNW * 64), lds, st, p);
  }
}

// Cz output channels (64, or 80..  in steps of 16: the 144 mid planes of R(2+1)D-18's layer 1) as groups of 64 with a last
// group of up to 80 (144 = 64 + 80: two launches; an 80-wide group has its own plan -- fewer rows per tile, 160-byte
// gradient positions): one launch and one reduce per group over the SAME slab region (stream order), the last group's
// reduce deferred on request.
int cw_run(const void* x, const void* dz, float* dw, void* workspace, int64_t N, int H, int W, int Cz, int accumulate,
                  int defer_reduce, dvt_splitk_pending* pending, int dtype, dvt_stream_t stream, const char* who) {
  DVT_REQUIRE(x && dz && dw && workspace && N > 0 && H > 0 && W > 0, "%s: bad arguments", who);
  DVT_REQUIRE(dvt_aligned16(x) && dvt_aligned16(dz) && dvt_aligned16(dw) && dvt_aligned16(workspace),
              "%s: buffers must be 16-byte aligned", who);
  DVT_REQUIRE(!defer_reduce || pending, "%s: defer_reduce needs a pending descriptor to fill", who);
  CwParams p64, p80;
  if (!dvt_conv3x3_c64_wgrad_supported(N, H, W, dtype) || Cz < 64 || Cz % 16)
    DVT_UNSUPPORTED("%s: needs a 16-bit dtype, Cout >= 64 in steps of 16 and two (patch + gradient tile) pairs in 160 KiB of LDS", who);
  plan(H, W, &p64);
  const bool wide_ok = plan(H, W, &p80, 160) != 0;        // (else the last 80 go as 64 + 16)
  hipStream_t st = (hipStream_t)stream;
  for (int c0 = 0; c0 < Cz;) {
    const int rem = Cz - c0;
    const int cv = rem > kC && rem <= 80 && wide_ok ? rem : (rem < kC ? rem : kC);
    CwParams p = cv > kC ? p80 : p64;
    p.x = x; p.dz = dz; p.slab = (float*)workspace;
    p.N = (int)N; p.H = H; p.W = W; p.ldz = Cz;
    p.tiles_per_img = (int)dvt_cdiv(H, p.R);
    p.ntiles = (int)(N * p.tiles_per_img);
    p.zc0 = c0; p.zcv = cv;
    const int grid = cw_grid(N, H, p);
    const int lds = 2 * (p.patch_bytes + p.dz_bytes);
    const int SN = cv > kC ? 80 : kC;                      // slab row length of the launch's kernel
    if (dtype == DVT_BF16) cw_launch<bf16>(p, grid, lds, st);
    else cw_launch<f16>(p, grid, lds, st);
    DVT_LAUNCH_CHECK(who);
    // the slabs are summed by the family's split-K reduce, which scatters [tap * 64 + ci][co] into the parameter's [co][ci][3][3]
    dvt_splitk_pending q{};
    q.slab = p.slab; q.splits = grid; q.valid = 1; q.M = kM; q.N = SN; q.C = dw + (int64_t)c0 * kM; q.ldc = SN;
    q.accumulate = accumulate; q.cs_accumulate = 0; q.cs_slab = nullptr; q.cs_out = nullptr;
    q.conv_cin = kC; q.conv_taps = 9; q.conv_cin_l = 0; q.conv_cout_l = cv < SN ? cv : 0;
    c0 += cv;
    if (defer_reduce && c0 >= Cz) {
      *pending = q;
      return DVT_OK;
    }
    const int rc = dvt_splitk_reduce_pending(&q, stream);
    if (rc != DVT_OK) return rc;
  }
  return DVT_OK;
}

}  // namespace

extern "C" {

int dvt_conv3x3_c64_wgrad_supported(int64_t N, int H, int W, int dtype) {
  CwParams q;
  return N > 0 && dvt_is_16bit(dtype) && plan(H, W, &q) && N * H * W < ((int64_t)1 << 31) ? 1 : 0;
}

size_t dvt_conv3x3_c64_wgrad_workspace_bytes(int64_t N, int H, int W) {
  CwParams q, q80;
  if (N <= 0 || !plan(H, W, &q)) return 0;
  int grid = cw_grid(N, H, q);
  if (plan(H, W, &q80, 160)) {
    const int g80 = cw_grid(N, H, q80);
    grid = g80 > grid ? g80 : grid;
  }
  return (size_t)grid * kM * 80 * sizeof(float);         // (rows of 80 floats: the widest group of dvt_conv3x3_c64_wgrad_wide)
}

int dvt_conv3x3_c64_wgrad(const void* x, const void* dz, float* dw, void* workspace, int64_t N, int H, int W, int accumulate,
                          int defer_reduce, dvt_splitk_pending* pending, int dtype, dvt_stream_t stream) {
  return cw_run(x, dz, dw, workspace, N, H, W, kC, accumulate, defer_reduce, pending, dtype, stream, "dvt_conv3x3_c64_wgrad");
}

int dvt_conv3x3_c64_wgrad_wide(const void* x, const void* dz, float* dw, void* workspace, int64_t N, int H, int W, int Cout,
                               int accumulate, int defer_reduce, dvt_splitk_pending* pending, int dtype, dvt_stream_t stream) {
  return cw_run(x, dz, dw, workspace, N, H, W, Cout, accumulate, defer_reduce, pending, dtype, stream, "dvt_conv3x3_c64_wgrad_wide");
}

}  // extern "C"
