// conv3x1_dbn.hip -- data gradient of the temporal half of R(2+1)D-18's layer-1 Conv2Plus1D (torchvision's layout behind
// frame_transformer.py:64-74: a (3, 1, 1) convolution 144 mid planes -> 64 planes; its data gradient is the (3, 1) convolution
// 64 -> 144 of dy with the rotated weights) WITH the backward of the mid-plane BatchNorm (+ ReLU) in front of that layer in its
// epilogue -- the operator of dvt_conv3x1_stream_bn_bwd, as a window kernel with helper waves (round 6).
//
//   d[n, t, p, c]  = sum over kt, j of dy[n, t + kt - 1, p, j] * Wd[c][kt * 64 + j]            (never stored)
//   dzm            = d (rounded to the map's type) under the ReLU mask recomputed from z
//   pass 1 (MODE 1): per channel  sum dzm,  sum dzm * xhat          -> one partial row per workgroup
//   pass 2 (MODE 2): dz = gamma * invstd * (dzm - sum dzm / rows - xhat * sum dzm xhat / rows)   (the sums: eval mode 0)
//
// The first form (conv3x3_stream.hip, MODE 1 / 2 of the streamed-weight kernel: 7 compute waves x 32 pixels x all 144
// channels, weights streamed through LDS, the z rows requested at the start of each tile's epilogue because 246 - 256
// registers leave no room to hold them) ran 155 - 167 + 175 - 186 us per layer: MFMA pipe 14 - 16 % busy, the waves parked at
// s_waitcnt / s_barrier for 47 % of their cycles (profiles/r06_conv3x1_bound.md).  Here, per tile = a segment of S pixels of one
// clip over all T frames (conv3x1_c64.hip's window: [position][64] in 144-byte rows, three taps = three position offsets):
//   * waves 0 - 8 (compute): wave u owns output channels [16 u, 16 u + 16) -- its 16 x 192 weights are six fragments in
//     registers for the whole launch -- and runs all of the tile's 16-position blocks; the results go, rounded, into one of two
//     staging images [position][144] (296-byte rows: the 16 lanes of a ds_write_b64 group on 16 different bank slots);
//   * waves 9 - 15 (helpers), in the same interval and behind the same ONE barrier per tile: request the window two tiles
//     ahead (LDS-DMA; no transform here, so a window has two intervals to land), request the z rows of the tile being
//     computed into registers, and run the BatchNorm epilogue of the PREVIOUS tile from its staging image and the z rows
//     requested an interval ago: a thread owns one 16-byte channel group for the whole launch (its 32 - 40 per-channel
//     constants in registers) and the rows r, r + 24, r + 48, ... of a tile.
// Same arithmetic, same rounding points and the same formulas as the first form (the folded affine of BnAffine::init decides
// every mask); the partial sums are added in another order.
#include "common.h"

namespace {

constexpr int kCI = 64, kCO = 144, kNC = 9, kNH = 7, kNWv = kNC + kNH;        // compute / helper waves
constexpr int kXRow = 144;                   // window bytes per position: 8 data slots + 1 padding slot (conv3x1_c64.hip)
constexpr int kSlots = 9;
constexpr int kMaxXP = 5;                    // window DMA pieces (1 KiB) per helper wave
constexpr int kSPitch = 296;                 // staging bytes per position
constexpr int kCH = kCO / 8;                 // 18 chunks of 16 bytes per position
constexpr int kHRows = (kNH * 64) / kCH;     // 24 row lanes among the helpers (432 of their 448 threads)
constexpr int kMaxRows = 4;                  // rows per helper thread and tile: tile <= 96 positions (registers: 16 waves, 128 each)

struct Win {
  int T, L, S, segs, KP, xpos, x_bytes;
};

struct DbParams {
  const void* dy;       // [N, T, L, 64]
  const void* w;        // [144][ldw] k-major, k = tap * 64 + j (data-gradient pack)
  const void* z;        // [N, T, L, 144]: the convolution output the BatchNorm normalised
  void* dz;             // [N, T, L, 144] (MODE 2)
  float* partial;       // [grid][2][144] (MODE 1)
  const float *mean, *invstd, *gamma, *beta;
  const float* loc;     // MODE 2: [2][144] sum dzm * xhat, sum dzm of the whole launch (bn_bwd_finalize)
  float inv_rows;
  int relu, training;
  Win w_;
  int ntiles, ldw;
};

__device__ __attribute__((aligned(16))) unsigned int db_zero16[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void db_wait_vm(int n) {      // n is wave-uniform
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
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
  }
}

// NB = 16-position blocks of a tile (KP / 16), MODE 1 = sums, 2 = corrected gradient
template <typename E, int NB, int MODE>
__global__ __launch_bounds__(kNWv * 64) void conv3x1_dbn_kernel(const DbParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using V8 = typename Elem16<E>::v8;
  using V4 = typename Elem16<E>::v4;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Win& w = p.w_;
  const int S = w.S;
  const int xb = w.x_bytes, gb = w.KP * kSPitch, goff = 3 * w.x_bytes;
  const int n_my = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // tiles of this workgroup (>= 1)
  constexpr int NR = (NB * 16 + kHRows - 1) / kHRows;        // rows per helper thread and tile

  if (wid >= kNC) {
    // ================================================================ helper waves
    const int hw = wid - kNC, htid = threadIdx.x - kNC * 64;
    const E* xg = (const E*)p.dy;
    const E* zg = (const E*)p.z;
    E* og = (E*)p.dz;
    const int xp = w.x_bytes >> 10;
    // window DMA pieces of this wave: frame row << 20 | pixel of the segment << 8 | channel of the chunk, bit 31 = never loaded
    unsigned xq[kMaxXP];
    int np = 0;
#pragma unroll
    for (int i = 0; i < kMaxXP; ++i) {
      const int piece = hw + kNH * i;
      const int sl = piece * 64 + lane;
      const int pos = sl / kSlots, c = sl - pos * kSlots;
      const int tt = pos / S, sx = pos - tt * S;
      const bool ok = piece < xp && pos < w.xpos && c < 8;
      xq[i] = ok ? ((unsigned)tt << 20) | ((unsigned)sx << 8) | (unsigned)(c * 8) : 0x80000000u;
      np += piece < xp ? 1 : 0;
    }
    auto pix_of = [&](int j) -> int64_t {
      const int tile = blockIdx.x + j * gridDim.x;
      const int n = tile / w.segs, sg = tile - n * w.segs;
      return (int64_t)n * w.T * w.L + (int64_t)sg * S;
    };
    auto load_window = [&](int j) {
      const int64_t pix0 = pix_of(j);
      char* dst = smem + (j % 3) * xb;
#pragma unroll
      for (int i = 0; i < kMaxXP; ++i) {
        const int piece = hw + kNH * i;
        if (piece < xp) {                          // wave-uniform
          const int frame = (int)((xq[i] >> 20) & 0x7FF) - 1;
          const bool ok = (int)xq[i] >= 0 && (unsigned)frame < (unsigned)w.T;
          const E* src = ok ? xg + (pix0 + (int64_t)frame * w.L + ((xq[i] >> 8) & 0xFFF)) * kCI + (xq[i] & 0xFF)
                            : reinterpret_cast<const E*>(db_zero16);
          dvt_dma16(src, dst + piece * 1024);
        }
      }
    };
    // this thread: channel group c18 (8 channels) of the rows rr, rr + 24, ... of every tile
    const int rr = htid / kCH, c18 = htid - rr * kCH;
    const bool live = rr < kHRows;
    const int ch0 = (live ? c18 : 0) * 8;
    int roff[NR];                                  // element offset of row q's pixel from the tile's first pixel, -1: none
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      const int r = q * kHRows + rr;
      const int t = r / S, sx = r - t * S;
      roff[q] = (live && r < NB * 16) ? (t * w.L + sx) * kCO + ch0 : -1;
    }
    // per-channel constants (the formulas of conv3x3_stream.hip's table; BnAffine::init's folded affine decides the mask)
    float ksc[8], ksh[8], ka[8], kb2[8], kc0[8];
    (void)kc0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = ch0 + k;
      const float mu = p.mean[c], is = p.invstd[c], gmm = p.gamma[c];
      const float sc = is * gmm;
      ksc[k] = sc;
      ksh[k] = fmaf(-mu, sc, p.beta[c]);
      if (MODE == 1) {
        ka[k] = is;                                // xhat = z * is + mis
        kb2[k] = -mu * is;
      } else {
        const float sb = p.training ? p.loc[kCO + c] * p.inv_rows : 0.f;      // sum dzm / rows
        const float sg = p.training ? p.loc[c] * p.inv_rows : 0.f;            // sum dzm xhat / rows
        const float gi = gmm * is;
        ka[k] = gi;                                // out = A dzm + B z + C0
        kb2[k] = -gi * sg * is;
        kc0[k] = gi * (sg * is * mu - sb);
      }
    }
    float bs[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // z rows of the tile whose epilogue runs in the NEXT interval (one register set: interval i consumes tile i - 1's rows and
    // refills the set with tile i's)
    V8 zq0[NR];
    auto request_z = [&](int j, V8 (&zs)[NR]) {
      const E* zt = zg + pix_of(j) * kCO;
#pragma unroll
      for (int q = 0; q < NR; ++q) zs[q] = *reinterpret_cast<const V8*>(zt + (roff[q] >= 0 ? roff[q] : 0));
    };
    // the BatchNorm epilogue of tile j from its staging image and its z rows; -> stores this WAVE has certainly issued
    auto epilogue = [&](int j, const V8 (&zs)[NR]) -> int {
      const char* stage = smem + goff + (j & 1) * gb;
      E* ot = og + pix_of(j) * kCO;
      int issued = 0;
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        const int r = q * kHRows + rr;
        if (MODE == 2) issued += (q * kHRows + (hw * 64) / kCH < NB * 16 && hw * 64 < kHRows * kCH) ? 1 : 0;
        if (roff[q] >= 0) {
          const V4 lo = *reinterpret_cast<const V4*>(stage + r * kSPitch + c18 * 16);
          const V4 hi = *reinterpret_cast<const V4*>(stage + r * kSPitch + c18 * 16 + 8);
          V8 v = V8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          const V8 zv = zs[q];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float xf = (float)zv[k];
            const bool on = !p.relu || fmaf(xf, ksc[k], ksh[k]) > 0.f;
            const float dzv = on ? (float)v[k] : 0.f;
            if (MODE == 1) {
              bs[k] += dzv;
              bq[k] = fmaf(dzv, fmaf(xf, ka[k], kb2[k]), bq[k]);
            } else {
              v[k] = (E)fmaf(ka[k], dzv, fmaf(kb2[k], xf, kc0[k]));
            }
          }
          if (MODE == 2) *reinterpret_cast<V8*>(ot + roff[q]) = v;
        }
      }
      return issued;
    };
    load_window(0);
    if (n_my > 1) load_window(1);
    request_z(0, zq0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                              // P0: windows 0 and 1 have landed
    // interval i (the compute waves run tile i): the window request first -- issued behind the epilogue it cost 0.17 ms per
    // frametransformer step -- then the epilogue of tile i - 1, then the z rows of tile i into the set the epilogue has left.
    // (Two z sets, requested before the epilogue: the same time within the spread, and 33 spilled registers in the sums pass.)
    for (int i = 0; i < n_my; ++i) {
      const bool req = i + 2 < n_my;
      if (req) load_window(i + 2);                                // into the buffer tile i - 1 has left
      int nst = 0;
      if (i >= 1) nst = epilogue(i - 1, zq0);                     // (its z rows were requested an interval ago)
      if (i >= 1) request_z(i, zq0);                              // (tile 0's were requested in the prologue)
      // Window i + 1 (requested in interval i - 1) must have landed before the next interval computes from it.  vmcnt retires
      // in issue order: what this interval itself has issued -- its window requests, its stores, its z rows -- may stay in flight.
      db_wait_vm(nst + (req ? np : 0) + (i >= 1 ? NR : 0));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();                                            // B_{i+1}
    }
    epilogue(n_my - 1, zq0);
    if (MODE == 1) {
      // threads with equal c18 hold the same 8 channels: fixed-order sum over the 24 row lanes -> this workgroup's partial row
      float* red = reinterpret_cast<float*>(smem);                // [2][448][8] = 28 KiB over the windows (all reads are done)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        red[(0 * kNH * 64 + htid) * 8 + k] = bs[k];
        red[(1 * kNH * 64 + htid) * 8 + k] = bq[k];
      }
    }
    __syncthreads();                                              // (the compute waves join this one too)
    if (MODE == 1 && htid < 2 * kCO) {
      float* red = reinterpret_cast<float*>(smem);
      const int stat = htid / kCO, ch = htid - stat * kCO, cc = ch >> 3, k = ch & 7;
      float t = 0.f;
      for (int j = 0; j < kHRows; ++j) t += red[(stat * kNH * 64 + j * kCH + cc) * 8 + k];
      p.partial[((int64_t)blockIdx.x * 2 + stat) * kCO + ch] = t;
    }
    return;
  }

  // ================================================================== compute waves: wave u <-> output channels [16 u, 16 u + 16)
  const int g = lane >> 4, li = lane & 15;
  const int u = wid;
  V8 wf[3][2];                                     // lane (g, li) <-> weight row 16 u + li, k = tap * 64 + 32 kk + 8 g
  {
    const E* wrow = (const E*)p.w + (int64_t)(16 * u + li) * p.ldw;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) wf[kt][kk] = *reinterpret_cast<const V8*>(wrow + kt * kCI + kk * 32 + g * 8);
  }
  int xo[3];
#pragma unroll
  for (int kt = 0; kt < 3; ++kt) xo[kt] = (li + kt * S) * kXRow + (g << 4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                                // P0
  for (int i = 0; i < n_my; ++i) {
    const int cxo = (i % 3) * xb;
    char* const stage = smem + goff + (i & 1) * gb;
    // two position blocks per trip: independent accumulators keep the MFMA pipe fed while the next fragments arrive
#pragma unroll
    for (int b = 0; b < NB; b += 2) {
      f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = f32x4{0.f, 0.f, 0.f, 0.f};
      V8 x0[6], x1[6];
#pragma unroll
      for (int st = 0; st < 6; ++st) {
        x0[st] = *reinterpret_cast<const V8*>(smem + cxo + xo[st >> 1] + b * (16 * kXRow) + (st & 1) * 64);
        x1[st] = *reinterpret_cast<const V8*>(smem + cxo + xo[st >> 1] + (b + 1) * (16 * kXRow) + (st & 1) * 64);
      }
#pragma unroll
      for (int st = 0; st < 6; ++st) {
        a0 = Elem16<E>::mma(wf[st >> 1][st & 1], x0[st], a0);
        a1 = Elem16<E>::mma(wf[st >> 1][st & 1], x1[st], a1);
      }
      // lane (g, li) holds d[position 16 b + li][16 u + 4 g .. + 3] -> the staging image (the helpers finished with its previous
      // content, tile i - 2, before the barrier that opened this interval)
      V4 o0, o1;
#pragma unroll
      for (int r = 0; r < 4; ++r) { o0[r] = (E)a0[r]; o1[r] = (E)a1[r]; }
      *reinterpret_cast<V4*>(stage + (b * 16 + li) * kSPitch + u * 32 + g * 8) = o0;
      *reinterpret_cast<V4*>(stage + ((b + 1) * 16 + li) * kSPitch + u * 32 + g * 8) = o1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                              // B_{i+1}
  }
  __syncthreads();                                                // (the helpers' statistics scratch)
}

int db_plan(int T, int L, Win* q) {
  if (T < 1 || L < 1 || T + 2 > 2047) return 0;
  for (int S = 16; S >= 2; --S) {
    if (L % S || (T * S) % 32) continue;
    const int KP = T * S, xpos = (T + 2) * S;
    if (KP > kMaxRows * kHRows) continue;
    const int xbytes = (xpos * kXRow + 1023) & ~1023;
    if (3 * xbytes + 2 * KP * kSPitch + 2048 > 160 * 1024) continue;
    if ((xbytes >> 10) > kNH * kMaxXP) continue;
    if (3 * xbytes + 2 * KP * kSPitch < 2 * kNH * 64 * 8 * 4) continue;      // (the statistics scratch overlays the images)
    q->T = T; q->L = L; q->S = S; q->segs = L / S; q->KP = KP; q->xpos = xpos; q->x_bytes = xbytes;
    return 1;
  }
  return 0;
}

int db_grid(int64_t N, const Win& q) {
  const int64_t ntiles = N * q.segs;
  return (int)(ntiles < dvt_num_cus() ? ntiles : dvt_num_cus());
}

template <typename E, int NB, int MODE>
void db_launch(const DbParams& p, int grid, int lds, hipStream_t st) {
  static DvtLdsAttr set;
  dvt_lds_attr(set, (const void*)conv3x1_dbn_kernel<E, NB, MODE>, 160 * 1024);
  hipLaunchKernelGGL((conv3x1_dbn_kernel<E, NB, MODE>), dim3(grid), dim3(kNWv * 64), lds, st, p);
}

template <typename E, int MODE>
void db_dispatch(const DbParams& p, int grid, int lds, hipStream_t st) {
  switch (p.w_.KP >> 4) {
    case 2: db_launch<E, 2, MODE>(p, grid, lds, st); break;
    case 4: db_launch<E, 4, MODE>(p, grid, lds, st); break;
    default: db_launch<E, 6, MODE>(p, grid, lds, st); break;
  }
}

}  // namespace

namespace dvt_internal {

int conv3x1_dbn_supported(int64_t N, int T, int L, int dtype) {
#ifdef DVT_NO_DBN
  return 0;
#endif
  Win q;
  return N > 0 && dvt_is_16bit(dtype) && db_plan(T, L, &q) && N * q.segs < ((int64_t)1 << 31) && N * T * L * 144 < ((int64_t)1 << 31) ? 1 : 0;
}

int conv3x1_dbn_parts(int64_t N, int T, int L) {
  Win q;
  if (N <= 0 || !db_plan(T, L, &q)) return 0;
  return db_grid(N, q);
}

// mode 1: the sums -> partial [parts][2][144]; mode 2: the corrected gradient -> dz (loc = this launch pair's reduced sums)
int conv3x1_dbn_pass(int mode, const void* dy, const void* w, int64_t ldw, const void* z, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, int relu, int training, float* partial, const float* loc, void* dz,
                     int64_t N, int T, int L, int dtype, hipStream_t st) {
  DbParams p{};
  if (!db_plan(T, L, &p.w_)) return DVT_ERR_UNSUPPORTED;
  p.dy = dy; p.w = w; p.ldw = (int)ldw; p.z = z; p.dz = dz; p.partial = partial; p.loc = loc;
  p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.beta = beta; p.relu = relu; p.training = training;
  p.inv_rows = 1.0f / (float)(N * T * L);
  p.ntiles = (int)(N * p.w_.segs);
  const int grid = db_grid(N, p.w_);
  const int lds = 3 * p.w_.x_bytes + 2 * p.w_.KP * kSPitch;
  const bool h = dtype == DVT_F16;
  if (mode == 1) { h ? db_dispatch<f16, 1>(p, grid, lds, st) : db_dispatch<bf16, 1>(p, grid, lds, st); }
  else { h ? db_dispatch<f16, 2>(p, grid, lds, st) : db_dispatch<bf16, 2>(p, grid, lds, st); }
  return DVT_OK;
}

}  // namespace dvt_internal
