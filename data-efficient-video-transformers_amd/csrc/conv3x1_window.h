// conv3x1_window.h -- the LDS "window" of the (3, 1, 1) temporal convolutions of R(2+1)D-18's layer 1 (video_resnet.py:30-31;
// 144 mid planes <-> 64 planes), shared by the forward (conv3x1_fwd.hip) and the weight gradient (conv3x1_wgrad.hip).
//
// A workgroup owns a segment of S pixels of one clip over ALL its T frames and stages the (T + 2) x S x 144 window of the
// 144-plane map once (frames -1 and T are zero rows); the three taps are the position offsets 0 / S / 2S into it.
//   * image: [position][144 channels] in rows of TEN 32-byte units (320 bytes): nine channel blocks of 16 plus one unit of
//     padding, block cb of position k stored at unit cb + ((k >> 3) & 1).  A 32-lane half of ds_read_b64_tr_b16 touches the
//     positions {q, 8 + q} + 4 hf: 320-byte rows put positions q = 0 .. 3 on banks 0 / 16 / 32 / 48 (+ 8 each) and the
//     one-unit shift moves positions 8 + q to banks 8 / 24 / 40 / 56 -- the eight 8-bank groups of one LDS cycle.  (No
//     linear pitch does that: positions k and k + 8 are 8 rows apart, a multiple of 256 bytes for every 32-byte-multiple
//     pitch.)  The same image serves ds_read_b128 row fragments (16 positions x 16 bytes: four 16-lane access groups, each
//     over 64 distinct banks).  The DMA is lane-linear, so the layout lives on the per-lane source address: slot = 16
//     bytes, 20 slots per position, two of them zero padding.
//   * optional "virtual" BatchNorm: the map in HBM is the convolution output z of the layer in front;
//     the normalised activation y = relu(z * s + t) -- the formula and the rounding of dvt_bn_apply_fwd (BnAffine::apply,
//     conv.hip) -- is formed in the staged window, once per tile, by window_transform; the two zero rows stay zero.
#pragma once
#include "common.h"

namespace dvt_window {

constexpr int kCI = 144, kCO = 64, kNW = 8;
constexpr int kCB = kCI / 16;                // 9 input-channel blocks
constexpr int kXU = kCB + 1;                 // 32-byte units per position (one of padding)
constexpr int kXRow = kXU * 32;              // 320 bytes
constexpr int kMaxXP = 6;                    // window DMA pieces (1 KiB) per wave: window <= 48 KiB

// geometry of a launch (host: window_plan)
struct Window {
  int T, L, S, segs;    // frames, pixels per frame, pixels per segment, segments per frame
  int KP;               // T * S: positions of an output / gradient tile (multiple of 32)
  int xpos;             // (T + 2) * S: positions of a window
  int x_bytes;          // window image size (multiple of 1 KiB)
};

// the BatchNorm (+ ReLU) in front of the map, or mean == nullptr (the map already holds the normalised activation)
struct Affine {
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
  int c_valid, relu;
};

// (max_pieces: 1 KiB DMA pieces the workgroup's waves carry between them)
inline int window_plan(int T, int L, Window* q, int extra_bytes_per_buffer, int extra_bytes_once, int nbuf,
                       int max_pieces = kNW * kMaxXP) {
  if (T < 1 || L < 1 || T + 2 > 2047) return 0;
  for (int S = 16; S >= 2; --S) {
    if (L % S || (T * S) % 32) continue;
    const int KP = T * S, xpos = (T + 2) * S;
    const int xbytes = (xpos * kXRow + 1023) & ~1023;
    if (nbuf * (xbytes + extra_bytes_per_buffer * KP) + extra_bytes_once * KP + 4096 > 160 * 1024) continue;
    if ((xbytes >> 10) > max_pieces) continue;
    q->T = T; q->L = L; q->S = S; q->segs = L / S; q->KP = KP; q->xpos = xpos; q->x_bytes = xbytes;
    return 1;
  }
  return 0;
}

#ifdef __HIPCC__
__device__ __attribute__((aligned(16))) static unsigned int window_zero16[4] = {0u, 0u, 0u, 0u};

// per-lane coordinates of this wave's window DMA pieces (fixed for the launch):
// frame row << 20 | pixel of the segment << 8 | channel (multiple of 8) of the 16-byte chunk, bit 31 = never loaded
template <int MAXP = kMaxXP>
__device__ __forceinline__ void window_coords(const Window& w, int wid, int lane, unsigned (&xq)[MAXP]) {
  const int xp = w.x_bytes >> 10;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int piece = wid + kNW * i;
    const int sl = piece * 64 + lane;
    const int pos = sl / (2 * kXU), h = sl - pos * (2 * kXU);
    const int cb = (h >> 1) - ((pos >> 3) & 1);
    const int tt = pos / w.S, sx = pos - tt * w.S;
    const bool ok = piece < xp && pos < w.xpos && cb >= 0 && cb < kCB;
    xq[i] = ok ? ((unsigned)tt << 20) | ((unsigned)sx << 8) | (unsigned)(cb * 16 + (h & 1) * 8) : 0x80000000u;
  }
}

// request this wave's pieces of the window of (clip pixel base pix0 = frame 0, first pixel of the segment)
template <typename E, int MAXP = kMaxXP>
__device__ __forceinline__ void window_load(const Window& w, const E* xg, int64_t pix0, const unsigned (&xq)[MAXP], int wid,
                                            char* dst) {
  const int xp = w.x_bytes >> 10;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int piece = wid + kNW * i;
    if (piece < xp) {                            // wave-uniform
      const int frame = (int)((xq[i] >> 20) & 0x7FF) - 1;
      const bool ok = (int)xq[i] >= 0 && (unsigned)frame < (unsigned)w.T;
      const E* src = ok ? xg + (pix0 + (int64_t)frame * w.L + ((xq[i] >> 8) & 0xFFF)) * kCI + (xq[i] & 0xFF)
                        : reinterpret_cast<const E*>(window_zero16);
      dvt_dma16(src, dst + piece * 1024);
    }
  }
}

// The virtual BatchNorm of a staged window: y = relu?(z * s + t) with s = invstd * gamma, t = beta - mean * s (the folded
// affine of BnAffine::init, conv.hip) on every live 16-byte slot.  A thread owns ONE channel group for the whole launch
// (thread = 18 q + slot: channel block slot >> 1, half slot & 1; positions q, q + 28, q + 56 ...), so its 8 scales / shifts
// sit in registers and a pass is one LDS read, 8 fma / max, one LDS write -- the first form (any thread, any slot: two
// integer divisions and four table reads per slot) cost ~50 us of a 157 us launch.
constexpr int kTfSlots = 2 * kCB;                       // 18 live 16-byte slots per position
constexpr int kTfPos = (kNW * 64) / kTfSlots;           // 28 positions per pass (504 of the 512 threads)
struct AffineRegs {
  float s[8], t[8];
  int pos0, slot_off;                                    // first position (>= xpos: idle thread); byte offset of the slot in a row
};
// (tid: the thread's index among the kNW * 64 threads that transform -- threadIdx.x, or its index among the helper waves of
//  the pipelined forward)
__device__ __forceinline__ void window_affine_regs(const Affine& a, AffineRegs& r, int tid = -1) {
  if (tid < 0) tid = threadIdx.x;
  const int q = tid / kTfSlots, slot = tid - q * kTfSlots;
  const int c0 = (slot >> 1) * 16 + (slot & 1) * 8;
  r.pos0 = q < kTfPos ? q : (1 << 30);
  r.slot_off = slot * 16;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = c0 + k;
    const float g = c < a.c_valid ? a.gamma[c] : 0.f, b = c < a.c_valid ? a.beta[c] : 0.f;
    r.s[k] = a.invstd[c] * g;
    r.t[k] = fmaf(-a.mean[c], r.s[k], b);
  }
}

// (all threads; barriers are the caller's; the zero rows of frames -1 and T stay zero)
// Four positions per thread in flight; the arithmetic as packed pairs (v_pk_fma_f32) and the ReLU on the ROUNDED 16-bit
// pairs as a signed-integer maximum with zero (v_pk_max_i16: rounding keeps the sign, so relu(round(y)) == round(relu(y))
// bit for bit, -0 included): ~20 vector instructions per slot instead of ~45.  The launch did not get shorter for it
// (160 us either way in tools/dev/tf_probe.py): with the parts of the forward kernel switched off one at a time the phases of
// a tile ADD UP -- 27 us loop + barriers, 30 transform, 58 fragment reads + MFMAs, 28 epilogue, 28 exposed window requests
// of a 170 us launch -- i.e. a phase costs its barrier-to-barrier latency chain (LDS round trip, drain, barrier) rather than
// its instruction count, and one workgroup per CU has nothing to overlap it with.
template <typename E>
__device__ __forceinline__ void window_transform(const Window& w, char* img_, const AffineRegs& r, int relu) {
  using V8 = typename Elem16<E>::v8;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef short i16x8 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) char lds_char;     // (callers pick the buffer at run time: keep the accesses ds_*)
  typedef __attribute__((address_space(3))) V8 lds_v8;
  lds_char* const img = (lds_char*)img_;
  const int lo = w.S, hi = (w.T + 1) * w.S;
  constexpr int kU = 4;
  for (int pos0 = r.pos0; pos0 < hi; pos0 += kU * kTfPos) {
    V8 v[kU];
    lds_v8* at[kU];
    bool live[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int pos = pos0 + u * kTfPos;
      live[u] = pos >= lo && pos < hi;
      at[u] = (lds_v8*)(img + pos * kXRow + (((pos >> 3) & 1) << 5) + r.slot_off);
      if (live[u]) v[u] = *at[u];
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (!live[u]) continue;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x2 x = {(float)v[u][2 * k], (float)v[u][2 * k + 1]};
        const f32x2 y = __builtin_elementwise_fma(x, f32x2{r.s[2 * k], r.s[2 * k + 1]}, f32x2{r.t[2 * k], r.t[2 * k + 1]});
        v[u][2 * k] = (E)y[0];
        v[u][2 * k + 1] = (E)y[1];
      }
      if (relu) v[u] = __builtin_bit_cast(V8, __builtin_elementwise_max(__builtin_bit_cast(i16x8, v[u]), i16x8{0, 0, 0, 0, 0, 0, 0, 0}));
      *at[u] = v[u];
    }
  }
}
#endif  // __HIPCC__

}  // namespace dvt_window
