// attention.hip -- fused softmax(q k^T * scale) v, forward and backward.
// Scores are never written to HBM.
//
// 16-bit fast path (bf16 / fp16, dh == 64, Lk <= 608): one workgroup per (batch, head).
//   forward   K and V of the head are staged once in LDS as [key][64] images whose
//             32-byte units are XOR-swizzled by (key>>1)&3 -- conflict-free both
//             for ds_read_b128 row reads (K as the A operand of S^T = K Q^T) and
//             for ds_read_b64_tr_b16 transposed reads (V^T as the A operand of
//             O^T = V^T P^T).  Each wave owns 16-query tiles; the query sits on the
//             MFMA *column* (lane & 15), so a lane owns one query, and the S^T
//             accumulators are, converted to 16 bits, directly the B operand of
//             the P V product (k order permuted identically on the V^T side).
//             Lk <= 224: the whole 16 x Lk score tile stays in registers (one row
//             maximum, no rescale, no per-step cross-lane exchange); longer: online
//             softmax over 32-key steps.
//   backward  two kernels with the same anatomy, no atomics, bitwise reproducible:
//             dq   (query on the lane; recomputes S^T, dP^T; dQ^T = K^T dS^T; writes
//                   delta = rowsum(dO * O) to the workspace)
//             dkdv (key on the lane; recomputes S, dP; dV^T = dO^T P, dK^T = Q^T dS;
//                   reads lse and delta)
//             7 MFMA products instead of the minimal 5, in exchange for no
//             cross-wave reduction of any gradient.  Key / query loops are unrolled at
//             compile time up to 256 (template parameter), rolled beyond.
//   staging   all loads of a batch before the first LDS write (one memory round trip);
//   stores    output tiles transposed through a per-wave 2 KiB LDS patch and written as
//             whole 128-byte rows.
// generic path: any dtype / head dim / length that fits LDS, fp32 FMA, one wave
// per row (fp32 parity mode; dh = 448 / 224 / 256 / 32 heads of the reference's
// 14-token encoders; attention-probability dropout).
#include "common.h"
#include <cstdlib>

namespace {

// LDS hand-off between lanes of ONE wave (its LDS operations complete in order).
__device__ __forceinline__ void wave_lds_sync() {
  // LDS operations of one wave complete in order; only the compiler must not reorder
  // across this point.  (A wavefront-scope fence would also emit s_waitcnt vmcnt(0)
  // and serialise the wave behind its outstanding global stores.)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

struct AttnParams {
  const void *q, *k, *v, *o, *d_o;
  void *out, *dq, *dk, *dv;
  float* lse;
  float* delta;  // workspace [B*H*Lq] (generic bwd)
  int B, H, Lq, Lk, dh;
  int64_t q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, v_sb, v_sh, v_sl, o_sb, o_sh, o_sl;
  float scale;
  int Lkp;  // Lk rounded up to 32
  int Lqp;  // Lq rounded up to 32
  int patch_off;  // byte offset in LDS of the per-wave 2 KiB output patches (whole-row stores), -1: per-lane stores
  // attention-probability dropout (generic kernels only): keep iff Philox word >= drop_thr, survivors * drop_scale
  uint32_t drop_thr;
  float drop_scale;
  const uint64_t* rng;
  uint64_t rng_off;
};


// dropout multiplier of probability (b, h, i, j): 0 or 1 / (1 - p); 1 when dropout is off
__device__ __forceinline__ float attn_drop(const AttnParams& p, int b, int h, int i, int j) {
  if (p.drop_thr == 0u) return 1.0f;
  const uint64_t idx = (((uint64_t)b * p.H + h) * p.Lq + i) * (uint64_t)p.Lk + j;
  return dvt_dropout_keep(p.rng[0], p.rng[1] + p.rng_off, idx, p.drop_thr) ? p.drop_scale : 0.0f;
}

// ======================================================================= generic
// forward: one wave per (b, h, i).  LDS per wave: q row [dh] + probabilities [Lk].
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_generic_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float* qs = lds_f + (size_t)wid * (p.dh + p.Lk);
  float* ps = qs + p.dh;
  const int64_t row = (int64_t)blockIdx.x * 4 + wid;
  const int64_t total = (int64_t)p.B * p.H * p.Lq;
  if (row >= total) return;
  const int i = (int)(row % p.Lq);
  const int h = (int)((row / p.Lq) % p.H);
  const int b = (int)(row / ((int64_t)p.Lq * p.H));
  const T* q = (const T*)p.q + b * p.q_sb + h * p.q_sh + (int64_t)i * p.q_sl;
  const T* k = (const T*)p.k + b * p.k_sb + h * p.k_sh;
  const T* v = (const T*)p.v + b * p.v_sb + h * p.v_sh;
  T* o = (T*)p.out + b * p.o_sb + h * p.o_sh + (int64_t)i * p.o_sl;
  for (int e = lane; e < p.dh; e += 64) qs[e] = to_f32<T>(q[e]);
  wave_lds_sync();
  float m = -INFINITY;
  for (int j = lane; j < p.Lk; j += 64) {
    const T* kj = k + (int64_t)j * p.k_sl;
    float s = 0.f;
    for (int e = 0; e < p.dh; ++e) s = fmaf(qs[e], to_f32<T>(kj[e]), s);
    s *= p.scale;
    ps[j] = s;
    m = fmaxf(m, s);
  }
  m = wave_max(m);
  float l = 0.f;
  for (int j = lane; j < p.Lk; j += 64) {
    const float e = expf(ps[j] - m);
    ps[j] = e * attn_drop(p, b, h, i, j);          // dropped probabilities leave the normaliser untouched
    l += e;
  }
  l = wave_sum(l);
  wave_lds_sync();
  const float inv = 1.0f / l;
  for (int e = lane; e < p.dh; e += 64) {
    float acc = 0.f;
    for (int j = 0; j < p.Lk; ++j) acc = fmaf(ps[j], to_f32<T>(v[(int64_t)j * p.v_sl + e]), acc);
    o[e] = from_f32<T>(acc * inv);
  }
  if (lane == 0) p.lse[row] = m + logf(l);
}

// delta[b,h,i] = sum_e dO[i,e] * O[i,e]
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const AttnParams p) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wid;
  const int64_t total = (int64_t)p.B * p.H * p.Lq;
  if (row >= total) return;
  const int i = (int)(row % p.Lq);
  const int h = (int)((row / p.Lq) % p.H);
  const int b = (int)(row / ((int64_t)p.Lq * p.H));
  const int64_t off = b * p.o_sb + h * p.o_sh + (int64_t)i * p.o_sl;
  const T* o = (const T*)p.o + off;
  const T* g = (const T*)p.d_o + off;
  float s = 0.f;
  for (int e = lane; e < p.dh; e += 64) s = fmaf(to_f32<T>(o[e]), to_f32<T>(g[e]), s);
  s = wave_sum(s);
  if (lane == 0) p.delta[row] = s;
}

// dq: one wave per (b,h,i).  LDS per wave: q [dh], dO [dh], ds [Lk].
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_generic_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float* qs = lds_f + (size_t)wid * (2 * p.dh + p.Lk);
  float* gs = qs + p.dh;
  float* ds = gs + p.dh;
  const int64_t row = (int64_t)blockIdx.x * 4 + wid;
  const int64_t total = (int64_t)p.B * p.H * p.Lq;
  if (row >= total) return;
  const int i = (int)(row % p.Lq);
  const int h = (int)((row / p.Lq) % p.H);
  const int b = (int)(row / ((int64_t)p.Lq * p.H));
  const T* q = (const T*)p.q + b * p.q_sb + h * p.q_sh + (int64_t)i * p.q_sl;
  const T* g = (const T*)p.d_o + b * p.o_sb + h * p.o_sh + (int64_t)i * p.o_sl;
  const T* k = (const T*)p.k + b * p.k_sb + h * p.k_sh;
  const T* v = (const T*)p.v + b * p.v_sb + h * p.v_sh;
  T* dq = (T*)p.dq + b * p.q_sb + h * p.q_sh + (int64_t)i * p.q_sl;
  for (int e = lane; e < p.dh; e += 64) {
    qs[e] = to_f32<T>(q[e]);
    gs[e] = to_f32<T>(g[e]);
  }
  wave_lds_sync();
  const float lse = p.lse[row], delta = p.delta[row];
  for (int j = lane; j < p.Lk; j += 64) {
    const T* kj = k + (int64_t)j * p.k_sl;
    const T* vj = v + (int64_t)j * p.v_sl;
    float s = 0.f, dp = 0.f;
    for (int e = 0; e < p.dh; ++e) {
      s = fmaf(qs[e], to_f32<T>(kj[e]), s);
      dp = fmaf(gs[e], to_f32<T>(vj[e]), dp);
    }
    const float pr = expf(s * p.scale - lse);
    ds[j] = pr * (dp * attn_drop(p, b, h, i, j) - delta) * p.scale;
  }
  wave_lds_sync();
  for (int e = lane; e < p.dh; e += 64) {
    float acc = 0.f;
    for (int j = 0; j < p.Lk; ++j) acc = fmaf(ds[j], to_f32<T>(k[(int64_t)j * p.k_sl + e]), acc);
    dq[e] = from_f32<T>(acc);
  }
}

// dk, dv: one wave per (b,h,j).  LDS per wave: k [dh], v [dh], p [Lq], ds [Lq].
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv_generic_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float* ks = lds_f + (size_t)wid * (2 * p.dh + 2 * p.Lq);
  float* vs = ks + p.dh;
  float* ps = vs + p.dh;
  float* ds = ps + p.Lq;
  const int64_t row = (int64_t)blockIdx.x * 4 + wid;
  const int64_t total = (int64_t)p.B * p.H * p.Lk;
  if (row >= total) return;
  const int j = (int)(row % p.Lk);
  const int h = (int)((row / p.Lk) % p.H);
  const int b = (int)(row / ((int64_t)p.Lk * p.H));
  const T* kj = (const T*)p.k + b * p.k_sb + h * p.k_sh + (int64_t)j * p.k_sl;
  const T* vj = (const T*)p.v + b * p.v_sb + h * p.v_sh + (int64_t)j * p.v_sl;
  const T* q = (const T*)p.q + b * p.q_sb + h * p.q_sh;
  const T* g = (const T*)p.d_o + b * p.o_sb + h * p.o_sh;
  T* dk = (T*)p.dk + b * p.k_sb + h * p.k_sh + (int64_t)j * p.k_sl;
  T* dv = (T*)p.dv + b * p.v_sb + h * p.v_sh + (int64_t)j * p.v_sl;
  const float* lse = p.lse + ((int64_t)b * p.H + h) * p.Lq;
  const float* delta = p.delta + ((int64_t)b * p.H + h) * p.Lq;
  for (int e = lane; e < p.dh; e += 64) {
    ks[e] = to_f32<T>(kj[e]);
    vs[e] = to_f32<T>(vj[e]);
  }
  wave_lds_sync();
  for (int i = lane; i < p.Lq; i += 64) {
    const T* qi = q + (int64_t)i * p.q_sl;
    const T* gi = g + (int64_t)i * p.o_sl;
    float s = 0.f, dp = 0.f;
    for (int e = 0; e < p.dh; ++e) {
      s = fmaf(to_f32<T>(qi[e]), ks[e], s);
      dp = fmaf(to_f32<T>(gi[e]), vs[e], dp);
    }
    const float pr = expf(s * p.scale - lse[i]);
    const float dm = attn_drop(p, b, h, i, j);
    ps[i] = pr * dm;
    ds[i] = pr * (dp * dm - delta[i]) * p.scale;
  }
  wave_lds_sync();
  for (int e = lane; e < p.dh; e += 64) {
    float av = 0.f, ak = 0.f;
    for (int i = 0; i < p.Lq; ++i) {
      av = fmaf(ps[i], to_f32<T>(g[(int64_t)i * p.o_sl + e]), av);
      ak = fmaf(ds[i], to_f32<T>(q[(int64_t)i * p.q_sl + e]), ak);
    }
    dv[e] = from_f32<T>(av);
    dk[e] = from_f32<T>(ak);
  }
}

// ======================================================================= short sequences, any head width
// The reference's default FrameTransformer runs a 4-layer encoder over 14 + 1 chunk tokens (frame_transformer.py:83-121;
// d = 896, 8 heads of 112): the generic kernels above give every (b, h, query) a wave whose lanes walk dh serially with
// 2-byte loads -- 25 us per launch for 392 dot products of 448 (d = 896 in 2 heads), and the backward is three launches
// (delta, dq, dk / dv).  Here ONE WORKGROUP owns a whole (b, h): Q, K, V (and dO) are staged in LDS as fp32 rows of dh + 1
// floats (an odd pitch: the lanes of a score pass read different rows at the same column, conflict-free), the Lq x Lk
// score matrix lives in LDS, and the backward recomputes the softmax and produces dQ, dK and dV in the same launch (no lse,
// no delta workspace).  Same arithmetic as the generic kernels: fp32 FMA, the dropout multiplier of attn_drop on the
// normalised probabilities.  LDS per workgroup: (3 or 4) L (dh + 1) + (1 or 2) L (L + 1) floats; L <= 32, dh <= 512.
constexpr int kSmallL = 32, kSmallDh = 512;

// rows of a [L][dh] operand (row stride sl elements) -> fp32 LDS rows of pitch dhp (a multiple of 4 floats).  16-byte
// loads, four in flight per thread, when the rows allow it: one 2-byte load per trip of a rolled loop serialised a DRAM
// round trip per element.
template <typename T>
__device__ __forceinline__ void small_stage(float* dst, const T* src, int L, int dh, int64_t sl, int dhp, int tid, int NT) {
  constexpr int VE = 16 / (int)sizeof(T);                       // elements per 16-byte load
  typedef T vec_t __attribute__((ext_vector_type(VE)));
  const bool vec = dh % VE == 0 && sl % VE == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
  if (vec) {
    const int vpr = dh / VE, nv = L * vpr;
    for (int base = 0; base < nv; base += 4 * NT) {
      vec_t v[4];
      int r[4], c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = base + u * NT + tid;
        r[u] = idx / vpr; c[u] = (idx - r[u] * vpr) * VE;
        if (idx < nv) v[u] = *reinterpret_cast<const vec_t*>(src + (int64_t)r[u] * sl + c[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (base + u * NT + tid < nv) {
#pragma unroll
          for (int k = 0; k < VE; k += 4)
            *reinterpret_cast<f32x4*>(dst + r[u] * dhp + c[u] + k) =
                f32x4{to_f32<T>(v[u][k]), to_f32<T>(v[u][k + 1]), to_f32<T>(v[u][k + 2]), to_f32<T>(v[u][k + 3])};
        }
      }
    }
  } else {
    for (int idx = tid; idx < L * dh; idx += NT) {
      const int i = idx / dh, e = idx - i * dh;
      dst[i * dhp + e] = to_f32<T>(src[(int64_t)i * sl + e]);
    }
  }
  for (int idx = tid; idx < L * (dhp - dh); idx += NT) {          // the pad columns are read by the 4-wide dot products
    const int i = idx / (dhp - dh), e = dh + idx - i * (dhp - dh);
    dst[i * dhp + e] = 0.f;
  }
}

// pitches: operand rows dh rounded up to 4 floats + 4 (16-byte aligned rows for ds_read_b128, consecutive rows 4 banks
// apart); score rows Lk rounded up to 4 floats + 4
__host__ __device__ inline int small_dhp(int dh) { return ((dh + 3) & ~3) + 4; }
__host__ __device__ inline int small_lp(int Lk) { return ((Lk + 3) & ~3) + 4; }

// scores of all (i, j) pairs: a pair's dot product is cut into P = 1, 2 or 4 column ranges so that every thread of the
// workgroup has one (196 pairs on 512 threads: two threads per pair); partial sums into part[P][Lq][LP].  16-byte LDS
// reads: with 4-byte reads the 448-long products of the frametransformer's encoder were 5.5 us of LDS instruction issue.
template <bool WITH_DP>
__device__ __forceinline__ void small_scores(const float* qs, const float* ks, const float* gs, const float* vs, float* part,
                                             float* dpart, int Lq, int Lk, int dh, int dhp, int LP, int P, int tid, int NT) {
  const int npair = Lq * Lk, span = (((dh + P - 1) / P) + 3) & ~3, dh4 = (dh + 3) & ~3;
  for (int idx = tid; idx < npair * P; idx += NT) {
    const int pr = idx / P, pt = idx - pr * P;
    const int i = pr / Lk, j = pr - i * Lk;
    const int e0 = pt * span, e1 = min(dh4, e0 + span);
    const float* qi = qs + i * dhp;
    const float* kj = ks + j * dhp;
    const float* gi = WITH_DP ? gs + i * dhp : nullptr;
    const float* vj = WITH_DP ? vs + j * dhp : nullptr;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
    for (int e = e0; e < e1; e += 4) {                             // (pad columns are zero)
      const f32x4 a = *reinterpret_cast<const f32x4*>(qi + e), c = *reinterpret_cast<const f32x4*>(kj + e);
#pragma unroll
      for (int u = 0; u < 4; ++u) s[u] = fmaf(a[u], c[u], s[u]);
      if (WITH_DP) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(gi + e), y = *reinterpret_cast<const f32x4*>(vj + e);
#pragma unroll
        for (int u = 0; u < 4; ++u) d[u] = fmaf(x[u], y[u], d[u]);
      }
    }
    part[(pt * Lq + i) * LP + j] = (s[0] + s[1]) + (s[2] + s[3]);
    if (WITH_DP) dpart[(pt * Lq + i) * LP + j] = (d[0] + d[1]) + (d[2] + d[3]);
  }
}

// column ranges per pair: as many as give every thread work, at most 4 (the partial matrices live in LDS)
__host__ __device__ inline int small_parts(int Lq, int Lk, int NT) {
  const int np = Lq * Lk;
  return np * 4 <= NT ? 4 : np * 2 <= NT ? 2 : 1;
}

// out[r][e] = sum_c m[r][c] * col[c] for this thread's column e: the matrix rows (pitch LP, zero beyond nc) are broadcast
// from LDS four at a time, the column sits in registers
#define DVT_SMALL_MATCOL(acc, mrow, col, nc)                                         \
  do {                                                                               \
    f32x4 t_ = {0.f, 0.f, 0.f, 0.f};                                                 \
    _Pragma("unroll") for (int c4_ = 0; c4_ < kSmallL; c4_ += 4) {                   \
      if (c4_ < (nc)) {                                                              \
        const f32x4 m_ = *reinterpret_cast<const f32x4*>((mrow) + c4_);              \
        t_[0] = fmaf(m_[0], (col)[c4_], t_[0]); t_[1] = fmaf(m_[1], (col)[c4_ + 1], t_[1]); \
        t_[2] = fmaf(m_[2], (col)[c4_ + 2], t_[2]); t_[3] = fmaf(m_[3], (col)[c4_ + 3], t_[3]); \
      }                                                                              \
    }                                                                                \
    (acc) = (t_[0] + t_[1]) + (t_[2] + t_[3]);                                       \
  } while (0)

template <typename T>
__global__ __launch_bounds__(512) void attn_small_fwd_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  const int tid = threadIdx.x, NT = blockDim.x;
  const int64_t bh = blockIdx.x;
  const int b = (int)(bh / p.H), h = (int)(bh % p.H);
  const int dhp = small_dhp(p.dh), LP = small_lp(p.Lk), P = small_parts(p.Lq, p.Lk, NT);
  float* qs = lds_f;
  float* ks = qs + p.Lq * dhp;
  float* vs = ks + p.Lk * dhp;
  float* sc = vs + p.Lk * dhp;                      // [P][Lq][LP] partial scores; [0] becomes the probabilities
  const T* q = (const T*)p.q + b * p.q_sb + h * p.q_sh;
  const T* k = (const T*)p.k + b * p.k_sb + h * p.k_sh;
  const T* v = (const T*)p.v + b * p.v_sb + h * p.v_sh;
  T* o = (T*)p.out + b * p.o_sb + h * p.o_sh;
  small_stage<T>(qs, q, p.Lq, p.dh, p.q_sl, dhp, tid, NT);
  small_stage<T>(ks, k, p.Lk, p.dh, p.k_sl, dhp, tid, NT);
  small_stage<T>(vs, v, p.Lk, p.dh, p.v_sl, dhp, tid, NT);
  __syncthreads();
  small_scores<false>(qs, ks, nullptr, nullptr, sc, nullptr, p.Lq, p.Lk, p.dh, dhp, LP, P, tid, NT);
  __syncthreads();
  // softmax: 32 lanes per query row (key j on lane j; Lk <= 32), the row's maximum / sum by xor-shuffles inside the half wave
  for (int i = tid >> 5; i < p.Lq; i += NT >> 5) {
    const int j = tid & 31;
    float sv = -INFINITY;
    if (j < p.Lk) {
      sv = 0.f;
      for (int pt = 0; pt < P; ++pt) sv += sc[(pt * p.Lq + i) * LP + j];
      sv *= p.scale;
    }
    float m = sv;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 32));
    const float e = j < p.Lk ? expf(sv - m) : 0.f;
    float l = e;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) l += __shfl_xor(l, off, 32);
    // (dropped probabilities leave the normaliser untouched; columns Lk .. LP - 1 of a row are zero for the 4-wide reads)
    if (j < LP) sc[i * LP + j] = j < p.Lk ? e * attn_drop(p, b, h, i, j) * (1.0f / l) : 0.f;
    if (j == 0) p.lse[bh * p.Lq + i] = m + logf(l);
  }
  __syncthreads();
  // O = P V: a thread owns an output column e, V's column in registers, the probabilities broadcast from LDS
  for (int e = tid; e < p.dh; e += NT) {
    float vc[kSmallL];
#pragma unroll
    for (int j = 0; j < kSmallL; ++j) vc[j] = j < p.Lk ? vs[j * dhp + e] : 0.f;
    for (int i = 0; i < p.Lq; ++i) {
      float acc;
      DVT_SMALL_MATCOL(acc, sc + i * LP, vc, p.Lk);
      o[(int64_t)i * p.o_sl + e] = from_f32<T>(acc);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(512) void attn_small_bwd_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  const int tid = threadIdx.x, NT = blockDim.x;
  const int64_t bh = blockIdx.x;
  const int b = (int)(bh / p.H), h = (int)(bh % p.H);
  const int dhp = small_dhp(p.dh), LP = small_lp(p.Lk), LQ = small_lp(p.Lq), P = small_parts(p.Lq, p.Lk, NT);
  float* qs = lds_f;
  float* gs = qs + p.Lq * dhp;                       // dO
  float* ks = gs + p.Lq * dhp;
  float* vs = ks + p.Lk * dhp;
  float* pm = vs + p.Lk * dhp;                       // [P][Lq][LP]: partial scores; [0]: dS
  float* dm = pm + P * p.Lq * LP;                    // [P][Lq][LP]: partial dP
  float* pt_ = dm + P * p.Lq * LP;                   // [Lk][LQ]: dropped probabilities, transposed (dV's operand)
  float* dt_ = pt_ + p.Lk * LQ;                      // [Lk][LQ]: dS, transposed (dK's operand)
  const T* q = (const T*)p.q + b * p.q_sb + h * p.q_sh;
  const T* k = (const T*)p.k + b * p.k_sb + h * p.k_sh;
  const T* v = (const T*)p.v + b * p.v_sb + h * p.v_sh;
  const T* g = (const T*)p.d_o + b * p.o_sb + h * p.o_sh;
  T* dq = (T*)p.dq + b * p.q_sb + h * p.q_sh;
  T* dk = (T*)p.dk + b * p.k_sb + h * p.k_sh;
  T* dv = (T*)p.dv + b * p.v_sb + h * p.v_sh;
  small_stage<T>(qs, q, p.Lq, p.dh, p.q_sl, dhp, tid, NT);
  small_stage<T>(gs, g, p.Lq, p.dh, p.o_sl, dhp, tid, NT);
  small_stage<T>(ks, k, p.Lk, p.dh, p.k_sl, dhp, tid, NT);
  small_stage<T>(vs, v, p.Lk, p.dh, p.v_sl, dhp, tid, NT);
  for (int idx = tid; idx < 2 * p.Lk * LQ; idx += NT) pt_[idx] = 0.f;       // (pad columns of the transposed matrices)
  __syncthreads();
  small_scores<true>(qs, ks, gs, vs, pm, dm, p.Lq, p.Lk, p.dh, dhp, LP, P, tid, NT);
  __syncthreads();
  for (int i = tid >> 5; i < p.Lq; i += NT >> 5) {   // 32 lanes per query row
    const int j = tid & 31;
    float sv = -INFINITY, dp = 0.f;
    if (j < p.Lk) {
      sv = 0.f;
      for (int pt = 0; pt < P; ++pt) {
        sv += pm[(pt * p.Lq + i) * LP + j];
        dp += dm[(pt * p.Lq + i) * LP + j];
      }
      sv *= p.scale;
    }
    float m = sv;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 32));
    const float e = j < p.Lk ? expf(sv - m) : 0.f;
    float l = e;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) l += __shfl_xor(l, off, 32);
    const float pr = e * (1.0f / l);
    const float dr = j < p.Lk ? attn_drop(p, b, h, i, j) : 0.f;
    const float dpd = dp * dr;                       // dP * drop
    float delta = pr * dpd;                          // sum_j: = sum_e dO[i][e] O[i][e]
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) delta += __shfl_xor(delta, off, 32);
    const float dsv = j < p.Lk ? pr * (dpd - delta) * p.scale : 0.f;
    if (j < LP) pm[i * LP + j] = dsv;                // dS (rows: dQ's operand), zero beyond Lk
    if (j < p.Lk) {
      dt_[j * LQ + i] = dsv;
      pt_[j * LQ + i] = pr * dr;                     // dropped probability
    }
  }
  __syncthreads();
  // a thread owns a column e of the head: dQ[:, e] from K's column, dK[:, e] / dV[:, e] from Q's and dO's columns
  for (int e = tid; e < p.dh; e += NT) {
    float c0[kSmallL], c1[kSmallL];
#pragma unroll
    for (int j = 0; j < kSmallL; ++j) c0[j] = j < p.Lk ? ks[j * dhp + e] : 0.f;
    for (int i = 0; i < p.Lq; ++i) {
      float acc;
      DVT_SMALL_MATCOL(acc, pm + i * LP, c0, p.Lk);
      dq[(int64_t)i * p.q_sl + e] = from_f32<T>(acc);
    }
#pragma unroll
    for (int i = 0; i < kSmallL; ++i) {
      c0[i] = i < p.Lq ? gs[i * dhp + e] : 0.f;
      c1[i] = i < p.Lq ? qs[i * dhp + e] : 0.f;
    }
    for (int j = 0; j < p.Lk; ++j) {
      float av, ak;
      DVT_SMALL_MATCOL(av, pt_ + j * LQ, c0, p.Lq);
      DVT_SMALL_MATCOL(ak, dt_ + j * LQ, c1, p.Lq);
      dv[(int64_t)j * p.v_sl + e] = from_f32<T>(av);
      dk[(int64_t)j * p.k_sl + e] = from_f32<T>(ak);
    }
  }
}
#undef DVT_SMALL_MATCOL

// ======================================================================= one query per (b, h), dh = 64
// The last layer of a stack is read at row 0 only (src/models/vit.py:119-120, :126), so its attention has ONE query per
// (sequence, head): a matrix-vector problem, bound by streaming K and V once (HBM), no MFMA.  One wave per (b, h): a
// wave-instruction covers 8 keys x 8 lanes, a lane holds 8 of the 64 head channels (16-byte loads: eight whole 128-byte
// key rows per instruction); the 64-channel dot products close over the 8 lanes of a key with three xor-shuffles.  The
// scores pass through LDS (Lk floats per wave) between the score pass and the weighted V sum; U key groups are requested
// before the first is consumed.
constexpr int kQ1Unroll = 4;

template <typename E>
__global__ __launch_bounds__(256) void attn_fwd_q1_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t bh = (int64_t)blockIdx.x * 4 + wid;
  if (bh >= (int64_t)p.B * p.H) return;
  float* sc = lds_f + (size_t)wid * p.Lkp;
  const int b = (int)(bh / p.H), h = (int)(bh % p.H);
  const int g = lane >> 3, c = (lane & 7) * 8;
  const E* q = (const E*)p.q + b * p.q_sb + h * p.q_sh + c;
  const E* k = (const E*)p.k + b * p.k_sb + h * p.k_sh + c;
  const E* v = (const E*)p.v + b * p.v_sb + h * p.v_sh + c;
  float qv[8];
  load8<E>(q, qv);
#pragma unroll
  for (int e = 0; e < 8; ++e) qv[e] *= p.scale;
  const int groups = (p.Lk + 7) >> 3;
  float m = -INFINITY;
  for (int i0 = 0; i0 < groups; i0 += kQ1Unroll) {
    float kv[kQ1Unroll][8];
#pragma unroll
    for (int u = 0; u < kQ1Unroll; ++u) {
      const int j = min((i0 + u) * 8 + g, p.Lk - 1);            // clamped: the surplus keys are dropped below
      load8<E>(k + (int64_t)j * p.k_sl, kv[u]);
    }
#pragma unroll
    for (int u = 0; u < kQ1Unroll; ++u) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s = fmaf(qv[e], kv[u][e], s);
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      const int j = (i0 + u) * 8 + g;
      if (j < p.Lk) {
        m = fmaxf(m, s);
        if ((lane & 7) == 0) sc[j] = s;
      }
    }
  }
  m = wave_max(m);
  wave_lds_sync();
  float l = 0.f, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i0 = 0; i0 < groups; i0 += kQ1Unroll) {
    float vv[kQ1Unroll][8];
#pragma unroll
    for (int u = 0; u < kQ1Unroll; ++u) {
      const int j = min((i0 + u) * 8 + g, p.Lk - 1);
      load8<E>(v + (int64_t)j * p.v_sl, vv[u]);
    }
#pragma unroll
    for (int u = 0; u < kQ1Unroll; ++u) {
      const int j = (i0 + u) * 8 + g;
      const float pr = j < p.Lk ? __expf(sc[min(j, p.Lk - 1)] - m) : 0.f;
      l += pr;                                                   // (the same value on the 8 lanes of a key)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(pr, vv[u][e], acc[e]);
    }
  }
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    l += __shfl_xor(l, o, 64);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
  }
  if (g == 0) {
    const float inv = 1.0f / l;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= inv;
    store8<E>((E*)p.out + b * p.o_sb + h * p.o_sh + c, acc);
    if (lane == 0) p.lse[bh] = m + __logf(l);
  }
}

// backward of the same: dV_j = p_j dO, dS_j = p_j (dO . v_j - delta), dK_j = scale dS_j q, dq = scale sum_j dS_j k_j --
// one pass over K and V, every dK / dV row written once (whole 128-byte rows), no atomics.
template <typename E>
__global__ __launch_bounds__(256) void attn_bwd_q1_kernel(const AttnParams p) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t bh = (int64_t)blockIdx.x * 4 + wid;
  if (bh >= (int64_t)p.B * p.H) return;
  const int b = (int)(bh / p.H), h = (int)(bh % p.H);
  const int g = lane >> 3, c = (lane & 7) * 8;
  const int64_t qoff = b * p.q_sb + h * p.q_sh + c, ooff = b * p.o_sb + h * p.o_sh + c;
  const int64_t koff = b * p.k_sb + h * p.k_sh + c, voff = b * p.v_sb + h * p.v_sh + c;
  float qv[8], gv[8], ov[8];
  load8<E>((const E*)p.q + qoff, qv);
  load8<E>((const E*)p.d_o + ooff, gv);
  load8<E>((const E*)p.o + ooff, ov);
  float delta = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) delta = fmaf(gv[e], ov[e], delta);
  delta += __shfl_xor(delta, 1, 64);
  delta += __shfl_xor(delta, 2, 64);
  delta += __shfl_xor(delta, 4, 64);
  const float lse = p.lse[bh];
  const E* k = (const E*)p.k + koff;
  const E* v = (const E*)p.v + voff;
  E* dk = (E*)p.dk + koff;
  E* dv = (E*)p.dv + voff;
  float dq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int groups = (p.Lk + 7) >> 3;
  for (int i0 = 0; i0 < groups; i0 += kQ1Unroll) {
    float kv[kQ1Unroll][8], vv[kQ1Unroll][8];
#pragma unroll
    for (int u = 0; u < kQ1Unroll; ++u) {
      const int j = min((i0 + u) * 8 + g, p.Lk - 1);
      load8<E>(k + (int64_t)j * p.k_sl, kv[u]);
      load8<E>(v + (int64_t)j * p.v_sl, vv[u]);
    }
#pragma unroll
    for (int u = 0; u < kQ1Unroll; ++u) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s = fmaf(qv[e], kv[u][e], s);
        dp = fmaf(gv[e], vv[u][e], dp);
      }
      s += __shfl_xor(s, 1, 64);   dp += __shfl_xor(dp, 1, 64);
      s += __shfl_xor(s, 2, 64);   dp += __shfl_xor(dp, 2, 64);
      s += __shfl_xor(s, 4, 64);   dp += __shfl_xor(dp, 4, 64);
      const int j = (i0 + u) * 8 + g;
      if (j < p.Lk) {
        const float pr = __expf(fmaf(s, p.scale, -lse));
        const float ds = pr * (dp - delta) * p.scale;
        float ok[8], ovv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          ok[e] = ds * qv[e];
          ovv[e] = pr * gv[e];
          dq[e] = fmaf(ds, kv[u][e], dq[e]);
        }
        store8<E>(dk + (int64_t)j * p.k_sl, ok);
        store8<E>(dv + (int64_t)j * p.v_sl, ovv);
      }
    }
  }
#pragma unroll
  for (int o = 8; o < 64; o <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) dq[e] += __shfl_xor(dq[e], o, 64);
  if (g == 0) store8<E>((E*)p.dq + qoff, dq);
}

// ======================================================================= MFMA bf16, dh = 64
constexpr int DH = 64;
constexpr int kRowBytes = DH * 2;  // 128

// byte offset of 16-byte chunk c (0..7) of row r in a dual-use [rows][64] image
__device__ __forceinline__ int img_off(int r, int c) {
  return r * kRowBytes + ((((c >> 1) ^ ((r >> 1) & 3)) << 5) | ((c & 1) << 4));
}

// stage rows [0, rows_pad) of two [L][64] matrices (row strides sla / slb) into their images; rows >= L are 0.
// All global loads of a batch (4 chunks per thread and image) are issued before the first LDS write, so a batch costs one
// memory round trip (a load -> wait -> write loop costs one per 16-byte chunk).
template <typename E>
__device__ __forceinline__ void stage_images(char* imga, const E* srca, int64_t sla, char* imgb, const E* srcb,
                                             int64_t slb, int L, int rows_pad) {
  using V8 = typename Elem16<E>::v8;
  constexpr int kBatch = 4;
  const int total = rows_pad * 8, nt = blockDim.x;
  for (int base = threadIdx.x; base < total; base += kBatch * nt) {
    V8 va[kBatch], vb[kBatch];
    // straight-line code: chunk indices past the end clamp to the last chunk (the same data written twice), rows >= L load
    // a clamped row and are zeroed afterwards -- a branch around a load or a store makes the compiler wait per chunk
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const int idx = min(base + u * nt, total - 1);
      const int r = idx >> 3, c = idx & 7;
      const int rr = r < L ? r : 0;
      const V8 a = *reinterpret_cast<const V8*>(srca + (int64_t)rr * sla + c * 8);
      const V8 b = *reinterpret_cast<const V8*>(srcb + (int64_t)rr * slb + c * 8);
      va[u] = a;
      vb[u] = b;
    }
    __builtin_amdgcn_sched_barrier(0);               // all 2 * kBatch loads in flight before the first LDS write (under
                                                     // register pressure the scheduler otherwise interleaves them in pairs)
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const int idx = min(base + u * nt, total - 1);
      const bool live = (idx >> 3) < L;
      *reinterpret_cast<V8*>(imga + img_off(idx >> 3, idx & 7)) = live ? va[u] : V8{0, 0, 0, 0, 0, 0, 0, 0};
      *reinterpret_cast<V8*>(imgb + img_off(idx >> 3, idx & 7)) = live ? vb[u] : V8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
}

// row fragment: lane (g, li) gets M[row0 + li][kk*32 + 8g .. +7]   (MFMA A/B operand, k = feature)
template <typename E>
__device__ __forceinline__ typename Elem16<E>::v8 img_row_frag(const char* img, int row0, int kk, int g, int li) {
  return *reinterpret_cast<const typename Elem16<E>::v8*>(img + img_off(row0 + li, kk * 4 + g));
}

// transposed fragment for a k-step whose 32 reduction rows are the two 16-row tiles
// ta, tb:  element j of lane (g, li) = M[16*(j<4 ? ta : tb) + 4g + (j&3)][col0 + li]
// (col0 multiple of 16).  This is the k order of an accumulator pair used as the
// other operand (see file header).
template <typename E>
__device__ __forceinline__ typename Elem16<E>::v8 img_tr_frag(const char* img, int ta, int tb, int col0, int g,
                                              int li) {
  const int qq = li >> 2, pp = li & 3;
  const int u = col0 >> 4;
  typename Elem16<E>::v4 half[2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const int r = 16 * (hf ? tb : ta) + 4 * g + qq;
    const char* a = img + r * kRowBytes + ((u ^ ((r >> 1) & 3)) << 5) + 8 * pp;
    half[hf] = Elem16<E>::tr_read(a);
  }
  // concatenation, not element inserts: the two 64-bit reads land in adjacent VGPR pairs
  return __builtin_shufflevector(half[0], half[1], 0, 1, 2, 3, 4, 5, 6, 7);
}

template <typename E>
__device__ __forceinline__ typename Elem16<E>::v8 pack_pair(const f32x4& a, const f32x4& b) {
  typename Elem16<E>::v8 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    o[r] = (E)a[r];
    o[4 + r] = (E)b[r];
  }
  return o;
}

template <typename E>
__device__ __forceinline__ void store4(E* p, const f32x4& v, float s) {
  typename Elem16<E>::v4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = (E)(v[r] * s);
  *reinterpret_cast<typename Elem16<E>::v4*>(p) = o;
}

// A wave's 16 x 64 output tile (lane (g, li): row li, columns dt*16 + 4g .. +3 of accumulator dt), scaled, as whole
// 128-byte rows: transposed through a wave-private 2 KiB LDS patch (16-byte chunks XOR-swizzled by (row >> 1) & 7) and
// written with two fully coalesced 16-byte-per-lane stores.  The per-lane form (8 bytes at a row stride: sixteen 32-byte
// fragments per instruction) costs ~150 cycles of the CU's memory pipeline per instruction (measured with per-wave s_memtime stamps).
template <typename E>
__device__ __forceinline__ void store_tile(char* patch, E* dst, int64_t row_stride, int row0, int row_lim,
                                           const f32x4* acc, float scale, int lane) {
  using V4 = typename Elem16<E>::v4;
  using V8 = typename Elem16<E>::v8;
  const int g = lane >> 4, li = lane & 15;
  if (patch == nullptr) {
    if (row0 + li < row_lim) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) store4<E>(dst + (int64_t)(row0 + li) * row_stride + dt * 16 + 4 * g, acc[dt], scale);
    }
    return;
  }
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    V4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (E)(acc[dt][r] * scale);
    const int chunk = (dt * 2 + (g >> 1)) ^ ((li >> 1) & 7);
    *reinterpret_cast<V4*>(patch + li * 128 + chunk * 16 + (g & 1) * 8) = o;
  }
  wave_lds_sync();
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int r = ps * 8 + (lane >> 3), c = lane & 7;
    const V8 v = *reinterpret_cast<const V8*>(patch + r * 128 + ((c ^ ((r >> 1) & 7)) * 16));
    if (row0 + r < row_lim) *reinterpret_cast<V8*>(dst + (int64_t)(row0 + r) * row_stride + c * 8) = v;
  }
  wave_lds_sync();
}

// "Use" a prefetched register here: the wait-count pass then waits for its load at this point (long landed) instead
// of at the first real use in the next iteration -- where the wait, vmcnt being one in-order counter and the number of
// conditional stores in between unknown, would be vmcnt(0) and drain this tile's output stores as well.
template <typename V>
__device__ __forceinline__ void consume(V& v) {
  asm volatile("" : "+v"(v));
}

// ---------------------------------------------------------------- forward
template <typename E>
__global__ __launch_bounds__(1024) void attn_fwd_mfma_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, W = blockDim.x >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const E* qb = (const E*)p.q + b * p.q_sb + h * p.q_sh;
  const E* kb = (const E*)p.k + b * p.k_sb + h * p.k_sh;
  const E* vb = (const E*)p.v + b * p.v_sb + h * p.v_sh;
  E* ob = (E*)p.out + b * p.o_sb + h * p.o_sh;
  float* lse = p.lse + ((int64_t)b * p.H + h) * p.Lq;
  char* ks = smem;
  char* vs = smem + p.Lkp * kRowBytes;
  stage_images<E>(ks, kb, p.k_sl, vs, vb, p.v_sl, p.Lk, p.Lkp);
  __syncthreads();

  const float c2 = p.scale * kLog2e;
  char* patch = p.patch_off >= 0 ? smem + p.patch_off + wid * 2048 : nullptr;
  const int nqt = (p.Lq + 15) >> 4, nkp = p.Lkp >> 5;
  for (int qt = wid; qt < nqt; qt += W) {
    const int qi = qt * 16 + li;
    const int qrow = qi < p.Lq ? qi : p.Lq - 1;
    typename Elem16<E>::v8 qf[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      qf[kk] = *reinterpret_cast<const typename Elem16<E>::v8*>(qb + (int64_t)qrow * p.q_sl + kk * 32 + g * 8);
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;
    for (int kp = 0; kp < nkp; ++kp) {
      f32x4 s[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        s[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          s[tt] = Elem16<E>::mma(
              img_row_frag<E>(ks, (2 * kp + tt) * 16, kk, g, li), qf[kk], s[tt]);
      }
      // softmax bookkeeping on the raw scores (c2 > 0: max and scaling commute); the key mask only exists in the
      // last 32-key step; the accumulator rescale is skipped when no lane of the wave raised its running maximum
      if (kp == nkp - 1) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if ((2 * kp + tt) * 16 + 4 * g + r >= p.Lk) s[tt][r] = -INFINITY;
      }
      float mx = fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3])),
                       fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m, mx * c2);
      const bool raised = mn > m;
      float ps = 0.f;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(fmaf(s[tt][r], c2, -mn));
          s[tt][r] = e;
          ps += e;
        }
      const typename Elem16<E>::v8 pf = pack_pair<E>(s[0], s[1]);
      if (__builtin_amdgcn_ballot_w64(raised) != 0) {       // wave-uniform branch
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        l *= alpha;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
      }
      l += ps;
      m = mn;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        o[dt] = Elem16<E>::mma(img_tr_frag<E>(vs, 2 * kp, 2 * kp + 1, dt * 16, g, li), pf, o[dt]);
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    store_tile<E>(patch, ob, p.o_sl, qt * 16, p.Lq, o, inv, lane);
    if (qi < p.Lq && g == 0) lse[qi] = (m + __builtin_amdgcn_logf(l)) * kLn2;
  }
}

// ---------------------------------------------------------------- forward, scores resident in registers
// Lk <= 32 * NKP <= 256: the 16 x Lkp score tile of a wave (8 * NKP floats per lane) stays in registers, so the row maximum
// is taken once (no running maximum, no accumulator rescale, no per-step cross-lane exchange) and the three phases
// (S^T = K Q^T; max / exp / pack; O^T += V^T P^T) are straight-line code the scheduler can overlap.  The next tile's
// query rows are fetched while this tile computes.  Single-instruction max helpers: fmaxf() on MFMA results would add a
// canonicalising v_max per operand.
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float vmax2(float a, float b) {
  float d;
  asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// all-reduce over the four 16-lane rows of the wave (lanes li, li+16, li+32, li+48) without touching LDS
template <bool kMax>
__device__ __forceinline__ float rows_allreduce(float x) {
  float a = x, b = x;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  x = kMax ? vmax2(a, b) : a + b;
  a = x;
  b = x;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return kMax ? vmax2(a, b) : a + b;
}

// NKP 8 .. 11 (225 .. 352 keys: the 325-token frames of BASELINE configs[4]) hold up to 88 score registers per lane and
// are built for at most 12 waves per workgroup (3 per SIMD, 168 registers); the images of such sequences exceed 80 KiB,
// so one workgroup per CU is resident either way.
template <typename E, int NKP>
__global__ __launch_bounds__(NKP > 7 ? 768 : 1024) void attn_fwd_mfma_res_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using V8 = typename Elem16<E>::v8;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, W = blockDim.x >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const E* qb = (const E*)p.q + b * p.q_sb + h * p.q_sh;
  const E* kb = (const E*)p.k + b * p.k_sb + h * p.k_sh;
  const E* vb = (const E*)p.v + b * p.v_sb + h * p.v_sh;
  E* ob = (E*)p.out + b * p.o_sb + h * p.o_sh;
  float* lse = p.lse + ((int64_t)b * p.H + h) * p.Lq;
  char* ks = smem;
  char* vs = smem + NKP * 32 * kRowBytes;
  const int nqt = (p.Lq + 15) >> 4;
  auto load_q = [&](int qt, V8* qf) {
    const int qi = qt * 16 + li;
    const int qrow = qi < p.Lq ? qi : p.Lq - 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) qf[kk] = *reinterpret_cast<const V8*>(qb + (int64_t)qrow * p.q_sl + kk * 32 + g * 8);
  };
  V8 qf[2];
  load_q(wid, qf);                                   // in flight while K / V are staged
  stage_images<E>(ks, kb, p.k_sl, vs, vb, p.v_sl, p.Lk, NKP * 32);
  __syncthreads();

  const float c2 = p.scale * kLog2e;
  char* patch = p.patch_off >= 0 ? smem + p.patch_off + wid * 2048 : nullptr;
  for (int qt = wid; qt < nqt; qt += W) {
    asm volatile("" ::: "memory");                   // K fragments are loop-invariant: keep their LDS reads in the loop
    f32x4 s[NKP][2];
#pragma unroll
    for (int kp = 0; kp < NKP; ++kp)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        s[kp][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          s[kp][tt] = Elem16<E>::mma(img_row_frag<E>(ks, (2 * kp + tt) * 16, kk, g, li), qf[kk], s[kp][tt]);
      }
    load_q(qt + W, qf);                              // next tile's rows (clamped to Lq - 1: a harmless re-read at the end)
    __builtin_amdgcn_sched_barrier(0);               // issue it here, a whole softmax + P V phase ahead of its use
    // keys >= Lk only exist in the last 32-key step
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if ((2 * (NKP - 1) + tt) * 16 + 4 * g + r >= p.Lk) s[NKP - 1][tt][r] = -INFINITY;
    float mx = vmax2(s[0][0][0], s[0][0][1]);
#pragma unroll
    for (int kp = 0; kp < NKP; ++kp)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        if (kp == 0 && tt == 0) {
          mx = vmax3(mx, s[0][0][2], s[0][0][3]);
        } else {
          mx = vmax3(mx, s[kp][tt][0], s[kp][tt][1]);
          mx = vmax3(mx, s[kp][tt][2], s[kp][tt][3]);
        }
      }
    mx = rows_allreduce<true>(mx);
    const float mn = mx * c2;                        // c2 > 0: max and scaling commute
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The kernel is bound by vector issue beside the MFMAs, so the row sum is taken by the matrix pipe:
    // a fifth "V^T" fragment of ones makes every row of ol the sum over the keys of the 16-bit probabilities that also
    // enter P V (numerator and normaliser see the same rounding; no per-element add, no cross-lane reduction), and the
    // scale / subtract-maximum step runs as packed f32 pairs.
    f32x4 ol = f32x4{0.f, 0.f, 0.f, 0.f};
    V8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (E)1.0f;
    const f32x2 c2v = {c2, c2}, mnv = {-mn, -mn};
#pragma unroll
    for (int kp = 0; kp < NKP; ++kp) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const f32x2 a = __builtin_elementwise_fma(f32x2{s[kp][tt][0], s[kp][tt][1]}, c2v, mnv);
        const f32x2 c = __builtin_elementwise_fma(f32x2{s[kp][tt][2], s[kp][tt][3]}, c2v, mnv);
        s[kp][tt][0] = __builtin_amdgcn_exp2f(a[0]);
        s[kp][tt][1] = __builtin_amdgcn_exp2f(a[1]);
        s[kp][tt][2] = __builtin_amdgcn_exp2f(c[0]);
        s[kp][tt][3] = __builtin_amdgcn_exp2f(c[1]);
      }
      const V8 pf = pack_pair<E>(s[kp][0], s[kp][1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        o[dt] = Elem16<E>::mma(img_tr_frag<E>(vs, 2 * kp, 2 * kp + 1, dt * 16, g, li), pf, o[dt]);
      ol = Elem16<E>::mma(ones, pf, ol);
    }
    const float l = ol[0];                           // rows 4g .. 4g+3 of ol all hold query li's sum
    const float inv = 1.0f / l;
    const int qi = qt * 16 + li;
    consume(qf[0]);
    consume(qf[1]);
    store_tile<E>(patch, ob, p.o_sl, qt * 16, p.Lq, o, inv, lane);
    if (qi < p.Lq && g == 0) lse[qi] = (mn + __builtin_amdgcn_logf(l)) * kLn2;
  }
}


// ---------------------------------------------------------------- backward: dq
// Query on the lane.  dQ^T[d][q] = scale * sum_key K[key][d] * (P (dP - delta))^T[key][q].
// NKP > 0: the key loop has a compile-time trip count and is fully unrolled, so the LDS reads, MFMAs and the
// exp / multiply work of different 32-key steps overlap (a rolled loop serialises read -> MFMA -> VALU -> MFMA per step
// and two or three waves per SIMD cannot hide that chain); NKP == 0 keeps the rolled loop for any length.
// Keys >= Lk only exist in the last step; their K rows are zero, the explicit zero keeps inf * 0 out of the MFMA.
template <typename E, int NKP>
__global__ __launch_bounds__(1024) void attn_bwd_dq_mfma_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using V8 = typename Elem16<E>::v8;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, W = blockDim.x >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const E* qb = (const E*)p.q + b * p.q_sb + h * p.q_sh;
  const E* kb = (const E*)p.k + b * p.k_sb + h * p.k_sh;
  const E* vb = (const E*)p.v + b * p.v_sb + h * p.v_sh;
  const E* ob = (const E*)p.o + b * p.o_sb + h * p.o_sh;
  const E* gb = (const E*)p.d_o + b * p.o_sb + h * p.o_sh;
  E* dqb = (E*)p.dq + b * p.q_sb + h * p.q_sh;
  const float* lse = p.lse + ((int64_t)b * p.H + h) * p.Lq;
  float* delta = p.delta + ((int64_t)b * p.H + h) * p.Lq;
  char* ks = smem;
  char* vs = smem + p.Lkp * kRowBytes;
  const int nqt = (p.Lq + 15) >> 4, nkp = NKP ? NKP : p.Lkp >> 5;
  auto load_rows = [&](int qt, V8* qf, V8* gf, V8* of, float& l2) {
    const int qi = qt * 16 + li;
    const int qrow = qi < p.Lq ? qi : p.Lq - 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      qf[kk] = *reinterpret_cast<const V8*>(qb + (int64_t)qrow * p.q_sl + kk * 32 + g * 8);
      gf[kk] = *reinterpret_cast<const V8*>(gb + (int64_t)qrow * p.o_sl + kk * 32 + g * 8);
      of[kk] = *reinterpret_cast<const V8*>(ob + (int64_t)qrow * p.o_sl + kk * 32 + g * 8);
    }
    l2 = lse[qrow] * kLog2e;                         // lse in log2 units
  };
  V8 qf[2], gf[2], of[2];
  float l2;
  load_rows(wid, qf, gf, of, l2);                    // in flight while K / V are staged
  stage_images<E>(ks, kb, p.k_sl, vs, vb, p.v_sl, p.Lk, p.Lkp);
  __syncthreads();

  const float c2 = p.scale * kLog2e;
  char* patch = p.patch_off >= 0 ? smem + p.patch_off + wid * 2048 : nullptr;
  for (int qt = wid; qt < nqt; qt += W) {
    asm volatile("" ::: "memory");                   // K / V fragments are loop-invariant: keep their LDS reads in the loop
    float dl = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 8; ++j) dl = fmaf((float)gf[kk][j], (float)of[kk][j], dl);
    dl = rows_allreduce<false>(dl);                  // delta of query li
    if (g == 0 && qt * 16 + li < p.Lq) delta[qt * 16 + li] = dl;   // the dk/dv kernel reads it instead of O
    V8 qn[2], gn[2];
    float l2n;
    load_rows(qt + W, qn, gn, of, l2n);              // next tile (rows clamp to Lq - 1: a harmless re-read at the end)
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x2 c2v = {c2, c2}, l2v = {-l2, -l2}, dlv = {-dl, -dl};
    auto step = [&](int kp, bool last) {
      f32x4 s[2], dp[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        s[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          s[tt] = Elem16<E>::mma(img_row_frag<E>(ks, (2 * kp + tt) * 16, kk, g, li), qf[kk], s[tt]);
          dp[tt] = Elem16<E>::mma(img_row_frag<E>(vs, (2 * kp + tt) * 16, kk, g, li), gf[kk], dp[tt]);
        }
      }
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {             // packed f32 pairs: the kernel is bound by vector issue
          const f32x2 x = __builtin_elementwise_fma(f32x2{s[tt][r], s[tt][r + 1]}, c2v, l2v);
          const f32x2 d = f32x2{dp[tt][r], dp[tt][r + 1]} + dlv;
          const f32x2 pr = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
          const f32x2 ds = pr * d;
          s[tt][r] = ds[0];
          s[tt][r + 1] = ds[1];
          if (last && (2 * kp + tt) * 16 + 4 * g + r >= p.Lk) s[tt][r] = 0.f;
          if (last && (2 * kp + tt) * 16 + 4 * g + r + 1 >= p.Lk) s[tt][r + 1] = 0.f;
        }
      const V8 dsf = pack_pair<E>(s[0], s[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        acc[dt] = Elem16<E>::mma(img_tr_frag<E>(ks, 2 * kp, 2 * kp + 1, dt * 16, g, li), dsf, acc[dt]);
    };
    if constexpr (NKP > 0) {
#pragma unroll
      for (int kp = 0; kp < NKP; ++kp) step(kp, kp == NKP - 1);
    } else {
      for (int kp = 0; kp < nkp - 1; ++kp) step(kp, false);
      step(nkp - 1, true);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      consume(qn[kk]);
      consume(gn[kk]);
      consume(of[kk]);
    }
    consume(l2n);
    store_tile<E>(patch, dqb, p.q_sl, qt * 16, p.Lq, acc, p.scale, lane);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      qf[kk] = qn[kk];
      gf[kk] = gn[kk];
    }
    l2 = l2n;
  }
}

// ---------------------------------------------------------------- backward: dk, dv
// Key on the lane.  S[q][key] = Q K^T, dP[q][key] = dO V^T (A = Q / dO row fragments
// from LDS images, B = K / V rows straight from HBM);
// dV^T[d][key] = sum_q dO[q][d] P[q][key];  dK^T[d][key] = scale * sum_q Q[q][d] (P (dP - delta))[q][key].
// Rows >= Lq of the Q / dO images are zero and their lse / delta entries are 0, so they contribute exact zeros without
// a mask.  NQP as NKP above.
template <typename E, int NQP>
__global__ __launch_bounds__(1024) void attn_bwd_dkv_mfma_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using V8 = typename Elem16<E>::v8;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, W = blockDim.x >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const E* qb = (const E*)p.q + b * p.q_sb + h * p.q_sh;
  const E* kb = (const E*)p.k + b * p.k_sb + h * p.k_sh;
  const E* vb = (const E*)p.v + b * p.v_sb + h * p.v_sh;
  const E* gb = (const E*)p.d_o + b * p.o_sb + h * p.o_sh;
  E* dkb = (E*)p.dk + b * p.k_sb + h * p.k_sh;
  E* dvb = (E*)p.dv + b * p.v_sb + h * p.v_sh;
  const float* lse = p.lse + ((int64_t)b * p.H + h) * p.Lq;
  const float* delta = p.delta + ((int64_t)b * p.H + h) * p.Lq;
  char* qs = smem;
  char* gs = smem + p.Lqp * kRowBytes;
  float* l2s = reinterpret_cast<float*>(smem + 2 * p.Lqp * kRowBytes);  // -lse * log2e, [Lqp]
  float* dls = l2s + p.Lqp;                                              // -delta, [Lqp]
  const int nkt = (p.Lk + 15) >> 4, nqp = NQP ? NQP : p.Lqp >> 5;
  auto load_rows = [&](int kt, V8* kf, V8* vf) {
    const int kj = kt * 16 + li;
    const int krow = kj < p.Lk ? kj : p.Lk - 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      kf[kk] = *reinterpret_cast<const V8*>(kb + (int64_t)krow * p.k_sl + kk * 32 + g * 8);
      vf[kk] = *reinterpret_cast<const V8*>(vb + (int64_t)krow * p.v_sl + kk * 32 + g * 8);
    }
  };
  V8 kf[2], vf[2];
  load_rows(wid, kf, vf);                            // in flight while Q / dO are staged
  // per-row statistics: lse in log2 units and delta = rowsum(dO * O) (written by the dq kernel); 0 for padding rows.
  // The first blockDim rows are fetched before the images so that everything shares one memory round trip.
  const int sr = (int)threadIdx.x < p.Lq ? (int)threadIdx.x : p.Lq - 1;
  const float lse0 = lse[sr], dl0 = delta[sr];
  stage_images<E>(qs, qb, p.q_sl, gs, gb, p.o_sl, p.Lq, p.Lqp);
  if ((int)threadIdx.x < p.Lqp) {                  // both negated: the step adds them (packed f32 add / fma, no sign flips)
    l2s[threadIdx.x] = (int)threadIdx.x < p.Lq ? -lse0 * kLog2e : 0.f;
    dls[threadIdx.x] = (int)threadIdx.x < p.Lq ? -dl0 : 0.f;
  }
  for (int r = threadIdx.x + blockDim.x; r < p.Lqp; r += blockDim.x) {
    l2s[r] = r < p.Lq ? -lse[r] * kLog2e : 0.f;
    dls[r] = r < p.Lq ? -delta[r] : 0.f;
  }
  __syncthreads();

  const float c2 = p.scale * kLog2e;
  char* patch = p.patch_off >= 0 ? smem + p.patch_off + wid * 2048 : nullptr;
  for (int kt = wid; kt < nkt; kt += W) {
    asm volatile("" ::: "memory");                   // Q / dO fragments are loop-invariant: keep their LDS reads in the loop
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const f32x2 c2v = {c2, c2};
    auto step = [&](int qp) {
      f32x4 s[2], dp[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        s[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          // D[row = query 4g+r][col = key li]:  A = Q / dO rows (image), B = K / V rows (regs)
          s[tt] = Elem16<E>::mma(img_row_frag<E>(qs, (2 * qp + tt) * 16, kk, g, li), kf[kk], s[tt]);
          dp[tt] = Elem16<E>::mma(img_row_frag<E>(gs, (2 * qp + tt) * 16, kk, g, li), vf[kk], dp[tt]);
        }
      }
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int q0 = (2 * qp + tt) * 16 + 4 * g;
        const f32x4 l2 = *reinterpret_cast<const f32x4*>(l2s + q0);
        const f32x4 dl = *reinterpret_cast<const f32x4*>(dls + q0);
#pragma unroll
        for (int r = 0; r < 4; r += 2) {             // packed f32 pairs (vector-issue-bound)
          const f32x2 x = __builtin_elementwise_fma(f32x2{s[tt][r], s[tt][r + 1]}, c2v, f32x2{l2[r], l2[r + 1]});
          const f32x2 d = f32x2{dp[tt][r], dp[tt][r + 1]} + f32x2{dl[r], dl[r + 1]};
          const f32x2 pr = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
          const f32x2 ds = pr * d;
          s[tt][r] = pr[0];
          s[tt][r + 1] = pr[1];
          dp[tt][r] = ds[0];
          dp[tt][r + 1] = ds[1];
        }
      }
      const V8 pf = pack_pair<E>(s[0], s[1]);
      const V8 dsf = pack_pair<E>(dp[0], dp[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dv[dt] = Elem16<E>::mma(img_tr_frag<E>(gs, 2 * qp, 2 * qp + 1, dt * 16, g, li), pf, dv[dt]);
        dk[dt] = Elem16<E>::mma(img_tr_frag<E>(qs, 2 * qp, 2 * qp + 1, dt * 16, g, li), dsf, dk[dt]);
      }
    };
    if constexpr (NQP > 0) {
#pragma unroll
      for (int qp = 0; qp < NQP; ++qp) step(qp);
    } else {
      for (int qp = 0; qp < nqp; ++qp) step(qp);
    }
    // the next tile's K / V rows are requested BEFORE this tile's stores: vmcnt is one in-order counter, so a load issued
    // behind the stores would wait for them to drain AND for its own latency; in front of them the two overlap
    if (kt + W < nkt) load_rows(kt + W, kf, vf);
    store_tile<E>(patch, dkb, p.k_sl, kt * 16, p.Lk, dk, p.scale, lane);
    store_tile<E>(patch, dvb, p.v_sl, kt * 16, p.Lk, dv, 1.0f, lane);
  }
}

// ---------------------------------------------------------------- backward in ONE pass (Lq, Lk <= 224, equal 32-row padding)
// One workgroup per (batch, head).  Q, dO and K are staged once as LDS images, P comes from the saved lse (no row
// reductions), delta = rowsum(dO * O) is taken while dO is staged: 5 MFMA products per (query, key) block instead of the 7 of
// the dq / dkdv pair, every operand through the CU's memory pipeline once (8 u of traffic per layer instead of 13 u).
//   * compute waves (one per 32 keys): the wave keeps the K and V rows of its two 16-key tiles in registers (B operands) and
//     dK^T / dV^T of those keys in accumulators for the whole kernel; per 32-query step it forms S and dP for its keys
//     (A = Q / dO row fragments of the images), P and dS, and adds dO^T P and Q^T dS to its accumulators -- the dkdv kernel's
//     step on two key tiles at once, so every Q / dO fragment read from LDS feeds two MFMAs;
//   * dQ needs the sum over ALL keys, i.e. over the waves.  Instead of reducing 32 x 64 fp32 partials per wave, the waves
//     write their dS blocks (16-bit, [key][32 queries], the layout their accumulators have) into a strip of 18 KiB; behind
//     the step's barrier ONE wave -- a dedicated dQ wave -- multiplies the whole strip (A = K^T from the K image, B = dS^T by
//     the transposing LDS read) and owns the complete sum over the keys: fixed order, no atomics, bitwise reproducible.  The
//     strip is double-buffered, so the dQ wave works on step s while the compute waves run step s + 1: one barrier per step.
constexpr int kStripRow = 80;        // bytes per key row of a strip: 32 queries x 2 B + 16 B of padding (bank spread)

template <typename E>
__device__ __forceinline__ typename Elem16<E>::v8 strip_tr_frag(const char* strip, int ta, int tb, int col0, int g, int li) {
  // element j of lane (g, li) = strip[16 * (j < 4 ? ta : tb) + 4 g + (j & 3)][col0 + li]   (col0 = 0 or 16)
  const int qq = li >> 2, pp = li & 3;
  typename Elem16<E>::v4 half[2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const int r = 16 * (hf ? tb : ta) + 4 * g + qq;
    half[hf] = Elem16<E>::tr_read(strip + r * kStripRow + col0 * 2 + 8 * pp);
  }
  return __builtin_shufflevector(half[0], half[1], 0, 1, 2, 3, 4, 5, 6, 7);
}

// Prologue of the one-pass backward: Q, dO and K into their images (rows >= L zero) and, from the dO / O chunks of the same
// loads, -delta[r] = -sum_e dO[r][e] * O[r][e] (the 8 chunks of a row sit on 8 consecutive lanes: blockDim is a multiple of
// 8).  EVERY load of the workgroup -- NCH chunks per thread and matrix -- is issued before the first LDS write: one memory
// round trip under full-chip load costs ~5 k clocks, and three staging loops in sequence paid it three times.
template <typename E, int NCH>
__device__ __forceinline__ void stage_bwd_operands(char* qs, const E* q, int64_t q_sl, char* gs, const E* g, const E* o, int64_t o_sl,
                                                   char* ks, const E* k, int64_t k_sl, int Lq, int Lk, int rows_pad, float* dls) {
  using V8 = typename Elem16<E>::v8;
  const int total = rows_pad * 8, nt = blockDim.x;
  V8 vq[NCH], vg[NCH], vo[NCH], vk[NCH];
#pragma unroll
  for (int u = 0; u < NCH; ++u) {
    const int idx = min((int)threadIdx.x + u * nt, total - 1);      // past the end: the last chunk again (same data, same slot)
    const int r = idx >> 3, c = idx & 7;
    const int rq = r < Lq ? r : 0, rk = r < Lk ? r : 0;
    vq[u] = *reinterpret_cast<const V8*>(q + (int64_t)rq * q_sl + c * 8);
    vg[u] = *reinterpret_cast<const V8*>(g + (int64_t)rq * o_sl + c * 8);
    vo[u] = *reinterpret_cast<const V8*>(o + (int64_t)rq * o_sl + c * 8);
    vk[u] = *reinterpret_cast<const V8*>(k + (int64_t)rk * k_sl + c * 8);
  }
  __builtin_amdgcn_sched_barrier(0);
  const V8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < NCH; ++u) {
    const int idx = min((int)threadIdx.x + u * nt, total - 1);
    const int r = idx >> 3, off = img_off(r, idx & 7);
    const bool lq = r < Lq, lk = r < Lk;
    *reinterpret_cast<V8*>(qs + off) = lq ? vq[u] : zero;
    *reinterpret_cast<V8*>(gs + off) = lq ? vg[u] : zero;
    *reinterpret_cast<V8*>(ks + off) = lk ? vk[u] : zero;
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) d = fmaf((float)vg[u][j], (float)vo[u][j], d);
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    d += __shfl_xor(d, 4, 64);
    if ((idx & 7) == 0) dls[r] = lq ? -d : 0.f;
  }
}

template <typename E, int NP>
__global__ __launch_bounds__(512) void attn_bwd_fused_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using V8 = typename Elem16<E>::v8;
  using V4 = typename Elem16<E>::v4;
  constexpr int LP = 32 * NP;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int W = blockDim.x >> 6, WC = W - 1;              // WC compute waves (32 keys each) + the dQ wave
  const int g = lane >> 4, li = lane & 15;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const E* qb = (const E*)p.q + b * p.q_sb + h * p.q_sh;
  const E* kb = (const E*)p.k + b * p.k_sb + h * p.k_sh;
  const E* vb = (const E*)p.v + b * p.v_sb + h * p.v_sh;
  const E* ob = (const E*)p.o + b * p.o_sb + h * p.o_sh;
  const E* gb = (const E*)p.d_o + b * p.o_sb + h * p.o_sh;
  E* dqb = (E*)p.dq + b * p.q_sb + h * p.q_sh;
  E* dkb = (E*)p.dk + b * p.k_sb + h * p.k_sh;
  E* dvb = (E*)p.dv + b * p.v_sb + h * p.v_sh;
  const float* lse = p.lse + ((int64_t)b * p.H + h) * p.Lq;
  char* qs = smem;
  char* gs = qs + LP * kRowBytes;
  char* ks = gs + LP * kRowBytes;
  char* strip0 = ks + LP * kRowBytes;
  float* l2s = reinterpret_cast<float*>(strip0 + 2 * LP * kStripRow);   // -lse * log2e, [LP]
  float* dls = l2s + LP;                                                // -delta, [LP]
  char* patch = reinterpret_cast<char*>(dls + LP) + wid * 2048;

  // phase: start
  // ---- prologue: the compute waves' K / V rows (in flight while the images are staged), images, per-row statistics
  V8 kf[2][2], vf[2][2];
  if (wid < WC) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kj = (2 * wid + j) * 16 + li;
      const int krow = kj < p.Lk ? kj : p.Lk - 1;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        kf[j][kk] = *reinterpret_cast<const V8*>(kb + (int64_t)krow * p.k_sl + kk * 32 + g * 8);
        vf[j][kk] = *reinterpret_cast<const V8*>(vb + (int64_t)krow * p.v_sl + kk * 32 + g * 8);
      }
    }
  }
  {
    const int r = threadIdx.x;                       // (blockDim >= LP: one row statistic per thread, in the same round trip)
    const float l = lse[r < p.Lq ? r : 0];
    // NCH chunks per thread cover an image: LP * 8 = 256 NP <= NCH * blockDim = 256 (NP + 1)
    stage_bwd_operands<E, 4>(qs, qb, p.q_sl, gs, gb, ob, p.o_sl, ks, kb, p.k_sl, p.Lq, p.Lk, LP, dls);
    if (r < LP) l2s[r] = r < p.Lq ? -l * kLog2e : 0.f;
  }
  __syncthreads();
  // phase: staged
  const float c2 = p.scale * kLog2e;
  if (wid < WC) {
    // ------------------------------------------------------------ compute wave: keys [32 wid, 32 wid + 32)
    f32x4 dk[2][4], dv[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dk[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        dv[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    const f32x2 c2v = {c2, c2};
    const bool kvalid[2] = {(2 * wid) * 16 + li < p.Lk, (2 * wid + 1) * 16 + li < p.Lk};
    char* myrow[2] = {strip0 + ((2 * wid) * 16 + li) * kStripRow + g * 8, strip0 + ((2 * wid + 1) * 16 + li) * kStripRow + g * 8};
#pragma unroll
    for (int qp = 0; qp < NP; ++qp) {
      asm volatile("" ::: "memory");                 // image fragments are loop-invariant in form: keep their reads in the step
      V8 qfr[2][2], gfr[2][2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          qfr[tt][kk] = img_row_frag<E>(qs, (2 * qp + tt) * 16, kk, g, li);
          gfr[tt][kk] = img_row_frag<E>(gs, (2 * qp + tt) * 16, kk, g, li);
        }
      f32x4 l2[2], dl[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        l2[tt] = *reinterpret_cast<const f32x4*>(l2s + (2 * qp + tt) * 16 + 4 * g);
        dl[tt] = *reinterpret_cast<const f32x4*>(dls + (2 * qp + tt) * 16 + 4 * g);
      }
      V8 pf[2], dsf[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 s[2], dp[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          s[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
          dp[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            // D[row = query 4g + r][col = key li]
            s[tt] = Elem16<E>::mma(qfr[tt][kk], kf[j][kk], s[tt]);
            dp[tt] = Elem16<E>::mma(gfr[tt][kk], vf[j][kk], dp[tt]);
          }
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const f32x2 x = __builtin_elementwise_fma(f32x2{s[tt][r], s[tt][r + 1]}, c2v, f32x2{l2[tt][r], l2[tt][r + 1]});
            const f32x2 d = f32x2{dp[tt][r], dp[tt][r + 1]} + f32x2{dl[tt][r], dl[tt][r + 1]};
            f32x2 pr = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
            if (!kvalid[j]) pr = f32x2{0.f, 0.f};    // keys >= Lk: their K / V rows are clamped copies
            const f32x2 ds = pr * d;
            s[tt][r] = pr[0];
            s[tt][r + 1] = pr[1];
            dp[tt][r] = ds[0];
            dp[tt][r + 1] = ds[1];
          }
        pf[j] = pack_pair<E>(s[0], s[1]);
        dsf[j] = pack_pair<E>(dp[0], dp[1]);
      }
      // dS^T of this step for the dQ wave: row = key, 32 queries of the step (this lane: queries 4g .. 4g+3 of either tile)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        char* row = myrow[j] + (qp & 1) * (LP * kStripRow);
        *reinterpret_cast<V4*>(row) = __builtin_shufflevector(dsf[j], dsf[j], 0, 1, 2, 3);
        *reinterpret_cast<V4*>(row + 32) = __builtin_shufflevector(dsf[j], dsf[j], 4, 5, 6, 7);
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const V8 gt = img_tr_frag<E>(gs, 2 * qp, 2 * qp + 1, dt * 16, g, li);
        const V8 qt = img_tr_frag<E>(qs, 2 * qp, 2 * qp + 1, dt * 16, g, li);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          dv[j][dt] = Elem16<E>::mma(gt, pf[j], dv[j][dt]);
          dk[j][dt] = Elem16<E>::mma(qt, dsf[j], dk[j][dt]);
        }
      }
      // phase: work
      __syncthreads();                                // strip (qp & 1) complete; the dQ wave takes it from here
      // phase: step
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k0 = (2 * wid + j) * 16;
      if (k0 < p.Lk) {                                // wave-uniform
        store_tile<E>(patch, dkb, p.k_sl, k0, p.Lk, dk[j], p.scale, lane);
        store_tile<E>(patch, dvb, p.v_sl, k0, p.Lk, dv[j], 1.0f, lane);
      }
    }
    // phase: stored
  } else {
    // ------------------------------------------------------------ dQ wave: dQ^T[d][q] = scale * sum_key K^T[d][key] dS^T[key][q]
    // The K^T fragments (A operands) do not change from step to step: all 4 NP of them are read once and stay in registers
    // (112 at NP = 7; this wave holds no dK / dV accumulators), so a step costs 2 NP strip reads and 8 NP MFMAs.
    V8 ka[NP][4];
#pragma unroll
    for (int kp = 0; kp < NP; ++kp)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) ka[kp][dt] = img_tr_frag<E>(ks, 2 * kp, 2 * kp + 1, dt * 16, g, li);
#pragma unroll 1
    for (int qp = 0; qp < NP; ++qp) {
      __syncthreads();
      const char* strip = strip0 + (qp & 1) * (LP * kStripRow);
      V8 bq[NP][2];
#pragma unroll
      for (int kp = 0; kp < NP; ++kp) {
        bq[kp][0] = strip_tr_frag<E>(strip, 2 * kp, 2 * kp + 1, 0, g, li);
        bq[kp][1] = strip_tr_frag<E>(strip, 2 * kp, 2 * kp + 1, 16, g, li);
      }
      // phase: dq read
      f32x4 acc[2][4];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[tt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kp = 0; kp < NP; ++kp)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          acc[0][dt] = Elem16<E>::mma(ka[kp][dt], bq[kp][0], acc[0][dt]);
          acc[1][dt] = Elem16<E>::mma(ka[kp][dt], bq[kp][1], acc[1][dt]);
        }
      // phase: dq mfma
      // this wave is the critical path of a step: its tiles go out as per-lane 8-byte stores straight from the accumulators
      // (the whole-row form through the LDS patch costs the wave two more LDS round trips per tile)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
        if ((2 * qp + tt) * 16 < p.Lq) store_tile<E>(nullptr, dqb, p.q_sl, (2 * qp + tt) * 16, p.Lq, acc[tt], p.scale, lane);
      // phase: dq unit
    }
  }
}

// ======================================================================= host
constexpr int kMaxLds = 160 * 1024;

int fill_params(const dvt_attn_desc* d, AttnParams& p, bool bwd, const char* name) {
  DVT_REQUIRE(d, "%s: null descriptor", name);
  DVT_REQUIRE(d->q && d->k && d->v && d->o && d->lse, "%s: null q/k/v/o/lse", name);
  DVT_REQUIRE(d->B >= 0 && d->H > 0 && d->Lq > 0 && d->Lk > 0 && d->dh > 0, "%s: bad sizes", name);
  DVT_REQUIRE(d->B * d->H < (1ll << 31) && d->Lq < (1 << 24) && d->Lk < (1 << 24), "%s: sizes too large", name);
  if (bwd) DVT_REQUIRE(d->d_o && d->dq && d->dk && d->dv, "%s: null gradient pointer", name);
  p.q = d->q; p.k = d->k; p.v = d->v; p.o = d->o; p.out = d->o; p.d_o = d->d_o;
  p.dq = d->dq; p.dk = d->dk; p.dv = d->dv; p.lse = d->lse; p.delta = nullptr;
  p.B = (int)d->B; p.H = (int)d->H; p.Lq = (int)d->Lq; p.Lk = (int)d->Lk; p.dh = (int)d->dh;
  p.q_sb = d->q_sb; p.q_sh = d->q_sh; p.q_sl = d->q_sl;
  p.k_sb = d->k_sb; p.k_sh = d->k_sh; p.k_sl = d->k_sl;
  p.v_sb = d->v_sb; p.v_sh = d->v_sh; p.v_sl = d->v_sl;
  p.o_sb = d->o_sb; p.o_sh = d->o_sh; p.o_sl = d->o_sl;
  p.scale = d->scale;
  DVT_REQUIRE(d->dropout_p >= 0.f && d->dropout_p < 1.f, "%s: dropout_p must be in [0, 1)", name);
  DVT_REQUIRE(d->dropout_p == 0.f || d->rng_state, "%s: dropout needs rng_state", name);
  p.drop_thr = 0u; p.drop_scale = 1.f; p.rng = d->rng_state; p.rng_off = d->rng_offset;
  if (d->dropout_p > 0.f) {
    const double th = (double)d->dropout_p * 4294967296.0;
    p.drop_thr = th >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)(th + 0.5);
    if (p.drop_thr == 0u) p.drop_thr = 1u;
    p.drop_scale = 1.0f / (1.0f - d->dropout_p);
  }
  p.Lkp = (p.Lk + 31) & ~31;
  p.Lqp = (p.Lq + 31) & ~31;
  p.patch_off = -1;

  return DVT_OK;
}

bool strides_vec_ok(const dvt_attn_desc* d) {
  const int64_t s[] = {d->q_sb, d->q_sh, d->q_sl, d->k_sb, d->k_sh, d->k_sl,
                       d->v_sb, d->v_sh, d->v_sl, d->o_sb, d->o_sh, d->o_sl};
  for (int64_t x : s)
    if (x % 8) return false;
  return true;
}

bool mfma_fwd_ok(const dvt_attn_desc* d, const AttnParams& p) {
  return d->dropout_p == 0.f && d->scale > 0.f && dvt_is_16bit(d->dtype) && d->dh == DH && strides_vec_ok(d) && dvt_aligned16(d->q) &&
         dvt_aligned16(d->k) && dvt_aligned16(d->v) && dvt_aligned16(d->o) &&
         2 * p.Lkp * kRowBytes <= kMaxLds;
}

// one query per (b, h): the streaming matrix-vector kernels (LDS: Lk floats per wave)
bool q1_ok(const dvt_attn_desc* d, const AttnParams& p) {
  return p.Lq == 1 && d->dropout_p == 0.f && dvt_is_16bit(d->dtype) && d->dh == DH && strides_vec_ok(d) &&
         dvt_aligned16(d->q) && dvt_aligned16(d->k) && dvt_aligned16(d->v) && dvt_aligned16(d->o) &&
         (size_t)4 * p.Lkp * sizeof(float) <= (size_t)kMaxLds;
}

bool mfma_bwd_ok(const dvt_attn_desc* d, const AttnParams& p) {
  return mfma_fwd_ok(d, p) && dvt_aligned16(d->d_o) && dvt_aligned16(d->dq) && dvt_aligned16(d->dk) &&
         dvt_aligned16(d->dv) && 2 * p.Lqp * kRowBytes + 2 * p.Lqp * 4 <= kMaxLds;
}

// the one-pass backward: both lengths padded to the same multiple of 32, at most 224 (three 28 KiB images + two strips)
bool fused_bwd_ok(const AttnParams& p) { return p.Lqp == p.Lkp && p.Lkp <= 224; }
size_t fused_bwd_lds(int NP, int waves) {
  const size_t LP = 32 * (size_t)NP;
  return 3 * LP * kRowBytes + 2 * LP * kStripRow + 2 * LP * sizeof(float) + (size_t)waves * 2048;
}

// waves per block so that the tile count splits evenly over as few rounds as possible.  When the staged images
// leave room for only one workgroup per CU (long sequences: > 80 KiB of LDS), that workgroup may use all 16 wave slots
// (the kernels need <= 115 VGPRs, i.e. 4 waves per SIMD fit).
int pick_waves(int tiles, size_t lds_bytes, int cap = 16) {
  int maxw = lds_bytes > 80 * 1024 ? 16 : 8;
  if (maxw > cap) maxw = cap;
  const int rounds = (tiles + maxw - 1) / maxw;
  int w = (tiles + rounds - 1) / rounds;
  return w < 1 ? 1 : w;
}

// LDS bytes including the per-wave 2 KiB output patches (store_tile) when they keep the occupancy the images allow
// (two workgroups per CU up to 80 KiB); sets p.patch_off
size_t with_patches(size_t lds, int waves, AttnParams& p) {
  const size_t tot = lds + (size_t)waves * 2048;
  const bool ok = lds <= 80 * 1024 ? 2 * tot <= (size_t)kMaxLds : tot <= (size_t)kMaxLds;
  p.patch_off = ok ? (int)lds : -1;
  return ok ? tot : lds;
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) {                  // (unconditional: the attribute is per device, the call is cheap)
    const hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return dvt_fail_hip(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
  }
  return 0;
}

}  // namespace

extern "C" {

size_t dvt_attention_bwd_workspace_bytes(const dvt_attn_desc* d) {
  if (!d) return 0;
  return (size_t)d->B * (size_t)d->H * (size_t)d->Lq * sizeof(float);
}

int dvt_attention_fwd(const dvt_attn_desc* d, dvt_stream_t stream) {
  AttnParams p;
  int rc = fill_params(d, p, false, "dvt_attention_fwd");
  if (rc) return rc;
  if (p.B == 0) return DVT_OK;
  hipStream_t st = (hipStream_t)stream;
  if (q1_ok(d, p)) {                             // one query per (b, h): the CLS-row form of a stack's last layer
    const dim3 grid((unsigned)dvt_cdiv((int64_t)p.B * p.H, 4)), block(256);
    const size_t lds = (size_t)4 * p.Lkp * sizeof(float);
    DVT_DISPATCH_16BIT(d->dtype, E, {
      if (int rc_ = set_lds(attn_fwd_q1_kernel<E>, lds)) return rc_;
      hipLaunchKernelGGL((attn_fwd_q1_kernel<E>), grid, block, lds, st, p);
    });
    DVT_LAUNCH_CHECK("dvt_attention_fwd(q1)");
    return DVT_OK;
  }
  if (mfma_fwd_ok(d, p)) {
    const size_t lds_img = (size_t)2 * p.Lkp * kRowBytes;
    const int nkp = p.Lkp >> 5;
    const int W = pick_waves((p.Lq + 15) / 16, lds_img, nkp > 7 && nkp <= 11 ? 12 : 16);
    const size_t lds = with_patches(lds_img, W, p);
    const dim3 grid((unsigned)(p.B * p.H)), block(64 * W);
#define DVT_ATTN_FWD_RES(NKP)                                                               \
  case NKP:                                                                                 \
    if (int rc_ = set_lds(attn_fwd_mfma_res_kernel<E, NKP>, lds)) return rc_;                                         \
    hipLaunchKernelGGL((attn_fwd_mfma_res_kernel<E, NKP>), grid, block, lds, st, p);        \
    break
    DVT_DISPATCH_16BIT(d->dtype, E, {
      switch (nkp) {                             // scores of up to 352 keys stay in registers
        DVT_ATTN_FWD_RES(1); DVT_ATTN_FWD_RES(2); DVT_ATTN_FWD_RES(3); DVT_ATTN_FWD_RES(4);
        DVT_ATTN_FWD_RES(5); DVT_ATTN_FWD_RES(6); DVT_ATTN_FWD_RES(7); DVT_ATTN_FWD_RES(8);
        DVT_ATTN_FWD_RES(9); DVT_ATTN_FWD_RES(10); DVT_ATTN_FWD_RES(11);
        default:
          if (int rc_ = set_lds(attn_fwd_mfma_kernel<E>, lds)) return rc_;
          hipLaunchKernelGGL((attn_fwd_mfma_kernel<E>), grid, block, lds, st, p);
      }
    });
#undef DVT_ATTN_FWD_RES
    DVT_LAUNCH_CHECK("dvt_attention_fwd(mfma)");
    return DVT_OK;
  }
  const int nt_s = p.Lq * p.dh >= 4096 ? 512 : 256;
  const size_t lds_s = ((size_t)(p.Lq + 2 * p.Lk) * small_dhp(p.dh) +
                        (size_t)small_parts(p.Lq, p.Lk, nt_s) * p.Lq * small_lp(p.Lk)) * sizeof(float);
  if (p.Lq <= kSmallL && p.Lk <= kSmallL && p.dh <= kSmallDh && lds_s <= (size_t)kMaxLds) {   // short sequences: a workgroup per (b, h)
    const dim3 grid((unsigned)(p.B * p.H)), block(nt_s);
#define DVT_ATTN_SMALL_FWD(T)                                                   \
  do {                                                                          \
    if (int rc_ = set_lds(attn_small_fwd_kernel<T>, lds_s)) return rc_;                                   \
    hipLaunchKernelGGL((attn_small_fwd_kernel<T>), grid, block, lds_s, st, p);  \
  } while (0)
    if (d->dtype == DVT_F32) DVT_ATTN_SMALL_FWD(float);
    else if (d->dtype == DVT_BF16) DVT_ATTN_SMALL_FWD(bf16);
    else if (d->dtype == DVT_F16) DVT_ATTN_SMALL_FWD(f16);
    else DVT_UNSUPPORTED("dvt_attention_fwd: dtype %d not supported", d->dtype);
#undef DVT_ATTN_SMALL_FWD
    DVT_LAUNCH_CHECK("dvt_attention_fwd(short)");
    return DVT_OK;
  }
  const size_t lds = (size_t)4 * (p.dh + p.Lk) * sizeof(float);
  if (lds > (size_t)kMaxLds) DVT_UNSUPPORTED("dvt_attention_fwd: Lk = %d, dh = %d exceed the LDS budget", p.Lk, p.dh);
  const int64_t rows = (int64_t)p.B * p.H * p.Lq;
  const dim3 grid((unsigned)dvt_cdiv(rows, 4)), block(256);
  if (d->dtype == DVT_F32) {
    if (int rc_ = set_lds(attn_fwd_generic_kernel<float>, lds)) return rc_;
    hipLaunchKernelGGL((attn_fwd_generic_kernel<float>), grid, block, lds, st, p);
  } else if (d->dtype == DVT_BF16) {
    if (int rc_ = set_lds(attn_fwd_generic_kernel<bf16>, lds)) return rc_;
    hipLaunchKernelGGL((attn_fwd_generic_kernel<bf16>), grid, block, lds, st, p);
  } else if (d->dtype == DVT_F16) {
    if (int rc_ = set_lds(attn_fwd_generic_kernel<f16>, lds)) return rc_;
    hipLaunchKernelGGL((attn_fwd_generic_kernel<f16>), grid, block, lds, st, p);
  } else {
    DVT_UNSUPPORTED("dvt_attention_fwd: dtype %d not supported", d->dtype);
  }
  DVT_LAUNCH_CHECK("dvt_attention_fwd(generic)");
  return DVT_OK;
}

int dvt_attention_bwd(const dvt_attn_desc* d, dvt_stream_t stream) {
  AttnParams p;
  int rc = fill_params(d, p, true, "dvt_attention_bwd");
  if (rc) return rc;
  if (p.B == 0) return DVT_OK;
  hipStream_t st = (hipStream_t)stream;
  if (q1_ok(d, p) && dvt_aligned16(d->d_o) && dvt_aligned16(d->dq) && dvt_aligned16(d->dk) && dvt_aligned16(d->dv)) {
    const dim3 grid((unsigned)dvt_cdiv((int64_t)p.B * p.H, 4)), block(256);
    DVT_DISPATCH_16BIT(d->dtype, E, hipLaunchKernelGGL((attn_bwd_q1_kernel<E>), grid, block, 0, st, p));
    DVT_LAUNCH_CHECK("dvt_attention_bwd(q1)");
    return DVT_OK;
  }
  if (mfma_bwd_ok(d, p) && fused_bwd_ok(p) && !d->bwd_two_pass) {      // one pass: Q, dO, K staged once, dK / dV in registers, dQ through LDS strips
    const int NP = p.Lkp >> 5;
    const int waves = NP + 1;                      // one compute wave per 32 keys + the dQ wave
    const size_t lds = fused_bwd_lds(NP, waves);
    const dim3 grid((unsigned)(p.B * p.H)), block(64 * waves);
#define DVT_ATTN_BWD_FUSED(N)                                                               \
  case N:                                                                                   \
    if (int rc_ = set_lds(attn_bwd_fused_kernel<E, N>, lds)) return rc_;                                              \
    hipLaunchKernelGGL((attn_bwd_fused_kernel<E, N>), grid, block, lds, st, p);             \
    break
    DVT_DISPATCH_16BIT(d->dtype, E, {
      switch (NP) {
        DVT_ATTN_BWD_FUSED(1); DVT_ATTN_BWD_FUSED(2); DVT_ATTN_BWD_FUSED(3); DVT_ATTN_BWD_FUSED(4);
        DVT_ATTN_BWD_FUSED(5); DVT_ATTN_BWD_FUSED(6); DVT_ATTN_BWD_FUSED(7);
      }
    });
#undef DVT_ATTN_BWD_FUSED
    DVT_LAUNCH_CHECK("dvt_attention_bwd(fused)");
    return DVT_OK;
  }
  if (mfma_bwd_ok(d, p)) {
    DVT_REQUIRE(d->workspace, "dvt_attention_bwd: workspace (dvt_attention_bwd_workspace_bytes) required");
    p.delta = (float*)d->workspace;             // dq kernel -> dk/dv kernel
    const size_t img_q = (size_t)2 * p.Lkp * kRowBytes;
    const size_t img_kv = (size_t)2 * p.Lqp * kRowBytes + (size_t)2 * p.Lqp * sizeof(float);
    const dim3 grid((unsigned)(p.B * p.H));
    const int wq = pick_waves((p.Lq + 15) / 16, img_q), wkv = pick_waves((p.Lk + 15) / 16, img_kv);
    const dim3 block_q(64 * wq), block_kv(64 * wkv);
    AttnParams pq = p, pkv = p;
    const size_t lds_q = with_patches(img_q, wq, pq), lds_kv = with_patches(img_kv, wkv, pkv);
#define DVT_ATTN_BWD_DQ(NKP)                                                                \
  case NKP:                                                                                 \
    if (int rc_ = set_lds(attn_bwd_dq_mfma_kernel<E, NKP>, lds_q)) return rc_;                                        \
    hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<E, NKP>), grid, block_q, lds_q, st, pq);     \
    break
#define DVT_ATTN_BWD_DKV(NQP)                                                               \
  case NQP:                                                                                 \
    if (int rc_ = set_lds(attn_bwd_dkv_mfma_kernel<E, NQP>, lds_kv)) return rc_;                                      \
    hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<E, NQP>), grid, block_kv, lds_kv, st, pkv);  \
    break
    DVT_DISPATCH_16BIT(d->dtype, E, {
      switch (p.Lkp >> 5) {                      // unrolled key loop up to 352 keys
        DVT_ATTN_BWD_DQ(1); DVT_ATTN_BWD_DQ(2); DVT_ATTN_BWD_DQ(3); DVT_ATTN_BWD_DQ(4);
        DVT_ATTN_BWD_DQ(5); DVT_ATTN_BWD_DQ(6); DVT_ATTN_BWD_DQ(7); DVT_ATTN_BWD_DQ(8);
        DVT_ATTN_BWD_DQ(9); DVT_ATTN_BWD_DQ(10); DVT_ATTN_BWD_DQ(11);
        default:
          if (int rc_ = set_lds(attn_bwd_dq_mfma_kernel<E, 0>, lds_q)) return rc_;
          hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<E, 0>), grid, block_q, lds_q, st, pq);
      }
      switch (p.Lqp >> 5) {                      // unrolled query loop up to 352 queries
        DVT_ATTN_BWD_DKV(1); DVT_ATTN_BWD_DKV(2); DVT_ATTN_BWD_DKV(3); DVT_ATTN_BWD_DKV(4);
        DVT_ATTN_BWD_DKV(5); DVT_ATTN_BWD_DKV(6); DVT_ATTN_BWD_DKV(7); DVT_ATTN_BWD_DKV(8);
        DVT_ATTN_BWD_DKV(9); DVT_ATTN_BWD_DKV(10); DVT_ATTN_BWD_DKV(11);
        default:
          if (int rc_ = set_lds(attn_bwd_dkv_mfma_kernel<E, 0>, lds_kv)) return rc_;
          hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<E, 0>), grid, block_kv, lds_kv, st, pkv);
      }
    });
#undef DVT_ATTN_BWD_DQ
#undef DVT_ATTN_BWD_DKV
    DVT_LAUNCH_CHECK("dvt_attention_bwd(dkdv)");
    return DVT_OK;
  }
  const int nt_s = p.Lq * p.dh >= 4096 ? 512 : 256;
  const size_t lds_s = ((size_t)2 * (p.Lq + p.Lk) * small_dhp(p.dh) +
                        (size_t)2 * small_parts(p.Lq, p.Lk, nt_s) * p.Lq * small_lp(p.Lk) +
                        (size_t)2 * p.Lk * small_lp(p.Lq)) * sizeof(float);
  if (p.Lq <= kSmallL && p.Lk <= kSmallL && p.dh <= kSmallDh && lds_s <= (size_t)kMaxLds) {   // short sequences: dQ, dK, dV of a (b, h) in one launch
    const dim3 grid((unsigned)(p.B * p.H)), block(nt_s);
#define DVT_ATTN_SMALL_BWD(T)                                                   \
  do {                                                                          \
    if (int rc_ = set_lds(attn_small_bwd_kernel<T>, lds_s)) return rc_;                                   \
    hipLaunchKernelGGL((attn_small_bwd_kernel<T>), grid, block, lds_s, st, p);  \
  } while (0)
    if (d->dtype == DVT_F32) DVT_ATTN_SMALL_BWD(float);
    else if (d->dtype == DVT_BF16) DVT_ATTN_SMALL_BWD(bf16);
    else if (d->dtype == DVT_F16) DVT_ATTN_SMALL_BWD(f16);
    else DVT_UNSUPPORTED("dvt_attention_bwd: dtype %d not supported", d->dtype);
#undef DVT_ATTN_SMALL_BWD
    DVT_LAUNCH_CHECK("dvt_attention_bwd(short)");
    return DVT_OK;
  }
  DVT_REQUIRE(d->workspace, "dvt_attention_bwd: workspace (dvt_attention_bwd_workspace_bytes) required");
  p.delta = (float*)d->workspace;
  const size_t lds_q = (size_t)4 * (2 * p.dh + p.Lk) * sizeof(float);
  const size_t lds_kv = (size_t)4 * (2 * p.dh + 2 * p.Lq) * sizeof(float);
  if (lds_q > (size_t)kMaxLds || lds_kv > (size_t)kMaxLds)
    DVT_UNSUPPORTED("dvt_attention_bwd: Lq = %d, Lk = %d, dh = %d exceed the LDS budget", p.Lq, p.Lk, p.dh);
  const int64_t qrows = (int64_t)p.B * p.H * p.Lq, krows = (int64_t)p.B * p.H * p.Lk;
  const dim3 block(256);
#define DVT_ATTN_BWD_GENERIC(T)                                                                   \
  do {                                                                                            \
    if (int rc_ = set_lds(attn_bwd_dq_generic_kernel<T>, lds_q)) return rc_;                                                \
    if (int rc_ = set_lds(attn_bwd_dkv_generic_kernel<T>, lds_kv)) return rc_;                                              \
    hipLaunchKernelGGL((attn_delta_kernel<T>), dim3((unsigned)dvt_cdiv(qrows, 4)), block, 0, st, p); \
    hipLaunchKernelGGL((attn_bwd_dq_generic_kernel<T>), dim3((unsigned)dvt_cdiv(qrows, 4)), block, lds_q, st, p); \
    hipLaunchKernelGGL((attn_bwd_dkv_generic_kernel<T>), dim3((unsigned)dvt_cdiv(krows, 4)), block, lds_kv, st, p); \
  } while (0)
  if (d->dtype == DVT_F32) DVT_ATTN_BWD_GENERIC(float);
  else if (d->dtype == DVT_BF16) DVT_ATTN_BWD_GENERIC(bf16);
  else if (d->dtype == DVT_F16) DVT_ATTN_BWD_GENERIC(f16);
  else DVT_UNSUPPORTED("dvt_attention_bwd: dtype %d not supported", d->dtype);
#undef DVT_ATTN_BWD_GENERIC
  DVT_LAUNCH_CHECK("dvt_attention_bwd(generic)");
  return DVT_OK;
}

}  // extern "C"
