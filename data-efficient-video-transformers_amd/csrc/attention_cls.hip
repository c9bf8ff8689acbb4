// attention_cls.hip -- the attention of a stack's LAST layer, evaluated for row 0 of every sequence, with the key and
// value projections folded into the one query (src/models/vit.py:46-58 behind :119-120 / :126).
//
// With a single query per (sequence, head) the two projections commute with the attention sums:
//
//   s_jh = scale q_h . (Wk_h LN(x_j)) = r_h . LN(x_j)        r_h = scale Wk_h^T q_h     (a d-vector per head)
//   o_h  = sum_j p_jh Wv_h LN(x_j)    = Wv_h m_h             m_h = sum_j p_jh LN(x_j)   (a d-vector per head)
//
// so neither K nor V nor LN(x) of the N - 1 rows that are never read again exist anywhere: the forward is ONE pass over
// the raw rows x_j (LayerNorm statistics, H dot products and H weighted sums per row), the backward one more (which also
// performs the LayerNorm backward of those rows), instead of LayerNorm + a [rows, d] x [d, 2 inner] GEMM + attention
// forward, and attention backward + two such GEMMs + LayerNorm backward (158.7 GF per step at the metric shape).
// Same values as the unfolded block up to summation order; everything between the 16-bit rows and the fp32 results is
// fp32.
//
// Kernels: one 8-wave workgroup per sequence; a wave owns rows w, w + 8, ... and keeps them in registers (16 bytes per
// lane and row, all requested before the first is used) across its passes; a lane owns 8 consecutive columns (d <= 512).
// The per-head d-vectors on either side (r, m and their gradients, [S, H, d] fp32) are produced / consumed by the three
// head-wise products at the bottom: they touch S rows only.
#include "common.h"
#include "ln_reduce.h"

namespace {

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
constexpr int kMaxH = 8;
constexpr int kGroup = 5;                                 // rows a wave processes without a branch in between

struct ClsParams {
  const void* x;
  int64_t xs0, xs1;
  const float* gamma;
  const float* beta;
  float eps;
  int S, N, d, H;
  const float* R;
  float* A;
  float* lse;
  float* P;
  float* mean;
  float* rstd;
  const float* dM;
  void* dx;
  float* G;
  float* partial;
};

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// Totals of 8 per-lane values over the wave: every step halves the values a lane still carries (the swaps exchange the
// upper values of one half of the wave with the lower values of the other), 6 swaps + 4 DPP steps instead of 8 full
// reductions.  Returns the total of v[(lane >> 3) & 7] (the same in the 8 lanes of a group).
__device__ __forceinline__ float wave_sum8(const float (&v)[8], int lane) {
  float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3], b0 = v[4], b1 = v[5], b2 = v[6], b3 = v[7];
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\t"
      "v_permlane32_swap_b32 %3, %7\n\ts_nop 1"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
  float c0 = a0 + b0, c1 = a1 + b1, d0 = a2 + b2, d1 = a3 + b3;   // lanes < 32: values 0..3, lanes >= 32: values 4..7
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\ts_nop 1"
      : "+v"(c0), "+v"(c1), "+v"(d0), "+v"(d1));
  const float e0 = c0 + d0, e1 = c1 + d1;                             // even rows: value 2 * bit5 .. + 0 / 1, odd rows: + 2 / 3
  const bool b3_ = lane & 8;
  const float keep = b3_ ? e1 : e0, send = b3_ ? e0 : e1;
  float t = keep + dpp_mov<kDppRor8>(send);
  t += dpp_mov<kDppXor1>(t);
  t += dpp_mov<kDppXor2>(t);
  t += dpp_mov<kDppHalfMirror>(t);
  return t;
}
#pragma clang diagnostic pop

template <typename E> struct Row8 { typedef E type __attribute__((ext_vector_type(8))); };

// The rows stay PACKED in registers between the passes: without this the compiler keeps the fp32 conversions of one pass
// alive for the next (8 instead of 4 registers per row and lane) and spills.
template <typename V, int R>
__device__ __forceinline__ void keep_packed(V (&row)[R]) {
#pragma unroll
  for (int i = 0; i < R; ++i) asm volatile("" : "+v"(row[i]));
}

// fp32 pairs: the per-row arithmetic is issue-bound, v_pk_{fma,mul,add}_f32 halves it
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float a) { return f32x2{a, a}; }
template <typename V>
__device__ __forceinline__ void unpack8(const V& r, f32x2 (&v)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = f32x2{(float)r[2 * k], (float)r[2 * k + 1]};
}
__device__ __forceinline__ void load8x2(const float* p, f32x2 (&v)[4]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  v[0] = f32x2{a[0], a[1]}; v[1] = f32x2{a[2], a[3]}; v[2] = f32x2{b[0], b[1]}; v[3] = f32x2{b[2], b[3]};
}
__device__ __forceinline__ void store8x2(float* p, const f32x2 (&v)[4]) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0][0], v[0][1], v[1][0], v[1][1]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[2][0], v[2][1], v[3][0], v[3][1]};
}
__device__ __forceinline__ float dot8(const f32x2 (&a)[4], const f32x2 (&b)[4]) {
  f32x2 t = a[0] * b[0];
  t = fma2(a[1], b[1], t);
  t = fma2(a[2], b[2], t);
  t = fma2(a[3], b[3], t);
  return t[0] + t[1];
}
// centred row: (x - mu) on the lanes that own columns, 0 elsewhere
template <typename V>
__device__ __forceinline__ void centred(const V& r, float mu, bool act, f32x2 (&v)[4]) {
  unpack8(r, v);
  const f32x2 m = splat2(act ? mu : 0.f);
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = v[k] - m;
}

// r'_h = gamma * r_h into registers, and this lane group's c_h = r_h . beta (the constant LN's shift adds to a score)
__device__ __forceinline__ float load_rprime(const float* __restrict__ Rf, const float* __restrict__ gamma,
                                             const float* __restrict__ beta, int H, int d, int c, bool act, int lane,
                                             f32x2 (&rp)[kMaxH][4]) {
  f32x2 g[4], b[4];
  float cmine = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) g[k] = b[k] = splat2(0.f);
  if (act) { load8x2(gamma + c, g); load8x2(beta + c, b); }
#pragma unroll
  for (int h = 0; h < kMaxH; ++h) {
    float cb = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) rp[h][k] = splat2(0.f);
    if (h < H) {
      if (act) {
        f32x2 rr[4];
        load8x2(Rf + (int64_t)h * d + c, rr);
        cb = dot8(rr, b);
#pragma unroll
        for (int k = 0; k < 4; ++k) rp[h][k] = rr[k] * g[k];
      }
      cb = wave_sum_dpp(cb);
    }
    if (((lane >> 3) & 7) == h) cmine = cb;
  }
  return cmine;
}

// ------------------------------------------------------------------------------------------------------ forward
// LDS: sc [NP][8] scores -> probabilities | st [NP][2] (mean, rstd) | red [8 waves][H * d]
template <typename E, int RPW>
__global__ __launch_bounds__(kThreads) void attn_cls_fwd_kernel(ClsParams p) {
  typedef typename Row8<E>::type v8;
  constexpr int NP = RPW * kWaves;                        // rows a workgroup holds in registers at a time
  static_assert(NP <= 256 && RPW % kGroup == 0 && RPW <= 64, "row grouping");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int f = blockIdx.x, N = p.N, d = p.d, H = p.H;
  // Sequences longer than NP rows (the 325-token frames of BASELINE configs[4]) are processed in NC chunks of NP rows:
  // every pass walks the chunks and (NC > 1 only) requests its rows again -- they come from L2 / the Infinity Cache.
  const int NC = (N + NP - 1) / NP, NT = NC * NP;
  float* sc = lds;                                       // one spare row each: where the rows behind N are written
  float* st = sc + (NT + 1) * 8;
  float* red = st + (NT + 1) * 2 + 6;                     // 16-byte aligned
  const int c = lane * 8;
  const bool act = c < d;
  const float inv_d = 1.0f / (float)d;
  const E* xf = (const E*)p.x + (int64_t)f * p.xs0;

  v8 row[RPW];
  auto request = [&](int base) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int j = base + w + kWaves * i;
      v8 z = {};
      row[i] = z;
      if (act && j < N) row[i] = *reinterpret_cast<const v8*>(xf + (int64_t)j * p.xs1 + c);
    }
  };
  request(0);

  f32x2 rp[kMaxH][4];
  const float cmine = load_rprime(p.R + (int64_t)f * H * d, p.gamma, p.beta, H, d, c, act, lane, rp);
  for (int i = N * 8 + tid; i < (NT + 1) * 8; i += kThreads) sc[i] = 0.f;      // rows behind N: zero probabilities,
  for (int i = N * 2 + tid; i < (NT + 1) * 2; i += kThreads) st[i] = 0.f;      // zero statistics
  __syncthreads();

  // pass A: statistics and the H scores of every row.  Branch-free inside a group of rows (a row behind N is all zeros
  // and lands in the spare LDS row): the compiler then overlaps the rows' reductions and LDS traffic
  for (int ch = 0; ch < NC; ++ch) {
    const int base = ch * NP;
    if (ch) { keep_packed(row); request(base); }
    float mu_keep = 0.f, rs_keep = 0.f;
#pragma unroll
    for (int g0 = 0; g0 < RPW; g0 += kGroup)
      if (base + w + kWaves * g0 < N) {
#pragma unroll
        for (int u = 0; u < kGroup; ++u) {
          const int i = g0 + u, j = base + w + kWaves * i;
          const int jj = j < N ? j : NT;
          f32x2 v[4];
          unpack8(row[i], v);
          const f32x2 s2 = (v[0] + v[1]) + (v[2] + v[3]);
          const float mu = wave_sum_dpp(s2[0] + s2[1]) * inv_d;
          const f32x2 m = splat2(act ? mu : 0.f);
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = v[k] - m;
          const float q = dot8(v, v);
          float dots[8];
#pragma unroll
          for (int h = 0; h < kMaxH; ++h) dots[h] = dot8(rp[h], v);
          const float rs = rsqrtf(wave_sum_dpp(q) * inv_d + p.eps);
          const float t = wave_sum8(dots, lane);
          sc[jj * 8 + (lane >> 3)] = fmaf(rs, t, cmine);       // the 8 lanes of a group store the same value
          st[2 * jj] = mu;
          st[2 * jj + 1] = rs;
          mu_keep = lane == i ? mu : mu_keep;
          rs_keep = lane == i ? rs : rs_keep;
        }
      }
    if (lane < RPW && base + w + kWaves * lane < N) {
      p.mean[(int64_t)f * N + base + w + kWaves * lane] = mu_keep;
      p.rstd[(int64_t)f * N + base + w + kWaves * lane] = rs_keep;
    }
  }
  __syncthreads();
  keep_packed(row);

  // softmax over the rows, one wave per head
  if (w < H) {
    float e[8], mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int j = lane + 64 * t;
      e[t] = j < N ? sc[j * 8 + w] : -INFINITY;
      mx = fmaxf(mx, e[t]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      e[t] = __builtin_amdgcn_exp2f((e[t] - mx) * 1.44269504088896340736f);   // exp2(-inf) = 0 for the rows beyond N
      sum += e[t];
    }
    sum = wave_sum_dpp(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int j = lane + 64 * t;
      if (j < N) sc[j * 8 + w] = e[t] * inv;
    }
    if (lane == 0) p.lse[(int64_t)f * H + w] = mx + __logf(sum);
  }
  __syncthreads();
  for (int i = tid; i < N * 8; i += kThreads) p.P[(int64_t)f * N * 8 + i] = sc[i];      // kept for the backward

  // pass B: a_h = sum_j p_jh n_j over this wave's rows
  f32x2 a[kMaxH][4];
#pragma unroll
  for (int h = 0; h < kMaxH; ++h)
#pragma unroll
    for (int k = 0; k < 4; ++k) a[h][k] = splat2(0.f);
  for (int ch = 0; ch < NC; ++ch) {
    const int base = ch * NP;
    if (NC > 1) { keep_packed(row); request(base); }
#pragma unroll
    for (int g0 = 0; g0 < RPW; g0 += kGroup)
      if (base + w + kWaves * g0 < N) {
#pragma unroll
        for (int u = 0; u < kGroup; ++u) {
          const int i = g0 + u, j = base + w + kWaves * i;     // j < NT: the rows behind N carry zero probabilities
          const float mu = st[2 * j], rs = st[2 * j + 1];
          const f32x4 p0 = *reinterpret_cast<const f32x4*>(sc + j * 8), p1 = *reinterpret_cast<const f32x4*>(sc + j * 8 + 4);
          const float ph[8] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
          f32x2 n[4];
          centred(row[i], mu, act, n);
#pragma unroll
          for (int h = 0; h < kMaxH; ++h) {
            const f32x2 t = splat2(ph[h] * rs);
#pragma unroll
            for (int k = 0; k < 4; ++k) a[h][k] = fma2(t, n[k], a[h][k]);
          }
        }
      }
  }
  // the waves' partial sums, added in wave order (fixed order: reproducible)
  const int HD = H * d;
  if (act) {
#pragma unroll
    for (int h = 0; h < kMaxH; ++h)
      if (h < H) store8x2(red + (int64_t)w * HD + h * d + c, a[h]);
  }
  __syncthreads();
  float* Af = p.A + (int64_t)f * HD;
  for (int e = tid; e < HD; e += kThreads) {
    float s = red[e];
#pragma unroll
    for (int ww = 1; ww < kWaves; ++ww) s += red[ww * HD + e];
    Af[e] = s;
  }
}

// ------------------------------------------------------------------------------------------------------ backward
// Given dm_h (gradient of m_h) per head:  dp_jh = dm_h . LN(x_j),  ds_jh = p_jh (dp_jh - sum_i p_ih dp_ih),
//   d LN(x_j) = sum_h p_jh dm_h + ds_jh r_h,   dr_h = gamma * G_h with G_h = sum_j ds_jh n_j,
//   dgamma = sum_h dm_h * A_h + r_h * G_h,  dbeta = sum_h dm_h  (sum_j p = 1, sum_j ds = 0),
// and dx_j = the LayerNorm backward of d LN(x_j).
// LDS: pd [NP][8][2] (p, then dp -> ds) | st [NP][2] | big = { rl [8][d], dl [8][d] } then { red [8 waves][H * d] }
template <typename E, int RPW>
__global__ __launch_bounds__(kThreads) void attn_cls_bwd_kernel(ClsParams p) {
  typedef typename Row8<E>::type v8;
  constexpr int NP = RPW * kWaves;
  constexpr int GR = kGroup;                             // rows per group in the d LN(x) pass
  static_assert(NP <= 256 && RPW % GR == 0, "row grouping");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int f = blockIdx.x, N = p.N, d = p.d, H = p.H, HD = H * d;
  const int NC = (N + NP - 1) / NP, NT = NC * NP;         // chunks of NP rows, as in the forward
  float* pd = lds;                                       // one spare row each: where the rows behind N are written
  float* st = pd + (NT + 1) * 16;
  float* big = st + (NT + 1) * 2 + 2;                     // 16-byte aligned
  const int c = lane * 8;
  const bool act = c < d;
  const float inv_d = 1.0f / (float)d;
  const E* xf = (const E*)p.x + (int64_t)f * p.xs0;
  E* dxf = (E*)p.dx + (int64_t)f * p.xs0;
  const float* Rf = p.R + (int64_t)f * HD;
  const float* dMf = p.dM + (int64_t)f * HD;
  float* rl = big;
  float* dl = big + kMaxH * d;

  v8 row[RPW];
  auto request = [&](int base) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int j = base + w + kWaves * i;
      v8 z = {};
      row[i] = z;
      if (act && j < N) row[i] = *reinterpret_cast<const v8*>(xf + (int64_t)j * p.xs1 + c);
    }
  };
  request(0);
  // gamma * r_h and gamma * dm_h for the d LN(x) pass (zero rows for h >= H), the rows' statistics
  for (int e = tid; e < kMaxH * d; e += kThreads) {
    const float g = p.gamma[e % d];
    rl[e] = e < HD ? Rf[e] * g : 0.f;
    dl[e] = e < HD ? dMf[e] * g : 0.f;
  }
  for (int j = tid; j <= NT; j += kThreads) {
    st[2 * j] = j < N ? p.mean[(int64_t)f * N + j] : 0.f;
    st[2 * j + 1] = j < N ? p.rstd[(int64_t)f * N + j] : 0.f;
  }
  for (int i = tid; i < (NT + 1) * 8; i += kThreads) {                                   // p_jh of the forward; zero behind N
    pd[2 * i] = i < N * 8 ? p.P[(int64_t)f * N * 8 + i] : 0.f;
    pd[2 * i + 1] = 0.f;
  }
  const int hmine = (lane >> 3) & 7;
  __syncthreads();

  f32x2 rp[kMaxH][4];
  // sweep 2: dp_jh = dm_h . LN(x_j) up to a per-head constant (dm_h . beta), which the softmax backward cancels
#pragma unroll
  for (int h = 0; h < kMaxH; ++h) {
#pragma unroll
    for (int k = 0; k < 4; ++k) rp[h][k] = splat2(0.f);
    if (act) load8x2(dl + h * d + c, rp[h]);
  }
  for (int ch = 0; ch < NC; ++ch) {
    const int base = ch * NP;
    if (ch) { keep_packed(row); request(base); }
#pragma unroll
    for (int g0 = 0; g0 < RPW; g0 += kGroup)
      if (base + w + kWaves * g0 < N) {
#pragma unroll
        for (int u = 0; u < kGroup; ++u) {
          const int i = g0 + u, j = base + w + kWaves * i;
          const int jj = j < N ? j : NT;
          const float mu = st[2 * j], rs = st[2 * j + 1];
          f32x2 v[4];
          centred(row[i], mu, act, v);
          float dots[8];
#pragma unroll
          for (int h = 0; h < kMaxH; ++h) dots[h] = dot8(rp[h], v);
          const float t = wave_sum8(dots, lane);
          pd[(jj * 8 + hmine) * 2 + 1] = rs * t;               // the 8 lanes of a group store the same value
          if (u & 1) __builtin_amdgcn_sched_barrier(0);         // two rows in flight: all five would not fit the registers
        }
      }
  }
  __syncthreads();

  keep_packed(row);
  // softmax backward, one wave per head: ds = p (dp - sum p dp)
  if (w < H) {
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int j = lane + 64 * t;
      if (j < N) acc = fmaf(pd[(j * 8 + w) * 2], pd[(j * 8 + w) * 2 + 1], acc);
    }
    const float D = wave_sum_dpp(acc);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int j = lane + 64 * t;
      if (j < N) pd[(j * 8 + w) * 2 + 1] = pd[(j * 8 + w) * 2] * (pd[(j * 8 + w) * 2 + 1] - D);
    }
  }
  __syncthreads();

  // d LN(x_j) = sum_h p_jh (gamma dm_h) + ds_jh (gamma r_h), then the LayerNorm backward of the row
  // where the rows behind N and the lanes behind d store: this sequence's own partial row (written for real at the very end)
  E* const trash = reinterpret_cast<E*>(p.partial + (int64_t)f * 2 * d) + ((w * 16 + (lane & 15)) * 8) % (4 * d);
  for (int ch = 0; ch < NC; ++ch) {
  const int base = ch * NP;
  if (NC > 1) { keep_packed(row); request(base); }
#pragma unroll
  for (int g0 = 0; g0 < RPW; g0 += GR)
    if (base + w + kWaves * g0 < N) {
      f32x2 dn[GR][4];
#pragma unroll
      for (int u = 0; u < GR; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) dn[u][k] = splat2(0.f);
#pragma unroll
      for (int h = 0; h < kMaxH; ++h)
        if (h < H) {                                           // a (uniform) branch per head on purpose: it keeps the eight heads'
        f32x2 rr[4], dd[4];                                    // vectors (128 registers) from being requested all at once
        int off = h * d + (act ? c : 0);
        asm volatile("" : "+v"(off));                          // ... and from being kept across the groups
        load8x2(rl + off, rr);
        load8x2(dl + off, dd);
#pragma unroll
        for (int u = 0; u < GR; ++u) {
          const int j = base + w + kWaves * (g0 + u);          // zero coefficients behind N and behind H
          const f32x2 cf = *reinterpret_cast<const f32x2*>(pd + (j * 8 + h) * 2);
          const f32x2 cp = splat2(act ? cf[0] : 0.f), cs = splat2(act ? cf[1] : 0.f);
#pragma unroll
          for (int k = 0; k < 4; ++k) dn[u][k] = fma2(cp, dd[k], fma2(cs, rr[k], dn[u][k]));
        }
      }
#pragma unroll
      for (int u = 0; u < GR; ++u) {
        const int j = base + w + kWaves * (g0 + u);
        const float mu = st[2 * j], rs = st[2 * j + 1];
        f32x2 n[4];
        centred(row[g0 + u], mu, act, n);
        const f32x2 r2 = splat2(rs);
#pragma unroll
        for (int k = 0; k < 4; ++k) n[k] = n[k] * r2;
        const f32x2 t1 = (dn[u][0] + dn[u][1]) + (dn[u][2] + dn[u][3]);
        const float c1 = wave_sum_dpp(t1[0] + t1[1]) * inv_d, c2 = wave_sum_dpp(dot8(dn[u], n)) * inv_d;
        const f32x2 c1v = splat2(c1), c2v = splat2(-c2);
        float o[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const f32x2 t = r2 * fma2(n[k], c2v, dn[u][k] - c1v);
          o[2 * k] = t[0];
          o[2 * k + 1] = t[1];
        }
        store8<E>(act && j < N ? dxf + (int64_t)j * p.xs1 + c : trash, o);
        if (u & 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  keep_packed(row);
  // G_h = sum_j ds_jh n_j
  f32x2 a[kMaxH][4];
#pragma unroll
  for (int h = 0; h < kMaxH; ++h)
#pragma unroll
    for (int k = 0; k < 4; ++k) a[h][k] = splat2(0.f);
  for (int ch = 0; ch < NC; ++ch) {
  const int base = ch * NP;
  if (NC > 1) { keep_packed(row); request(base); }
#pragma unroll
  for (int g0 = 0; g0 < RPW; g0 += kGroup)
    if (base + w + kWaves * g0 < N) {
#pragma unroll
      for (int u = 0; u < kGroup; ++u) {
        const int i = g0 + u, j = base + w + kWaves * i;
        const float mu = st[2 * j], rs = st[2 * j + 1];
        f32x2 n[4];
        centred(row[i], mu, act, n);
#pragma unroll
        for (int h = 0; h < kMaxH; ++h) {
          const f32x2 t = splat2(pd[(j * 8 + h) * 2 + 1] * rs);
#pragma unroll
          for (int k = 0; k < 4; ++k) a[h][k] = fma2(t, n[k], a[h][k]);
        }
        if (u & 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __syncthreads();                                       // rl / dl are dead: the region becomes the waves' partial sums
  float* red = big;
  if (act) {
#pragma unroll
    for (int h = 0; h < kMaxH; ++h)
      if (h < H) store8x2(red + (int64_t)w * HD + h * d + c, a[h]);
  }
  __syncthreads();
  float* Gf = p.G + (int64_t)f * HD;
  const float* Af = p.A + (int64_t)f * HD;
  float* part = p.partial + (int64_t)f * 2 * d;
  for (int cc = tid; cc < d; cc += kThreads) {
    float dg = 0.f, db = 0.f;
    for (int h = 0; h < H; ++h) {
      const int e = h * d + cc;
      float s = red[e];
#pragma unroll
      for (int ww = 1; ww < kWaves; ++ww) s += red[ww * HD + e];
      Gf[e] = s;
      dg = fmaf(dMf[e], Af[e], fmaf(Rf[e], s, dg));
      db += dMf[e];
    }
    part[cc] = dg;
    part[d + cc] = db;
  }
}

constexpr int kRPW = 25;                                  // 200 rows per chunk (197 tokens of a 224^2 frame at patch 16)
constexpr int kMaxChunks = 2;                             // N <= 400 (325 tokens of a 288^2 frame: two chunks)

size_t cls_fwd_lds(int N, int H, int d) {
  const int nt = (N + kRPW * kWaves - 1) / (kRPW * kWaves) * (kRPW * kWaves);
  return sizeof(float) * ((size_t)(nt + 1) * 10 + 6 + (size_t)kWaves * H * d);
}
size_t cls_bwd_lds(int N, int H, int d) {
  const int nt = (N + kRPW * kWaves - 1) / (kRPW * kWaves) * (kRPW * kWaves);
  const size_t big = (size_t)kWaves * H * d > (size_t)2 * kMaxH * d ? (size_t)kWaves * H * d : (size_t)2 * kMaxH * d;
  return sizeof(float) * ((size_t)(nt + 1) * 18 + 2 + big);
}

int cls_check(const char* name, const dvt_attn_cls_desc* q, bool bwd) {
  DVT_REQUIRE(q, "%s: null descriptor", name);
  DVT_REQUIRE(q->x && q->gamma && q->beta && q->R && q->A && q->lse && q->P && q->mean && q->rstd, "%s: null pointer", name);
  DVT_REQUIRE(q->S > 0 && q->N > 0 && q->d > 0 && q->H > 0, "%s: bad sizes", name);
  if (!dvt_attn_cls_supported(q))
    DVT_UNSUPPORTED("%s: needs a 16-bit dtype, d %% 8 == 0, d <= 512, H <= 8, N <= %d, strides %% 8 == 0 (got d = %lld, H = %lld, "
                    "N = %lld)", name, kMaxChunks * kRPW * kWaves, (long long)q->d, (long long)q->H, (long long)q->N);
  DVT_REQUIRE(dvt_aligned16(q->x) && dvt_aligned16(q->gamma) && dvt_aligned16(q->beta) && dvt_aligned16(q->R) &&
                  dvt_aligned16(q->A), "%s: buffers must be 16-byte aligned", name);
  if (bwd) {
    DVT_REQUIRE(q->dM && q->dx && q->G && q->dgamma && q->dbeta && q->workspace, "%s: null pointer (backward operands)", name);
    DVT_REQUIRE(dvt_aligned16(q->dM) && dvt_aligned16(q->dx) && dvt_aligned16(q->G) && dvt_aligned16(q->workspace),
                "%s: buffers must be 16-byte aligned", name);
  }
  return DVT_OK;
}

ClsParams cls_params(const dvt_attn_cls_desc* q) {
  ClsParams p;
  p.x = q->x; p.xs0 = q->xs0; p.xs1 = q->xs1; p.gamma = q->gamma; p.beta = q->beta; p.eps = q->eps;
  p.S = (int)q->S; p.N = (int)q->N; p.d = (int)q->d; p.H = (int)q->H;
  p.R = q->R; p.A = q->A; p.lse = q->lse; p.P = q->P; p.mean = q->mean; p.rstd = q->rstd;
  p.dM = q->dM; p.dx = q->dx; p.G = q->G; p.partial = (float*)q->workspace;
  return p;
}

// ------------------------------------------------------------------------------------------------------ head-wise products
// 67 MFLOP each at the metric shape: what matters is that every workgroup pays ONE memory round trip (all of its operand
// requests in flight before the first use) and that the grid fills the chip.
//
// out[f, h, c] = alpha sum_e in[f, h dh + e] W[h dh + e, c]: 16 sequences x one head x 128 columns per workgroup; a thread
// owns two columns of four sequences and holds its 64 weight pairs in registers.
template <typename E>
__device__ __forceinline__ void heads_expand_tile(const int bx, const int by, const int bz, float* lds /* [dh rounded up to 64][16] */,
                                                  const E* __restrict__ in, int64_t ld_in, const E* __restrict__ W, int64_t ldw,
                                                  float* __restrict__ out, int S, int H, int dh, int d, float alpha) {
  typedef E v2 __attribute__((ext_vector_type(2)));
  const int f0 = bx * 16, h = by, tid = threadIdx.x;              // lds: a[e][f], zero rows behind dh
  const int c = bz * 128 + (tid & 63) * 2, fg = (tid >> 6) * 4;
  const bool cok = c < d;
  const int dhp = (dh + 63) & ~63;
  f32x2 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x2{0.f, 0.f};
  const E* Wh = W + (int64_t)h * dh * ldw + (cok ? c : 0);
  for (int e0 = 0; e0 < dh; e0 += 64) {
    v2 w[64];                                            // unconditional requests (clamped rows), all in flight together
#pragma unroll
    for (int e = 0; e < 64; ++e) w[e] = *reinterpret_cast<const v2*>(Wh + (int64_t)min(e0 + e, dh - 1) * ldw);
    __builtin_amdgcn_sched_barrier(0);
    if (e0 == 0) {                                       // the activations' round trip runs under the weights'
      for (int i = tid; i < 16 * dhp; i += 256) {
        const int ee = i >> 4, ff = i & 15;
        lds[i] = f0 + ff < S && ee < dh ? to_f32<E>(in[(int64_t)(f0 + ff) * ld_in + h * dh + ee]) : 0.f;
      }
      __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < 64; ++e) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(lds + (e0 + e) * 16 + fg);    // zero behind dh
      const f32x2 wv = f32x2{(float)w[e][0], (float)w[e][1]};
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_elementwise_fma(f32x2{a4[i], a4[i]}, wv, acc[i]);
    }
  }
  if (cok) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (f0 + fg + i < S) *reinterpret_cast<f32x2*>(out + ((int64_t)(f0 + fg + i) * H + h) * d + c) = acc[i] * f32x2{alpha, alpha};
  }
}

// out[f, h dh + e] = alpha sum_c v[f, h, c] W[h dh + e, c],  v = gamma * in + beta (either NULL: in itself); d <= 512.
// 8 sequences x one head per workgroup; a lane owns 8 columns of the 8 sequences (64 registers), a wave takes the output
// columns e = wave, wave + 4, ... in rounds of 16 weight rows requested together.
template <typename E>
__device__ __forceinline__ void heads_contract_tile(const int bx, const int by, const float* __restrict__ in,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const E* __restrict__ W, int64_t ldw, E* __restrict__ out, int64_t ld_out, int S,
                                                    int H, int dh, int d, float alpha) {
  typedef typename Row8<E>::type v8;
  const int f0 = bx * 8, h = by, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane * 8;
  const bool act = c < d;
  f32x2 v[8][4];
  f32x2 g[4], b[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { g[k] = f32x2{1.f, 1.f}; b[k] = f32x2{0.f, 0.f}; }
  if (act && gamma) load8x2(gamma + c, g);
  if (act && beta) load8x2(beta + c, b);
  v8 wr[16];                                             // the first 16 weight rows of this wave: requested with the operands
#pragma unroll
  for (int i = 0; i < 16; ++i)
    wr[i] = *reinterpret_cast<const v8*>(W + (int64_t)(h * dh + min(w + 4 * i, dh - 1)) * ldw + (act ? c : 0));
#pragma unroll
  for (int ff = 0; ff < 8; ++ff)                         // unconditional requests (clamped), all in flight together
    load8x2(in + ((int64_t)min(f0 + ff, S - 1) * H + h) * d + (act ? c : 0), v[ff]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ff = 0; ff < 8; ++ff)
#pragma unroll
    for (int k = 0; k < 4; ++k) v[ff][k] = act && f0 + ff < S ? __builtin_elementwise_fma(v[ff][k], g[k], b[k]) : f32x2{0.f, 0.f};
  for (int e0 = w; e0 < dh; e0 += 64) {
    if (e0 != w) {                                       // further rounds (dh > 64): unconditional requests, all in flight
#pragma unroll
      for (int i = 0; i < 16; ++i)
        wr[i] = *reinterpret_cast<const v8*>(W + (int64_t)(h * dh + min(e0 + 4 * i, dh - 1)) * ldw + (act ? c : 0));
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (e0 + 4 * i < dh) {
        f32x2 wv[4];
        unpack8(wr[i], wv);
        if (!act) wv[0] = wv[1] = wv[2] = wv[3] = f32x2{0.f, 0.f};
        float acc[8];
#pragma unroll
        for (int ff = 0; ff < 8; ++ff) acc[ff] = dot8(v[ff], wv);
        const float t = wave_sum8(acc, lane);
        const int ff = (lane >> 3) & 7;
        if ((lane & 7) == 0 && f0 + ff < S) out[(int64_t)(f0 + ff) * ld_out + h * dh + e0 + 4 * i] = from_f32<E>(alpha * t);
      }
  }
}

// dW[h dh + e, c] (+)= alpha sum_f a[f, h dh + e] v[f, h, c],  v = gamma * b + beta (either NULL: b itself).
// One (head, 64 columns, 16 rows e) tile per workgroup (256 workgroups at the metric shape); sequences staged 128 at a
// time through LDS, all of a chunk's requests in flight together; a thread owns one column of four rows.
template <typename E>
__device__ __forceinline__ void heads_outer_tile(const int bx, const int by, const int bz, float (*vt)[64], float (*at)[16],
                                                 const E* __restrict__ a, int64_t lda, const float* __restrict__ b,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 float* __restrict__ dW, int64_t ldw, int S, int H, int dh, int d, float alpha,
                                                 int accumulate) {
  const int h = bx, c0 = by * 64, e0 = bz * 16, tid = threadIdx.x;
  const int cl = tid & 63, eg = (tid >> 6) * 4;
  const int c4 = (tid & 15) * 4;                        // staging: this thread's four columns of rows tid / 16 + 16 k
  f32x4 g4 = {1.f, 1.f, 1.f, 1.f}, b4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (c0 + c4 + k < d) {
      if (gamma) g4[k] = gamma[c0 + c4 + k];
      if (beta) b4[k] = beta[c0 + c4 + k];
    }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int fb = 0; fb < S; fb += 128) {
    f32x4 st[8];                                         // unconditional requests (clamped rows / columns)
    const bool colok = c0 + c4 < d;                      // d % 4 == 0: a thread's four columns are valid together
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ff = min(fb + (tid >> 4) + 16 * k, S - 1);
      st[k] = *reinterpret_cast<const f32x4*>(b + ((int64_t)ff * H + h) * d + (colok ? c0 + c4 : 0));
    }
    float sa[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = tid + 256 * k, ff = min(fb + (i >> 4), S - 1), ee = min(e0 + (i & 15), dh - 1);
      sa[k] = to_f32<E>(a[(int64_t)ff * lda + h * dh + ee]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (fb) __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ff = (tid >> 4) + 16 * k;
      f32x4 t = st[k] * g4 + b4;
      if (fb + ff >= S || !colok) t = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(&vt[ff][c4]) = t;
      const int i = tid + 256 * k;
      at[i >> 4][i & 15] = fb + (i >> 4) < S && e0 + (i & 15) < dh ? sa[k] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int ff = 0; ff < 128; ++ff) {
      const float vv = vt[ff][cl];
      const f32x4 aa = *reinterpret_cast<const f32x4*>(&at[ff][eg]);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = fmaf(aa[i], vv, acc[i]);
    }
  }
  if (c0 + cl < d) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (e0 + eg + i < dh) {
        float* o = dW + (int64_t)(h * dh + e0 + eg + i) * ldw + c0 + cl;
        *o = accumulate ? *o + alpha * acc[i] : alpha * acc[i];
      }
  }
}

template <typename E>
__global__ __launch_bounds__(256) void heads_expand_kernel(const E* in, int64_t ld_in, const E* W, int64_t ldw, float* out, int S, int H,
                                                           int dh, int d, float alpha) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  heads_expand_tile<E>(blockIdx.x, blockIdx.y, blockIdx.z, lds, in, ld_in, W, ldw, out, S, H, dh, d, alpha);
}

template <typename E>
__global__ __launch_bounds__(256) void heads_contract_kernel(const float* in, const float* gamma, const float* beta, const E* W,
                                                             int64_t ldw, E* out, int64_t ld_out, int S, int H, int dh, int d,
                                                             float alpha) {
  heads_contract_tile<E>(blockIdx.x, blockIdx.y, in, gamma, beta, W, ldw, out, ld_out, S, H, dh, d, alpha);
}

template <typename E>
__global__ __launch_bounds__(256) void heads_outer_kernel(const E* a, int64_t lda, const float* b, const float* gamma,
                                                          const float* beta, float* dW, int64_t ldw, int S, int H, int dh, int d,
                                                          float alpha, int accumulate) {
  __shared__ __attribute__((aligned(16))) float vt[128][64];
  __shared__ __attribute__((aligned(16))) float at[128][16];
  heads_outer_tile<E>(blockIdx.x, blockIdx.y, blockIdx.z, vt, at, a, lda, b, gamma, beta, dW, ldw, S, H, dh, d, alpha, accumulate);
}

// The two products in front of the row pass of the backward (dm = expand(do, Wv), dWv (+)= outer(do, gamma A + beta)) and
// the two behind it (dq = contract(gamma G, Wk), dWk (+)= outer(q, gamma G)) are independent of each other: one launch
// per pair, the workgroups of the first member in front (a dependent launch costs ~2.7 us inside the replayed graph).
struct HeadsPair {
  // first member: expand (kind 0) or contract (kind 1)
  const void* in16; int64_t ld_in; const float* vin; const float* gamma1; const float* beta1;
  const void* W; int64_t ldw; void* out; int64_t ld_out; float alpha1;
  int gx, gy, gz;                                        // its grid
  // second member: outer
  const void* a; int64_t lda; const float* v; const float* gamma2; const float* beta2; float* dW; int64_t ld_dw; float alpha2;
  int accumulate; int ox, oy, oz;
  int S, H, dh, d;
};

template <typename E, int KIND>
__global__ __launch_bounds__(256) void heads_pair_kernel(const HeadsPair q) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ __attribute__((aligned(16))) float vt[128][64];
  __shared__ __attribute__((aligned(16))) float at[128][16];
  const int n1 = q.gx * q.gy * q.gz;
  int b = blockIdx.x;
  if (b < n1) {
    const int bx = b % q.gx, by = (b / q.gx) % q.gy, bz = b / (q.gx * q.gy);
    if (KIND == 0) heads_expand_tile<E>(bx, by, bz, lds, (const E*)q.in16, q.ld_in, (const E*)q.W, q.ldw, (float*)q.out, q.S, q.H, q.dh, q.d, q.alpha1);
    else heads_contract_tile<E>(bx, by, q.vin, q.gamma1, q.beta1, (const E*)q.W, q.ldw, (E*)q.out, q.ld_out, q.S, q.H, q.dh, q.d, q.alpha1);
  } else {
    b -= n1;
    const int bx = b % q.ox, by = (b / q.ox) % q.oy, bz = b / (q.ox * q.oy);
    heads_outer_tile<E>(bx, by, bz, vt, at, (const E*)q.a, q.lda, q.v, q.gamma2, q.beta2, q.dW, q.ld_dw, q.S, q.H, q.dh, q.d, q.alpha2, q.accumulate);
  }
}

int heads_check(const char* name, const void* a, const void* b, const void* c, int64_t S, int64_t H, int64_t dh, int64_t d,
                int dtype) {
  DVT_REQUIRE(a && b && c, "%s: null pointer", name);
  DVT_REQUIRE(S > 0 && H > 0 && dh > 0 && d > 0, "%s: bad sizes", name);
  if (!dvt_is_16bit(dtype)) DVT_UNSUPPORTED("%s: 16-bit dtypes only", name);
  return DVT_OK;
}

}  // namespace

extern "C" {

int dvt_attn_cls_supported(const dvt_attn_cls_desc* q) {
  if (!q) return 0;
  return dvt_is_16bit(q->dtype) && q->d % 8 == 0 && q->d <= 512 && q->H <= kMaxH && q->N <= kMaxChunks * kRPW * kWaves &&
         q->xs0 % 8 == 0 && q->xs1 % 8 == 0 && q->xs1 >= q->d;
}

size_t dvt_attn_cls_bwd_workspace_bytes(const dvt_attn_cls_desc* q) {
  return q ? (size_t)q->S * 2 * (size_t)q->d * sizeof(float) : 0;
}

int dvt_attn_cls_fwd(const dvt_attn_cls_desc* q, dvt_stream_t stream) {
  int rc = cls_check("dvt_attn_cls_fwd", q, false);
  if (rc) return rc;
  const ClsParams p = cls_params(q);
  const size_t lds = cls_fwd_lds(p.N, p.H, p.d);
  DVT_DISPATCH_16BIT(q->dtype, E, {
    static DvtLdsAttr set;
    dvt_lds_attr(set, (const void*)attn_cls_fwd_kernel<E, kRPW>, 160 * 1024);
    hipLaunchKernelGGL((attn_cls_fwd_kernel<E, kRPW>), dim3((unsigned)p.S), dim3(kThreads), lds, (hipStream_t)stream, p);
  });
  DVT_LAUNCH_CHECK("dvt_attn_cls_fwd");
  return DVT_OK;
}

int dvt_attn_cls_bwd(const dvt_attn_cls_desc* q, dvt_stream_t stream) {
  int rc = cls_check("dvt_attn_cls_bwd", q, true);
  if (rc) return rc;
  const ClsParams p = cls_params(q);
  const size_t lds = cls_bwd_lds(p.N, p.H, p.d);
  DVT_DISPATCH_16BIT(q->dtype, E, {
    static DvtLdsAttr set;
    dvt_lds_attr(set, (const void*)attn_cls_bwd_kernel<E, kRPW>, 160 * 1024);
    hipLaunchKernelGGL((attn_cls_bwd_kernel<E, kRPW>), dim3((unsigned)p.S), dim3(kThreads), lds, (hipStream_t)stream, p);
  });
  DVT_LAUNCH_CHECK("dvt_attn_cls_bwd");
  dvt_ln_partials_reduce(p.partial, p.S, p.d, q->dgamma, q->dbeta, q->accumulate_gamma, q->accumulate_beta, (hipStream_t)stream);
  DVT_LAUNCH_CHECK("dvt_attn_cls_bwd(reduce)");
  return DVT_OK;
}

int dvt_heads_expand(const void* in, int64_t ld_in, const void* W, int64_t ldw, float* out, int64_t S, int64_t H, int64_t dh,
                     int64_t d, float alpha, int dtype, dvt_stream_t stream) {
  int rc = heads_check("dvt_heads_expand", in, W, out, S, H, dh, d, dtype);
  if (rc) return rc;
  DVT_REQUIRE(d % 2 == 0, "dvt_heads_expand: d must be even");
  DVT_REQUIRE(ldw % 2 == 0 && (reinterpret_cast<uintptr_t>(W) & 3u) == 0 && (reinterpret_cast<uintptr_t>(out) & 7u) == 0,
              "dvt_heads_expand: W rows must be 4-byte aligned, out 8-byte aligned");
  DVT_REQUIRE(dh <= 512, "dvt_heads_expand: dh too large");
  DVT_DISPATCH_16BIT(dtype, E, hipLaunchKernelGGL((heads_expand_kernel<E>), dim3((unsigned)dvt_cdiv(S, 16), (unsigned)H,
                                                  (unsigned)dvt_cdiv(d, 128)), dim3(256), 16 * (size_t)((dh + 63) / 64 * 64) * sizeof(float),
                                                  (hipStream_t)stream, (const E*)in, ld_in, (const E*)W, ldw, out, (int)S,
                                                  (int)H, (int)dh, (int)d, alpha));
  DVT_LAUNCH_CHECK("dvt_heads_expand");
  return DVT_OK;
}

int dvt_heads_contract(const float* in, const float* gamma, const float* beta, const void* W, int64_t ldw, void* out,
                       int64_t ld_out, int64_t S, int64_t H, int64_t dh, int64_t d, float alpha, int dtype,
                       dvt_stream_t stream) {
  int rc = heads_check("dvt_heads_contract", in, W, out, S, H, dh, d, dtype);
  if (rc) return rc;
  DVT_REQUIRE(d % 8 == 0 && ldw % 8 == 0 && dvt_aligned16(W) && dvt_aligned16(in), "dvt_heads_contract: d, ldw multiples of 8, "
              "16-byte aligned operands");
  if (d > 512) DVT_UNSUPPORTED("dvt_heads_contract: d = %lld > 512", (long long)d);
  DVT_REQUIRE(dvt_aligned16(gamma) && dvt_aligned16(beta), "dvt_heads_contract: gamma / beta misaligned");
  DVT_DISPATCH_16BIT(dtype, E, hipLaunchKernelGGL((heads_contract_kernel<E>), dim3((unsigned)dvt_cdiv(S, 8), (unsigned)H), dim3(256),
                                                  0, (hipStream_t)stream, in, gamma, beta, (const E*)W, ldw, (E*)out, ld_out,
                                                  (int)S, (int)H, (int)dh, (int)d, alpha));
  DVT_LAUNCH_CHECK("dvt_heads_contract");
  return DVT_OK;
}

int dvt_heads_outer(const void* a, int64_t lda, const float* b, const float* gamma, const float* beta, float* dW, int64_t ldw,
                    int64_t S, int64_t H, int64_t dh, int64_t d, float alpha, int accumulate, int dtype, dvt_stream_t stream) {
  int rc = heads_check("dvt_heads_outer", a, b, dW, S, H, dh, d, dtype);
  if (rc) return rc;
  DVT_REQUIRE(d % 4 == 0 && dvt_aligned16(b), "dvt_heads_outer: d must be a multiple of 4, b 16-byte aligned");
  DVT_DISPATCH_16BIT(dtype, E, hipLaunchKernelGGL((heads_outer_kernel<E>), dim3((unsigned)H, (unsigned)dvt_cdiv(d, 64),
                                                  (unsigned)dvt_cdiv(dh, 16)), dim3(256), 0,
                                                  (hipStream_t)stream, (const E*)a, lda, b, gamma, beta, dW, ldw, (int)S, (int)H,
                                                  (int)dh, (int)d, alpha, accumulate));
  DVT_LAUNCH_CHECK("dvt_heads_outer");
  return DVT_OK;
}

int dvt_heads_expand_outer(const void* in, int64_t ld_in, const void* W, int64_t ldw, float* out, float alpha_out, const float* v,
                           const float* gamma, const float* beta, float* dW, int64_t ld_dw, float alpha_dw, int accumulate,
                           int64_t S, int64_t H, int64_t dh, int64_t d, int dtype, dvt_stream_t stream) {
  int rc = heads_check("dvt_heads_expand_outer", in, W, out, S, H, dh, d, dtype);
  if (rc) return rc;
  DVT_REQUIRE(v && dW, "dvt_heads_expand_outer: null pointer");
  DVT_REQUIRE(d % 4 == 0 && ldw % 2 == 0 && dvt_aligned16(v) && (reinterpret_cast<uintptr_t>(W) & 3u) == 0 &&
                  (reinterpret_cast<uintptr_t>(out) & 7u) == 0,
              "dvt_heads_expand_outer: alignment (see dvt_heads_expand / dvt_heads_outer)");
  // 40 KiB of static LDS (the outer-product tiles) + 16 * ceil64(dh) floats of dynamic LDS must stay inside the 64 KiB a
  // kernel gets without the max-dynamic-LDS attribute
  if (dh > 384) DVT_UNSUPPORTED("dvt_heads_expand_outer: dh = %lld > 384 (LDS budget of the paired launch)", (long long)dh);
  HeadsPair q{};
  q.in16 = in; q.ld_in = ld_in; q.W = W; q.ldw = ldw; q.out = out; q.alpha1 = alpha_out;
  q.gx = (int)dvt_cdiv(S, 16); q.gy = (int)H; q.gz = (int)dvt_cdiv(d, 128);
  q.a = in; q.lda = ld_in; q.v = v; q.gamma2 = gamma; q.beta2 = beta; q.dW = dW; q.ld_dw = ld_dw; q.alpha2 = alpha_dw;
  q.accumulate = accumulate; q.ox = (int)H; q.oy = (int)dvt_cdiv(d, 64); q.oz = (int)dvt_cdiv(dh, 16);
  q.S = (int)S; q.H = (int)H; q.dh = (int)dh; q.d = (int)d;
  const unsigned grid = (unsigned)(q.gx * q.gy * q.gz + q.ox * q.oy * q.oz);
  DVT_DISPATCH_16BIT(dtype, E, hipLaunchKernelGGL((heads_pair_kernel<E, 0>), dim3(grid), dim3(256),
                                                  16 * (size_t)((dh + 63) / 64 * 64) * sizeof(float), (hipStream_t)stream, q));
  DVT_LAUNCH_CHECK("dvt_heads_expand_outer");
  return DVT_OK;
}

int dvt_heads_contract_outer(const float* v, const float* gamma, const void* W, int64_t ldw, void* out, int64_t ld_out,
                             float alpha_out, const void* a, int64_t lda, float* dW, int64_t ld_dw, float alpha_dw,
                             int accumulate, int64_t S, int64_t H, int64_t dh, int64_t d, int dtype, dvt_stream_t stream) {
  int rc = heads_check("dvt_heads_contract_outer", v, W, out, S, H, dh, d, dtype);
  if (rc) return rc;
  DVT_REQUIRE(a && dW, "dvt_heads_contract_outer: null pointer");
  DVT_REQUIRE(d % 8 == 0 && ldw % 8 == 0 && dvt_aligned16(W) && dvt_aligned16(v) && dvt_aligned16(gamma),
              "dvt_heads_contract_outer: alignment (see dvt_heads_contract / dvt_heads_outer)");
  if (d > 512) DVT_UNSUPPORTED("dvt_heads_contract_outer: d = %lld > 512", (long long)d);
  HeadsPair q{};
  q.vin = v; q.gamma1 = gamma; q.beta1 = nullptr; q.W = W; q.ldw = ldw; q.out = out; q.ld_out = ld_out; q.alpha1 = alpha_out;
  q.gx = (int)dvt_cdiv(S, 8); q.gy = (int)H; q.gz = 1;
  q.a = a; q.lda = lda; q.v = v; q.gamma2 = gamma; q.beta2 = nullptr; q.dW = dW; q.ld_dw = ld_dw; q.alpha2 = alpha_dw;
  q.accumulate = accumulate; q.ox = (int)H; q.oy = (int)dvt_cdiv(d, 64); q.oz = (int)dvt_cdiv(dh, 16);
  q.S = (int)S; q.H = (int)H; q.dh = (int)dh; q.d = (int)d;
  const unsigned grid = (unsigned)(q.gx * q.gy * q.gz + q.ox * q.oy * q.oz);
  DVT_DISPATCH_16BIT(dtype, E, hipLaunchKernelGGL((heads_pair_kernel<E, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, q));
  DVT_LAUNCH_CHECK("dvt_heads_contract_outer");
  return DVT_OK;
}

}  // extern "C"
