// head.hip -- the classification head and its loss as one launch, and the grouped scaled store that is its backward.
//
// vit.py:97-100,126-128: logits = Linear(LayerNorm(x[:, 0])) on the temporal Transformer's output, whose own final
// LayerNorm (vit.py:43) commutes with the CLS pooling, so the head is LN -> LN -> Linear on b rows (8 at the metric shape),
// followed by the caller's BCEWithLogitsLoss.  As separate launches that is 5 forward + 6 backward kernels of ~5 us each
// (launch floor of a graph node) around a few kiloflops.  Here ONE workgroup does all of it: a wave owns a row (lane l holds
// columns l, l + 64, ...), both normalisations and the class dot products are wave reductions, the backward through the
// Linear and both LayerNorms runs on the same registers, and the cross-row sums (dgamma, dbeta, dW, dc) are taken by
// column-owner threads from per-row vectors parked in the workspace.  The gradients are those of an upstream gradient of
// 1; backward is the scaled store below (the loss scale / 1 / world factor arrives as a device scalar).
#include "common.h"

namespace {

constexpr int kHeadMaxV = 16;       // d <= 1024
constexpr int kHeadMaxRows = 32;
constexpr int kHeadMaxC = 64;

struct HeadLayout {
  int64_t dx, g1, b1, g2, b2, w, c, s_x1, s_d1, s_x2, s_d2, total;
};
__host__ __device__ inline HeadLayout head_layout(int rows, int d, int C) {
  HeadLayout L;
  int64_t o = 0;
  L.dx = o; o += (int64_t)rows * d;
  L.g1 = o; o += d;
  L.b1 = o; o += d;
  L.g2 = o; o += d;
  L.b2 = o; o += d;
  L.w = o; o += (int64_t)C * d;
  L.c = o; o += (C + 63) / 64 * 64;
  L.s_x1 = o; o += (int64_t)rows * d;
  L.s_d1 = o; o += (int64_t)rows * d;
  L.s_x2 = o; o += (int64_t)rows * d;
  L.s_d2 = o; o += (int64_t)rows * d;
  L.total = o;
  return L;
}

// The weight matrix is staged in LDS by the whole workgroup up front (one memory round trip under the row's own load; the
// class loops then never wait on memory), and NV = d / 64 is a template parameter: the kernel runs ONCE per step from a cold
// instruction cache, so compact straight-line code matters more than anything else in it (a runtime NV with predicated
// 16-fold unrolling was 11 k instructions, 1.2 k branches and 46 us; the arithmetic is a few hundred nanoseconds).
constexpr int kHeadWLds = 12288;      // floats: classes * d up to 48 KiB

template <typename TX, int NV>
__global__ __launch_bounds__(512) void head_bce_kernel(const dvt_head_bce_desc p) {
  __shared__ float s_loss[8];
  __shared__ float s_dz[kHeadMaxRows * kHeadMaxC];
  __shared__ float s_c[kHeadMaxC];
  __shared__ float s_w[kHeadWLds];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  constexpr int d = NV * 64;
  const int C = p.classes;
  const float invd = 1.f / (float)d, gsc = 1.f / ((float)p.rows * (float)C);
  const HeadLayout L = head_layout(p.rows, d, C);
  float* __restrict__ G = p.grads;
  const TX* __restrict__ X = (const TX*)p.x;
  const bool has1 = p.g1 != nullptr;
  float loss_acc = 0.f;

  // every load of the prologue is issued before anything waits: the first row, the four LayerNorm vectors, W, the bias
  float x1[NV], ga1[NV], be1[NV], ga2[NV], be2[NV];
  {
    const int r = min(wave, p.rows - 1);
    const float* g1 = has1 ? p.g1 : p.g2;                // (a valid address either way; unused without the first norm)
    const float* b1 = has1 ? p.b1 : p.b2;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = lane + 64 * i;
      x1[i] = to_f32<TX>(X[(int64_t)r * d + j]);
      ga1[i] = g1[j];
      be1[i] = b1[j];
      ga2[i] = p.g2[j];
      be2[i] = p.b2[j];
    }
  }
  {
    constexpr int kB = 20;         // loads in flight per thread before the first LDS store (a plain copy loop compiles to
    const int n = C * d;           // load - wait - store per element)
    for (int base = 0; base < n; base += kB * (int)blockDim.x) {
      float t[kB];
#pragma unroll
      for (int u = 0; u < kB; ++u) t[u] = p.w[min(base + u * (int)blockDim.x + (int)threadIdx.x, n - 1)];
#pragma unroll
      for (int u = 0; u < kB; ++u) {
        const int i = base + u * (int)blockDim.x + (int)threadIdx.x;
        if (i < n) s_w[i] = t[u];
      }
    }
  }
  for (int i = threadIdx.x; i < kHeadMaxRows * kHeadMaxC; i += blockDim.x) s_dz[i] = 0.f;
  if ((int)threadIdx.x < kHeadMaxC) s_c[threadIdx.x] = (p.c && (int)threadIdx.x < C) ? p.c[threadIdx.x] : 0.f;
  __syncthreads();

  for (int r0 = 0; r0 < p.rows; r0 += nw) {             // the same trip count for every wave: barriers inside are legal
    const bool live = r0 + wave < p.rows;
    const int r = min(r0 + wave, p.rows - 1);            // (a wave without a row shadows the last one and stores nothing)
    float x2[NV], dh[NV];
    float rstd1 = 1.f;
    if (r0 > 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) x1[i] = to_f32<TX>(X[(int64_t)r * d + lane + 64 * i]);
    }
    const float tgt = p.target[(int64_t)r * C + min(lane, C - 1)];
    // ---- first LayerNorm (x1 <- normalised x; x2 <- h1)
    if (has1) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) s += x1[i];
      const float mean = wave_sum_dpp(s) * invd;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) { x1[i] -= mean; q += x1[i] * x1[i]; }
      rstd1 = rsqrtf(wave_sum_dpp(q) * invd + p.eps1);
#pragma unroll
      for (int i = 0; i < NV; ++i) { x1[i] *= rstd1; x2[i] = x1[i] * ga1[i] + be1[i]; }
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) x2[i] = x1[i];
    }
    // ---- second LayerNorm (x2 <- normalised h1; dh <- h2, its affine output, until the logits are done)
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s2 += x2[i];
    const float mean2 = wave_sum_dpp(s2) * invd;
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) { x2[i] -= mean2; q2 += x2[i] * x2[i]; }
    const float rstd2 = rsqrtf(wave_sum_dpp(q2) * invd + p.eps2);
#pragma unroll
    for (int i = 0; i < NV; ++i) { x2[i] *= rstd2; dh[i] = x2[i] * ga2[i] + be2[i]; }
    // ---- Linear: lane c keeps logit c
    float zmine = 0.f;
#pragma unroll 1
    for (int c = 0; c < C; ++c) {
      const float* wr = s_w + c * d + lane;
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) a += dh[i] * wr[64 * i];
      const float z = wave_sum_dpp(a) + s_c[c];
      zmine = lane == c ? z : zmine;
    }
    // ---- BCE on lanes < C, dz through LDS to the whole wave
    float term = fmaxf(zmine, 0.f) - zmine * tgt + log1pf(expf(-fabsf(zmine)));
    term = lane < C ? term : 0.f;
    if (lane < C && live) {
      p.logits[(int64_t)r * C + lane] = zmine;
      s_dz[r * kHeadMaxC + lane] = gsc * (1.0f / (1.0f + expf(-zmine)) - tgt);
    }
    const float tsum = wave_sum_dpp(term);
    loss_acc += live ? tsum : 0.f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) dh[i] = 0.f;
#pragma unroll 1
    for (int c = 0; c < C; ++c) {
      const float* wr = s_w + c * d + lane;
      const float dz = s_dz[r * kHeadMaxC + c];
#pragma unroll
      for (int i = 0; i < NV; ++i) dh[i] += dz * wr[64 * i];
    }
    // ---- second LayerNorm backward: dh = dL/dh2 -> dL/dh1 (kept in dh)
    float m1 = 0.f, m2 = 0.f;
    if (live) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        G[L.s_d2 + (int64_t)r * d + lane + 64 * i] = dh[i];
        G[L.s_x2 + (int64_t)r * d + lane + 64 * i] = x2[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) { dh[i] *= ga2[i]; m1 += dh[i]; m2 += dh[i] * x2[i]; }
    m1 = wave_sum_dpp(m1) * invd;
    m2 = wave_sum_dpp(m2) * invd;
#pragma unroll
    for (int i = 0; i < NV; ++i) dh[i] = rstd2 * (dh[i] - m1 - x2[i] * m2);
    // ---- first LayerNorm backward -> dL/dx
    if (has1) {
      if (live) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          G[L.s_d1 + (int64_t)r * d + lane + 64 * i] = dh[i];
          G[L.s_x1 + (int64_t)r * d + lane + 64 * i] = x1[i];
        }
      }
      m1 = 0.f; m2 = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) { dh[i] *= ga1[i]; m1 += dh[i]; m2 += dh[i] * x1[i]; }
      m1 = wave_sum_dpp(m1) * invd;
      m2 = wave_sum_dpp(m2) * invd;
#pragma unroll
      for (int i = 0; i < NV; ++i) dh[i] = rstd1 * (dh[i] - m1 - x1[i] * m2);
    }
    if (live) {
#pragma unroll
      for (int i = 0; i < NV; ++i) G[L.dx + (int64_t)r * d + lane + 64 * i] = dh[i];
    }
  }
  if (lane == 0) s_loss[wave] = loss_acc;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < nw; ++w) t += s_loss[w];
    p.loss[0] = t * gsc;
  }
  // ---- cross-row sums by column owners, eight rows' loads in flight at a time (rows beyond the last: weight 0)
  for (int j = threadIdx.x; j < d; j += blockDim.x) {
    float dg1 = 0.f, db1 = 0.f, dg2 = 0.f, db2 = 0.f;
    const float gj = p.g2[j], bj = p.b2[j];
    for (int rb = 0; rb < p.rows; rb += 8) {
      float d2[8], xx2[8], d1[8], xx1[8], hh[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t o = (int64_t)min(rb + u, p.rows - 1) * d + j;
        d2[u] = G[L.s_d2 + o];
        xx2[u] = G[L.s_x2 + o];
        d1[u] = G[(has1 ? L.s_d1 : L.s_d2) + o];
        xx1[u] = G[(has1 ? L.s_x1 : L.s_x2) + o];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float m = rb + u < p.rows ? 1.f : 0.f;
        dg2 += m * d2[u] * xx2[u];
        db2 += m * d2[u];
        dg1 += m * d1[u] * xx1[u];
        db1 += m * d1[u];
        hh[u] = m * (xx2[u] * gj + bj);
      }
#pragma unroll 1
      for (int c = 0; c < C; ++c) {                      // (rows of s_dz beyond the last are zero)
        float a = rb ? G[L.w + (int64_t)c * d + j] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) a += s_dz[min(rb + u, kHeadMaxRows - 1) * kHeadMaxC + c] * hh[u];
        G[L.w + (int64_t)c * d + j] = a;
      }
    }
    G[L.g1 + j] = has1 ? dg1 : 0.f; G[L.b1 + j] = has1 ? db1 : 0.f; G[L.g2 + j] = dg2; G[L.b2 + j] = db2;
  }
  if ((int)threadIdx.x < C) {
    float a = 0.f;
    for (int r = 0; r < p.rows; ++r) a += s_dz[r * kHeadMaxC + threadIdx.x];
    G[L.c + threadIdx.x] = a;
  }
}

template <typename T>
int launch_head(const dvt_head_bce_desc& p, int threads, hipStream_t st) {
  switch (p.d / 64) {
#define DVT_HEAD_NV(NVV) \
    case NVV: hipLaunchKernelGGL((head_bce_kernel<T, NVV>), dim3(1), dim3(threads), 0, st, p); return 0;
    DVT_HEAD_NV(1) DVT_HEAD_NV(2) DVT_HEAD_NV(3) DVT_HEAD_NV(4) DVT_HEAD_NV(6) DVT_HEAD_NV(8) DVT_HEAD_NV(12) DVT_HEAD_NV(16)
#undef DVT_HEAD_NV
  }
  return 1;
}

constexpr int kEmitGroup = 16;
constexpr int kEmitPerBlock = 2048;
struct EmitGroup { dvt_emit_entry e[kEmitGroup]; int begin[kEmitGroup + 1]; int n; };

template <typename D>
__device__ __forceinline__ void emit_elems(const dvt_emit_entry& q, int64_t i0, float sc) {
  const int64_t end = min(q.n, i0 + kEmitPerBlock);
  for (int64_t i = i0 + threadIdx.x; i < end; i += 256) {
    float v = sc * q.src[i];
    if (q.dst) {
      if (q.accumulate) v += q.dst[i];
      q.dst[i] = v;
    }
    if (q.dst_lp) ((D*)q.dst_lp)[i] = from_f32<D>(v);
  }
}

__global__ __launch_bounds__(256) void scaled_emit_group_kernel(const float* __restrict__ scale, const EmitGroup g) {
  int e = 0;
  while (e + 1 < g.n && (int)blockIdx.x >= g.begin[e + 1]) ++e;
  const dvt_emit_entry& q = g.e[e];
  const int64_t i0 = (int64_t)((int)blockIdx.x - g.begin[e]) * kEmitPerBlock;
  const float sc = scale[0];
  if (q.lp_dtype == DVT_F16) emit_elems<f16>(q, i0, sc);
  else emit_elems<bf16>(q, i0, sc);
}

}  // namespace

extern "C" {

int dvt_head_bce_supported(int rows, int d, int classes) {
  if (!(rows > 0 && rows <= kHeadMaxRows && d > 0 && d % 64 == 0 && classes > 0 && classes <= kHeadMaxC &&
        classes * d <= kHeadWLds))
    return 0;
  const int nv = d / 64;
  return nv == 1 || nv == 2 || nv == 3 || nv == 4 || nv == 6 || nv == 8 || nv == 12 || nv == 16;
}

int64_t dvt_head_bce_grads_elems(int rows, int d, int classes) {
  if (!dvt_head_bce_supported(rows, d, classes)) return -1;
  return head_layout(rows, d, classes).total;
}

int dvt_head_bce_fwd(const dvt_head_bce_desc* p, dvt_stream_t stream) {
  DVT_REQUIRE(p && p->x && p->g2 && p->b2 && p->w && p->target && p->logits && p->loss && p->grads,
              "dvt_head_bce_fwd: null pointer");
  DVT_REQUIRE((p->g1 == nullptr) == (p->b1 == nullptr), "dvt_head_bce_fwd: g1 and b1 come together");
  DVT_REQUIRE(dvt_head_bce_supported(p->rows, p->d, p->classes),
              "dvt_head_bce_fwd: rows <= %d, d / 64 in {1,2,3,4,6,8,12,16}, classes <= %d, classes * d <= %d (got %d, %d, %d)",
              kHeadMaxRows, kHeadMaxC, kHeadWLds, p->rows, p->d, p->classes);
  const int waves = p->rows < 8 ? p->rows : 8;                         // a wave per row (eight at a time)
  const int threads = p->d >= 512 ? 512 : (waves * 64 > p->d ? waves * 64 : p->d);   // and column owners behind (d % 64 == 0)
  hipStream_t st = (hipStream_t)stream;
  int miss = 1;
  DVT_DISPATCH_DTYPE(p->x_dtype, T, miss = launch_head<T>(*p, threads, st));
  DVT_REQUIRE(miss == 0, "dvt_head_bce_fwd: no instantiation for d = %d", p->d);
  DVT_LAUNCH_CHECK("dvt_head_bce_fwd");
  return DVT_OK;
}

int dvt_scaled_emit_group(const float* scale, const dvt_emit_entry* entries, int count, dvt_stream_t stream) {
  DVT_REQUIRE(scale && count >= 0 && (count == 0 || entries), "dvt_scaled_emit_group: bad arguments");
  for (int base = 0; base < count; base += kEmitGroup) {
    EmitGroup g{};
    int blocks = 0;
    g.n = count - base < kEmitGroup ? count - base : kEmitGroup;
    for (int i = 0; i < g.n; ++i) {
      const dvt_emit_entry& q = entries[base + i];
      DVT_REQUIRE(q.src && (q.dst || q.dst_lp) && q.n > 0 && (!q.dst_lp || dvt_is_16bit(q.lp_dtype)),
                  "dvt_scaled_emit_group: bad entry %d", base + i);
      g.e[i] = q;
      g.begin[i] = blocks;
      blocks += (int)dvt_cdiv(q.n, kEmitPerBlock);
    }
    g.begin[g.n] = blocks;
    hipLaunchKernelGGL(scaled_emit_group_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, scale, g);
    DVT_LAUNCH_CHECK("dvt_scaled_emit_group");
  }
  return DVT_OK;
}

}  // extern "C"
