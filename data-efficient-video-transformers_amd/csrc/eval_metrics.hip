// eval_metrics.hip -- evaluation reductions on the device.
//
// Replaces the host-side scikit-learn calls of src/callbacks/callbacks.py:36-55 on the accumulated
// running_logits / running_labels: samples-averaged F1 at a sweep of thresholds, and average precision
// ("samples" and support-"weighted").  Not on the training hot path: clarity over speed, deterministic order.
#include <cstring>

#include "common.h"

#include <rocprim/device/device_segmented_radix_sort.hpp>

namespace {

constexpr int kMaxRowClasses = 64;   // classes per sample for the per-row AP (19 / 15 in the reference)
constexpr int kMaxThresholds = 16;

struct Thresholds { float t[kMaxThresholds]; };

// per-row F1 at every threshold: f[n, j] = 2 |P & T| / (|P| + |T|), 0 if both empty
__global__ void f1_rows_kernel(const float* __restrict__ probs, const unsigned char* __restrict__ labels, int64_t N,
                               int C, Thresholds th, int T, float* __restrict__ f) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  for (int j = 0; j < T; ++j) {
    int tp = 0, den = 0;
    for (int c = 0; c < C; ++c) {
      const int p = probs[n * C + c] > th.t[j], l = labels[n * C + c] != 0;
      tp += p & l;
      den += p + l;
    }
    f[n * T + j] = den > 0 ? (float)(2.0 * tp / den) : 0.f;
  }
}

// column means of a [N, T] f32 matrix, two stages, fixed order
__global__ void colmean_partial_kernel(const float* __restrict__ x, int64_t N, int T, int64_t rows_per_block,
                                       double* __restrict__ part) {
  const int j = threadIdx.x;
  if (j >= T) return;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
  double s = 0.0;
  for (int64_t r = r0; r < r1; ++r) s += (double)x[r * T + j];
  part[(int64_t)blockIdx.x * T + j] = s;
}

__global__ void colmean_final_kernel(const double* __restrict__ part, int nblocks, int T, double inv, float* __restrict__ out) {
  const int j = threadIdx.x;
  if (j >= T) return;
  double s = 0.0;
  for (int b = 0; b < nblocks; ++b) s += part[(int64_t)b * T + j];
  out[j] = (float)(s * inv);
}

// AP of one sample over its C class scores (insertion sort, descending), written to ap_rows[n]
__global__ void ap_rows_kernel(const float* __restrict__ probs, const unsigned char* __restrict__ labels, int64_t N,
                               int C, float* __restrict__ ap_rows) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s[kMaxRowClasses];
  unsigned char l[kMaxRowClasses];
  int total = 0;
  for (int c = 0; c < C; ++c) {
    const float v = probs[n * C + c];
    const unsigned char y = labels[n * C + c] != 0;
    total += y;
    int i = c;
    while (i > 0 && s[i - 1] < v) { s[i] = s[i - 1]; l[i] = l[i - 1]; --i; }
    s[i] = v; l[i] = y;
  }
  double ap = 0.0, r_prev = 0.0;
  int tp = 0;
  if (total > 0) {
    for (int i = 0; i < C; ++i) {
      tp += l[i];
      if (i == C - 1 || s[i + 1] != s[i]) {
        const double r = (double)tp / total;
        ap += (r - r_prev) * ((double)tp / (i + 1));
        r_prev = r;
      }
    }
  }
  ap_rows[n] = (float)ap;
}

// [N, C] -> class-major keys [C, N] (f32) and labels [C, N] (u8)
__global__ void to_class_major_kernel(const float* __restrict__ probs, const unsigned char* __restrict__ labels,
                                      int64_t N, int C, float* __restrict__ keys, unsigned char* __restrict__ vals) {
  const int64_t total = N * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / C;
    const int c = (int)(i % C);
    keys[(int64_t)c * N + n] = probs[i];
    vals[(int64_t)c * N + n] = labels[i] != 0;
  }
}

__global__ void offsets_kernel(int* __restrict__ off, int C, int64_t N) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c <= C) off[c] = (int)(c * N);
}

// one thread per class over its descending-sorted scores; also the support-weighted and the samples average
__global__ void ap_class_kernel(const float* __restrict__ keys, const unsigned char* __restrict__ vals, int64_t N, int C,
                                float* __restrict__ ap_class, int* __restrict__ support) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float* s = keys + (int64_t)c * N;
  const unsigned char* l = vals + (int64_t)c * N;
  int64_t total = 0;
  for (int64_t i = 0; i < N; ++i) total += l[i];
  double ap = 0.0, r_prev = 0.0;
  int64_t tp = 0;
  if (total > 0) {
    for (int64_t i = 0; i < N; ++i) {
      tp += l[i];
      if (i == N - 1 || s[i + 1] != s[i]) {
        const double r = (double)tp / (double)total;
        ap += (r - r_prev) * ((double)tp / (double)(i + 1));
        r_prev = r;
      }
    }
  }
  ap_class[c] = (float)ap;
  support[c] = (int)total;
}

__global__ void ap_weighted_kernel(const float* __restrict__ ap_class, const int* __restrict__ support, int C,
                                   float* __restrict__ out_weighted) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double num = 0.0, den = 0.0;
  for (int c = 0; c < C; ++c) { num += (double)ap_class[c] * support[c]; den += support[c]; }
  out_weighted[0] = den > 0 ? (float)(num / den) : 0.f;
}

inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

struct ApPlan {
  size_t off_keys_in, off_keys_out, off_vals_in, off_vals_out, off_offsets, off_ap_rows, off_part, off_support, off_temp,
      temp_bytes, bytes;
  int nblocks;
  int64_t rpb;
};

bool ap_plan(int64_t N, int C, ApPlan* p) {
  if (N <= 0 || C <= 0 || N * C >= (int64_t)1 << 31) return false;
  size_t temp = 0;
  if (rocprim::segmented_radix_sort_pairs_desc(nullptr, temp, (const float*)nullptr, (float*)nullptr,
                                               (const unsigned char*)nullptr, (unsigned char*)nullptr, (unsigned)(N * C),
                                               (unsigned)C, (const int*)nullptr, (const int*)nullptr, 0, 32,
                                               (hipStream_t)0) != hipSuccess)
    return false;
  p->temp_bytes = temp;
  p->nblocks = (int)(N < 4096 ? 1 : (N / 4096 < 1024 ? N / 4096 : 1024));
  p->rpb = dvt_cdiv(N, p->nblocks);
  p->nblocks = (int)dvt_cdiv(N, p->rpb);
  size_t o = 0;
  p->off_keys_in = o;  o = al256(o + sizeof(float) * N * C);
  p->off_keys_out = o; o = al256(o + sizeof(float) * N * C);
  p->off_vals_in = o;  o = al256(o + (size_t)N * C);
  p->off_vals_out = o; o = al256(o + (size_t)N * C);
  p->off_offsets = o;  o = al256(o + sizeof(int) * (C + 1));
  p->off_ap_rows = o;  o = al256(o + sizeof(float) * N);
  p->off_part = o;     o = al256(o + sizeof(double) * (size_t)p->nblocks * kMaxThresholds);
  p->off_support = o;  o = al256(o + sizeof(int) * C);
  p->off_temp = o;     o = al256(o + temp);
  p->bytes = o;
  return true;
}

}  // namespace

extern "C" {

size_t dvt_f1_samples_workspace_bytes(int64_t N, int T) {
  if (N <= 0 || T <= 0 || T > kMaxThresholds) return 0;
  return al256(sizeof(float) * (size_t)N * T) + al256(sizeof(double) * 1024 * kMaxThresholds);
}

int dvt_f1_samples(const float* probs, const unsigned char* labels, int64_t N, int C, const float* thresholds, int T,
                   float* out, void* workspace, dvt_stream_t stream) {
  DVT_REQUIRE(probs && labels && thresholds && out && workspace && N > 0 && C > 0, "dvt_f1_samples: bad arguments");
  DVT_REQUIRE(T > 0 && T <= kMaxThresholds, "dvt_f1_samples: 1..%d thresholds", kMaxThresholds);
  hipStream_t st = (hipStream_t)stream;
  Thresholds th;
  for (int j = 0; j < kMaxThresholds; ++j) th.t[j] = j < T ? thresholds[j] : 0.f;
  float* f = (float*)workspace;
  double* part = (double*)((char*)workspace + al256(sizeof(float) * (size_t)N * T));
  hipLaunchKernelGGL(f1_rows_kernel, dim3((unsigned)dvt_cdiv(N, 256)), dim3(256), 0, st, probs, labels, N, C, th, T, f);
  int nblocks = (int)(N < 4096 ? 1 : (N / 4096 < 1024 ? N / 4096 : 1024));
  const int64_t rpb = dvt_cdiv(N, nblocks);
  nblocks = (int)dvt_cdiv(N, rpb);
  hipLaunchKernelGGL(colmean_partial_kernel, dim3(nblocks), dim3(kMaxThresholds), 0, st, (const float*)f, N, T, rpb, part);
  hipLaunchKernelGGL(colmean_final_kernel, dim3(1), dim3(kMaxThresholds), 0, st, (const double*)part, nblocks, T,
                     1.0 / (double)N, out);
  DVT_LAUNCH_CHECK("dvt_f1_samples");
  return DVT_OK;
}

size_t dvt_average_precision_workspace_bytes(int64_t N, int C) {
  ApPlan p;
  return ap_plan(N, C, &p) ? p.bytes : 0;
}

int dvt_average_precision(const float* probs, const unsigned char* labels, int64_t N, int C, float* out_samples,
                          float* out_weighted, float* out_per_class, void* workspace, dvt_stream_t stream) {
  DVT_REQUIRE(probs && labels && out_samples && out_weighted && out_per_class && workspace && N > 0 && C > 0,
              "dvt_average_precision: bad arguments");
  DVT_REQUIRE(C <= kMaxRowClasses, "dvt_average_precision: at most %d classes", kMaxRowClasses);
  ApPlan p;
  DVT_REQUIRE(ap_plan(N, C, &p), "dvt_average_precision: N*C too large");
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  float* keys_in = (float*)(ws + p.off_keys_in);
  float* keys_out = (float*)(ws + p.off_keys_out);
  unsigned char* vals_in = (unsigned char*)(ws + p.off_vals_in);
  unsigned char* vals_out = (unsigned char*)(ws + p.off_vals_out);
  int* offsets = (int*)(ws + p.off_offsets);
  float* ap_rows = (float*)(ws + p.off_ap_rows);
  double* part = (double*)(ws + p.off_part);
  int* support = (int*)(ws + p.off_support);
  // "samples": per-row AP, then the mean over rows
  hipLaunchKernelGGL(ap_rows_kernel, dim3((unsigned)dvt_cdiv(N, 128)), dim3(128), 0, st, probs, labels, N, C, ap_rows);
  hipLaunchKernelGGL(colmean_partial_kernel, dim3(p.nblocks), dim3(kMaxThresholds), 0, st, (const float*)ap_rows, N, 1,
                     p.rpb, part);
  hipLaunchKernelGGL(colmean_final_kernel, dim3(1), dim3(kMaxThresholds), 0, st, (const double*)part, p.nblocks, 1,
                     1.0 / (double)N, out_samples);
  DVT_LAUNCH_CHECK("dvt_average_precision(samples)");
  // "weighted": per-class sort (rocPRIM segmented radix sort, descending, stable) + one scan per class
  hipLaunchKernelGGL(to_class_major_kernel, dim3((unsigned)(dvt_cdiv(N * C, 256) < 4096 ? dvt_cdiv(N * C, 256) : 4096)),
                     dim3(256), 0, st, probs, labels, N, C, keys_in, vals_in);
  hipLaunchKernelGGL(offsets_kernel, dim3((unsigned)dvt_cdiv(C + 1, 64)), dim3(64), 0, st, offsets, C, N);
  DVT_LAUNCH_CHECK("dvt_average_precision(layout)");
  size_t temp = p.temp_bytes;
  const hipError_t e = rocprim::segmented_radix_sort_pairs_desc(
      (void*)(ws + p.off_temp), temp, (const float*)keys_in, keys_out, (const unsigned char*)vals_in, vals_out,
      (unsigned)(N * C), (unsigned)C, (const int*)offsets, (const int*)(offsets + 1), 0, 32, st);
  DVT_REQUIRE(e == hipSuccess, "dvt_average_precision: segmented sort failed: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(ap_class_kernel, dim3((unsigned)dvt_cdiv(C, 64)), dim3(64), 0, st, (const float*)keys_out,
                     (const unsigned char*)vals_out, N, C, out_per_class, support);
  hipLaunchKernelGGL(ap_weighted_kernel, dim3(1), dim3(64), 0, st, (const float*)out_per_class, (const int*)support, C,
                     out_weighted);
  DVT_LAUNCH_CHECK("dvt_average_precision(classes)");
  return DVT_OK;
}

}  // extern "C"
