// input_stage.hip -- decoded uint8 RGB frames -> normalised clip tensor, on the device.
//
// Replaces the per-frame host pipeline of the reference's loader
//   transforms.Compose([Resize(S), CenterCrop(C), ToTensor(), Normalize(mean, std)])
// (src/dataloaders/mmx/MMX_Light_dl.py:203-217, applied frame by frame by 10 PIL workers at :161-162,270-273).
// Bit-exact with Pillow's 8-bit bilinear resample (ImagingResample): triangle filter of support max(scale, 1),
// weights normalised in double precision and rounded to 22-bit fixed point, horizontal pass then vertical pass
// through a uint8 intermediate.  Double-precision expressions must round exactly like the host C code, so
// floating-point contraction is disabled for this file.
#pragma clang fp contract(off)
#include "common.h"

namespace {

constexpr int kPrec = 32 - 8 - 2;   // Pillow PRECISION_BITS
constexpr int kB = 256;

__device__ __forceinline__ int clip8(int v) {
  v >>= kPrec;                      // arithmetic shift, then the clip8 lookup of Pillow
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// One thread per output coordinate: Pillow precompute_coeffs + normalize_coeffs_8bpc (bilinear, support 1).
__global__ void resample_coeff_kernel(int in_size, int out_size, int ksize, int* __restrict__ xmin,
                                      int* __restrict__ cnt, int* __restrict__ kk) {
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= out_size) return;
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const double ss = 1.0 / filterscale;
  const double center = ((double)xx + 0.5) * scale;
  int lo = (int)(center - support + 0.5);
  if (lo < 0) lo = 0;
  int hi = (int)(center + support + 0.5);
  if (hi > in_size) hi = in_size;
  const int n = hi - lo;
  double ww = 0.0;
  for (int x = 0; x < n; ++x) {
    double a = ((double)(x + lo) - center + 0.5) * ss;
    if (a < 0.0) a = -a;
    ww += a < 1.0 ? 1.0 - a : 0.0;
  }
  int* k = kk + (int64_t)xx * ksize;
  for (int x = 0; x < ksize; ++x) {
    int q = 0;
    if (x < n) {
      double a = ((double)(x + lo) - center + 0.5) * ss;
      if (a < 0.0) a = -a;
      double w = a < 1.0 ? 1.0 - a : 0.0;
      if (ww != 0.0) w = w / ww;
      q = w < 0.0 ? (int)(-0.5 + w * (double)(1 << kPrec)) : (int)(0.5 + w * (double)(1 << kPrec));
    }
    k[x] = q;
  }
  xmin[xx] = lo;
  cnt[xx] = n;
}

// Horizontal pass: src [F, H0, W0, 3] -> tmp [F, H0, Wc, 3] for the Wc resized columns [left, left + Wc).
__global__ void resample_h_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ tmp,
                                  const int* __restrict__ xmin, const int* __restrict__ cnt,
                                  const int* __restrict__ kk, int ksize, int64_t rows /* F*H0 */, int W0, int Wc,
                                  int left) {
  const int64_t total = rows * Wc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xo = (int)(i % Wc);
    const int64_t r = i / Wc;
    const int xx = xo + left;
    const int lo = xmin[xx], n = cnt[xx];
    const int* k = kk + (int64_t)xx * ksize;
    const unsigned char* p = src + (r * W0 + lo) * 3;
    int s0 = 1 << (kPrec - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < n; ++x) {
      const int c = k[x];
      s0 += (int)p[3 * x + 0] * c;
      s1 += (int)p[3 * x + 1] * c;
      s2 += (int)p[3 * x + 2] * c;
    }
    unsigned char* o = tmp + i * 3;
    o[0] = (unsigned char)clip8(s0); o[1] = (unsigned char)clip8(s1); o[2] = (unsigned char)clip8(s2);
  }
}

// Vertical pass over tmp + centre crop + ToTensor (/255) + Normalize ((t - mean) / std), written as NCHW.
template <typename D>
__global__ void resample_v_norm_kernel(const unsigned char* __restrict__ tmp, D* __restrict__ dst,
                                       const int* __restrict__ ymin, const int* __restrict__ cnt,
                                       const int* __restrict__ kk, int ksize, int64_t F, int H0, int Wc, int Hc,
                                       int top, float m0, float m1, float m2, float d0, float d1, float d2) {
  const int64_t total = F * Hc * Wc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xo = (int)(i % Wc), yo = (int)((i / Wc) % Hc);
    const int64_t f = i / ((int64_t)Wc * Hc);
    const int yy = yo + top;
    const int lo = ymin[yy], n = cnt[yy];
    const int* k = kk + (int64_t)yy * ksize;
    const unsigned char* p = tmp + ((f * H0 + lo) * Wc + xo) * 3;
    int s0 = 1 << (kPrec - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < n; ++y) {
      const int c = k[y];
      const unsigned char* q = p + (int64_t)y * Wc * 3;
      s0 += (int)q[0] * c; s1 += (int)q[1] * c; s2 += (int)q[2] * c;
    }
    const float v0 = __fdiv_rn(__fdiv_rn((float)clip8(s0), 255.0f) - m0, d0);
    const float v1 = __fdiv_rn(__fdiv_rn((float)clip8(s1), 255.0f) - m1, d1);
    const float v2 = __fdiv_rn(__fdiv_rn((float)clip8(s2), 255.0f) - m2, d2);
    const int64_t plane = (int64_t)Hc * Wc;
    D* o = dst + f * 3 * plane + (int64_t)yo * Wc + xo;
    o[0] = from_f32<D>(v0); o[plane] = from_f32<D>(v1); o[2 * plane] = from_f32<D>(v2);
  }
}

struct Plan {
  int h, w, top, left, ks_h, ks_w;
  size_t off_xmin, off_xcnt, off_xk, off_ymin, off_ycnt, off_yk, off_tmp, bytes;
};

int ksize_of(int in_size, int out_size) {
  const double scale = (double)in_size / (double)out_size;
  const double support = scale < 1.0 ? 1.0 : scale;
  return (int)ceil(support) * 2 + 1;
}

bool make_plan(int64_t frames, int H0, int W0, int resize, int crop, Plan* p) {
  // torchvision Resize(int): shorter side -> resize, longer -> int(resize * long / short)
  if (W0 <= H0) { p->w = resize; p->h = (int)((double)resize * H0 / W0); }
  else { p->h = resize; p->w = (int)((double)resize * W0 / H0); }
  if (p->h < crop || p->w < crop) return false;
  p->top = (int)lrint((p->h - crop) / 2.0);      // Python round(): half to even, like lrint in the default mode
  p->left = (int)lrint((p->w - crop) / 2.0);
  p->ks_w = ksize_of(W0, p->w);
  p->ks_h = ksize_of(H0, p->h);
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t o = 0;
  p->off_xmin = o; o = al(o + sizeof(int) * p->w);
  p->off_xcnt = o; o = al(o + sizeof(int) * p->w);
  p->off_xk = o;   o = al(o + sizeof(int) * (size_t)p->w * p->ks_w);
  p->off_ymin = o; o = al(o + sizeof(int) * p->h);
  p->off_ycnt = o; o = al(o + sizeof(int) * p->h);
  p->off_yk = o;   o = al(o + sizeof(int) * (size_t)p->h * p->ks_h);
  p->off_tmp = o;  o = al(o + (size_t)frames * H0 * crop * 3);
  p->bytes = o;
  return true;
}

inline int grid_for(int64_t items) {
  int64_t b = dvt_cdiv(items, kB);
  const int64_t cap = (int64_t)dvt_num_cus() * 16;
  if (b > cap) b = cap;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" {

size_t dvt_frames_preprocess_workspace_bytes(int64_t frames, int H0, int W0, int resize, int crop) {
  Plan p;
  if (frames < 0 || H0 <= 0 || W0 <= 0 || resize <= 0 || crop <= 0 || !make_plan(frames, H0, W0, resize, crop, &p))
    return 0;
  return p.bytes;
}

int dvt_frames_preprocess(const void* src, void* dst, int dst_dtype, int64_t frames, int H0, int W0, int resize,
                          int crop, const float* mean, const float* std, void* workspace, dvt_stream_t stream) {
  DVT_REQUIRE(src && dst && mean && std && workspace && frames >= 0 && H0 > 0 && W0 > 0 && resize > 0 && crop > 0,
              "dvt_frames_preprocess: bad arguments");
  Plan p;
  DVT_REQUIRE(make_plan(frames, H0, W0, resize, crop, &p),
              "dvt_frames_preprocess: crop %d exceeds the resized frame (%d x %d -> shorter side %d)", crop, H0, W0, resize);
  DVT_REQUIRE(std[0] != 0.f && std[1] != 0.f && std[2] != 0.f, "dvt_frames_preprocess: zero std");
  if (frames == 0) return DVT_OK;
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  int* xmin = (int*)(ws + p.off_xmin); int* xcnt = (int*)(ws + p.off_xcnt); int* xk = (int*)(ws + p.off_xk);
  int* ymin = (int*)(ws + p.off_ymin); int* ycnt = (int*)(ws + p.off_ycnt); int* yk = (int*)(ws + p.off_yk);
  unsigned char* tmp = (unsigned char*)(ws + p.off_tmp);
  hipLaunchKernelGGL(resample_coeff_kernel, dim3((unsigned)dvt_cdiv(p.w, 64)), dim3(64), 0, st, W0, p.w, p.ks_w, xmin,
                     xcnt, xk);
  hipLaunchKernelGGL(resample_coeff_kernel, dim3((unsigned)dvt_cdiv(p.h, 64)), dim3(64), 0, st, H0, p.h, p.ks_h, ymin,
                     ycnt, yk);
  DVT_LAUNCH_CHECK("dvt_frames_preprocess(coefficients)");
  hipLaunchKernelGGL(resample_h_kernel, dim3(grid_for(frames * H0 * crop)), dim3(kB), 0, st, (const unsigned char*)src,
                     tmp, xmin, xcnt, xk, p.ks_w, frames * H0, W0, crop, p.left);
  DVT_LAUNCH_CHECK("dvt_frames_preprocess(horizontal)");
  DVT_DISPATCH_DTYPE(dst_dtype, D, hipLaunchKernelGGL((resample_v_norm_kernel<D>), dim3(grid_for(frames * crop * crop)),
                                                      dim3(kB), 0, st, tmp, (D*)dst, ymin, ycnt, yk, p.ks_h, frames, H0,
                                                      crop, crop, p.top, mean[0], mean[1], mean[2], std[0], std[1],
                                                      std[2]));
  DVT_LAUNCH_CHECK("dvt_frames_preprocess(vertical)");
  return DVT_OK;
}

}  // extern "C"
