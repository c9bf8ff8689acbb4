// common.h -- shared device/host helpers for libdvt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/dvt_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define DVT_WAVE 64

// ---------------------------------------------------------------- host: errors
int dvt_fail(int code, const char* fmt, ...);
int dvt_fail_hip(hipError_t e, const char* where);

#define DVT_REQUIRE(cond, ...)                                \
  do {                                                        \
    if (!(cond)) return dvt_fail(DVT_ERR_BAD_ARG, __VA_ARGS__); \
  } while (0)

#define DVT_UNSUPPORTED(...) return dvt_fail(DVT_ERR_UNSUPPORTED, __VA_ARGS__)

#define DVT_LAUNCH_CHECK(name)                          \
  do {                                                  \
    hipError_t e__ = hipGetLastError();                 \
    if (e__ != hipSuccess) return dvt_fail_hip(e__, name); \
  } while (0)

static inline bool dvt_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline size_t dvt_dtype_size(int dt) { return dt == DVT_F32 ? 4 : 2; }
static inline bool dvt_is_16bit(int dt) { return dt == DVT_BF16 || dt == DVT_F16; }
static inline int64_t dvt_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
int dvt_num_cus();

// hipFuncAttributeMaxDynamicSharedMemorySize is an attribute of a kernel PER DEVICE: one flag per (call site, device),
// atomic (two host threads racing on a first call both set it: idempotent), the return code recorded -- the launch behind a
// failed call fails as well and DVT_LAUNCH_CHECK reports it.
// conv.hip: reduction of BatchNorm-backward partial rows produced by another translation unit's kernel
namespace dvt_internal {
// conv3x1_c64.hip: the 64 -> 64 form of the (3, 1, 1) window convolution, behind dvt_conv3x1_fwd (conv3x1_fwd.hip)
int conv3x1_c64_supported(int64_t N, int T, int L, int dtype);
int64_t conv3x1_c64_stats_parts(int64_t N, int T, int L);
int conv3x1_c64_fwd(const void* x, const void* w, int64_t ldw, void* y, float* stats_partial, int64_t N, int T, int L, int dtype,
                    hipStream_t st);
void bn_bwd_finalize(hipStream_t st, const float* partial, int nparts, int C, float* loc, int accumulate, float* dgamma,
                     float* dbeta, int c_valid);
// conv3x1_dbn.hip: the temporal data gradient 64 -> 144 + mid-plane BatchNorm backward as a window kernel with helper waves
// (behind dvt_conv3x1_stream_bn_bwd, conv3x3_stream.hip)
int conv3x1_dbn_supported(int64_t N, int T, int L, int dtype);
int conv3x1_dbn_parts(int64_t N, int T, int L);
int conv3x1_dbn_pass(int mode, const void* dy, const void* w, int64_t ldw, const void* z, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, int relu, int training, float* partial, const float* loc, void* dz,
                     int64_t N, int T, int L, int dtype, hipStream_t st);
}

struct DvtLdsAttr { unsigned long long done = 0ull; };
static inline void dvt_lds_attr(DvtLdsAttr& f, const void* kernel, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
  if (dev >= 0 && ((__atomic_load_n(&f.done, __ATOMIC_ACQUIRE) >> dev) & 1ull)) return;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) { (void)dvt_fail_hip(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)"); return; }
  if (dev >= 0) __atomic_fetch_or(&f.done, 1ull << dev, __ATOMIC_RELEASE);
}

// ---------------------------------------------------------------- device: scalar conversions
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 v) { return (float)v; }
template <> __device__ __forceinline__ float to_f32<f16>(f16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ f16 from_f32<f16>(float v) { return (f16)v; }
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// 8 contiguous elements <-> 8 floats (16-byte accesses for bf16, 2x16 for f32).
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&o)[8]);
template <> __device__ __forceinline__ void load8<bf16>(const bf16* p, float (&o)[8]) {
  bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
}
template <> __device__ __forceinline__ void load8<f16>(const f16* p, float (&o)[8]) {
  f16x8 v = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&o)[8]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { o[i] = a[i]; o[4 + i] = b[i]; }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&o)[8]);
template <> __device__ __forceinline__ void store8<bf16>(bf16* p, const float (&o)[8]) {
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (bf16)o[i];
  *reinterpret_cast<bf16x8*>(p) = v;
}
template <> __device__ __forceinline__ void store8<f16>(f16* p, const float (&o)[8]) {
  f16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (f16)o[i];
  *reinterpret_cast<f16x8*>(p) = v;
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&o)[8]) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = o[i]; b[i] = o[4 + i]; }
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}

// streaming (non-temporal) 16-byte store of 8 converted values: written once, read by a later kernel -- keeps the GEMM's
// output stream from evicting the operand panels its neighbours on the XCD are re-reading from L2
template <typename T> __device__ __forceinline__ void store8_nt(T* p, const float (&o)[8]) {
  typedef T v8t __attribute__((ext_vector_type(8)));
  typedef int i4t __attribute__((ext_vector_type(4)));
  v8t v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (T)o[i];
  // from asm: beside an ordinary store of the same value under a runtime flag the compiler merges the two and drops `nt`.
  // The s_nop is part of the instruction's contract: a store of more than 8 bytes reads its data registers over the wait
  // states that follow, and the hazard recogniser cannot see into the asm -- a VALU write that recycles those registers in
  // the very next slot corrupted the rows of the last lanes (tools/dev/gemm_astat.hip: rows 29..31 of a 32-row block).
  asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(__builtin_bit_cast(i4t, v)) : "memory");
}

// ---------------------------------------------------------------- device: 16-bit MFMA element traits
// bf16 and fp16 share every data path (LDS images, DMA, swizzles); only the matrix instruction, the
// transposing LDS read and the conversions differ.
template <typename E> struct Elem16;
template <> struct Elem16<bf16> {
  typedef bf16x8 v8;
  typedef bf16x4 v4;
  static __device__ __forceinline__ f32x4 mma(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ v4 tr_read(const char* lds) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) v4*)(lds));
  }
};
template <> struct Elem16<f16> {
  typedef f16x8 v8;
  typedef f16x4 v4;
  static __device__ __forceinline__ f32x4 mma(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ v4 tr_read(const char* lds) {
    typedef __attribute__((ext_vector_type(4))) short i16x4;    // same instruction, 16-bit payload reinterpreted
    const i16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(lds));
    return __builtin_bit_cast(v4, r);
  }
};

// ---------------------------------------------------------------- device: wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Wave64 sum for wave-uniform control flow (ALL 64 lanes active), without LDS round trips: DPP inside a row of 16 lanes,
// v_permlane{16,32}_swap between the rows (a ds_bpermute chain costs ~100 cycles per step of the butterfly).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140, kDppRor8 = 0x128;

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_mov<kDppXor1>(v);
  v += dpp_mov<kDppXor2>(v);
  v += dpp_mov<kDppHalfMirror>(v);
  v += dpp_mov<kDppMirror>(v);                       // every row of 16 lanes holds its own total
  float a = v, b = v;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  v = a + b;
  a = v;
  b = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
#pragma clang diagnostic pop

// ---------------------------------------------------------------- device: activations
// Exact-erf GELU evaluated with the Abramowitz-Stegun 7.1.26 rational form
// (|erf error| <= 1.5e-7 absolute), sharing one exponential between the cdf and
// the pdf: exp(-x^2/2) is both the tail factor of erf(x/sqrt2) and the Gaussian
// density.  ~14 VALU ops instead of erff+expf (~45).  The GEMM epilogues that fuse GELU / GELU' pay for them all the same
// (2 x 103 M elements per feed-forward layer at the metric shape, ~35 us of a 163 us launch), so the form is kept short:
// the argument scale 1/sqrt2 is folded into the constant of t, and the sign goes on with a bit-field insert
// (0.5 + copysign(h, x) is 0.5 - h for negative x bit for bit) instead of a compare and two selects.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
  const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.23164189f, 1.0f));        // 1 / (1 + 0.3275911 |x| / sqrt2)
  const float E = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);  // exp(-x^2/2)
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  const float half_erf = 0.5f - 0.5f * (p * t) * E;  // 0.5 * erf(|x|/sqrt2)
  cdf = 0.5f + __builtin_copysignf(half_erf, x);
  pdf = 0.39894228040143267794f * E;
}
__device__ __forceinline__ float gelu_erf_f(float x) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  return x * cdf;
}
// d/dx [ x Phi(x) ] = Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_erf_grad_f(float x) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  return fmaf(x, pdf, cdf);
}
// value and derivative from the same parts: what a GEMM's GELU epilogue stores as C and as aux (the backward epilogue then
// only multiplies -- re-evaluating the derivative from a stored pre-activation cost the backward launch what the GELU costs
// the forward one, ~35 us at the metric shape, for two more vector instructions here)
__device__ __forceinline__ float gelu_erf_both_f(float x, float& grad) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  grad = fmaf(x, pdf, cdf);
  return x * cdf;
}

// ---------------------------------------------------------------- device: counter-based RNG (dropout)
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// keep-decision of dropout element `idx` under (seed, base): word (idx & 3) of Philox block base + idx / 4
__device__ __forceinline__ bool dvt_dropout_keep(uint64_t seed, uint64_t base, uint64_t idx, uint32_t threshold) {
  const uint64_t ctr = base + (idx >> 2);
  uint32_t r[4];
  philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
  return r[idx & 3] >= threshold;
}

// Dispatch a templated launcher on the activation dtype.
#define DVT_DISPATCH_DTYPE(dtype, T, ...)                                   \
  do {                                                                      \
    if ((dtype) == DVT_F32) { typedef float T; __VA_ARGS__; }               \
    else if ((dtype) == DVT_BF16) { typedef bf16 T; __VA_ARGS__; }          \
    else if ((dtype) == DVT_F16) { typedef f16 T; __VA_ARGS__; }            \
    else DVT_UNSUPPORTED("dtype %d not supported by this kernel", (int)(dtype)); \
  } while (0)

// Same for kernels that exist for the two 16-bit element types only (MFMA paths).
#define DVT_DISPATCH_16BIT(dtype, E, ...)                                   \
  do {                                                                      \
    if ((dtype) == DVT_BF16) { typedef bf16 E; __VA_ARGS__; }               \
    else if ((dtype) == DVT_F16) { typedef f16 E; __VA_ARGS__; }            \
    else DVT_UNSUPPORTED("dtype %d is not a 16-bit MFMA element type", (int)(dtype)); \
  } while (0)

// LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane, LDS destination = wave-uniform base + 16 * lane), issued from inline
// asm so that the compiler's wait-count pass does not see it: that pass models an LDS-DMA as a pending LDS write and, with
// no alias scopes on dynamic LDS, waits vmcnt(0) before the next ds_read -- i.e. right after the issue, serialising the
// very transfer the pipeline overlaps with compute.  Ordering is by the hand-placed s_waitcnt vmcnt(N) + s_barrier.
// (The compiler's own vmcnt(N) for ordinary loads stays correct: uncounted younger operations only make it wait longer.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dvt_dma16(const void* src, const char* lds_dst) {
  const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_dst);   // generic LDS address: low word = offset
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(l) : "memory", "m0");
}
#pragma clang diagnostic pop
