// Dev experiment (not shipped): PRICE of the store-wave persistent tile from measured components.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/dev/gemm_storewave_price.hip <csrc>/gemm256_pp.hip <csrc>/runtime.hip -o gemm_sw_price
// The "register-transpose epilogue" columns need tools/dev/perm_epilogue.patch applied to the csrc tree first
// (git apply tools/dev/perm_epilogue.patch; build with -DDVT_PERM_EPI=2 for both forms in one binary, =1 / =0 for one form
// without the registers of the other); without the patch they repeat the staged epilogue (define DVT_NO_PERM).
//
// The proposal: 256 x 128 x 64 tiles (2 x 48 KiB stages + a 64 KiB parked C tile), one persistent workgroup per CU, the
// finished tile parked in LDS and drained by dedicated store waves while the compute waves run the next tile's main loop.
// Its time per launch is bounded below by  max(T_loop(256x128x64), T_epilogue)  -- the main loops of a CU's tiles back to
// back, the stores hidden completely -- plus whatever does not overlap.  Both terms are measurable on the shipped kernel
// family without building the persistent form: the same gemm_dma_kernel, per configuration, (a) as shipped, (b) with its
// global stores compiled to nothing (everything else -- DMA, MFMAs, LDS staging, epilogue arithmetic, residual / aux loads
// -- still runs), (c) with K = 0 (no main loop: staging + arithmetic + stores of every tile).
#define DVT_C_STORE(ptr, vals)                                                            \
  do {                                                                                    \
    if (p.stream_out > 0) store8_nt<E>(ptr, vals);                                        \
    else if (p.stream_out == 0) store8<E>(ptr, vals);                                     \
  } while (0)
#define DVT_C_STORE16(ptr, v4)                                                            \
  do {                                                                                    \
    if (p.stream_out >= 0) store16(ptr, v4, p.stream_out > 0);                            \
    else asm volatile("" ::"v"(v4));                                                      \
  } while (0)
#include "../../data-efficient-video-transformers_amd/csrc/gemm256.hip"
#include <vector>
#include <algorithm>
#include <string.h>
#include <random>

namespace {

template <typename E, bool AK, bool BK, int CFG, int EPI>
void run_cfg(const GemmParams& pin, int mode, hipStream_t st) {          // mode 0 full, 1 no stores, 2 no main loop; + 4: register-transpose epilogue
  typedef Cfg<CFG> C;
  GemmParams p = pin;
#ifndef DVT_NO_PERM
  p.perm_epi = mode >= 4;
#endif
  mode &= 3;
  p.stream_out = mode == 1 ? -1 : 1;
  if (mode == 2) { p.K = 0; p.k_per_split = 0; }
  const int tiles_m = (int)dvt_cdiv(p.M, C::TM);
  p.tiles_n = (int)dvt_cdiv(p.N, C::TN);
  const dim3 grid((unsigned)(tiles_m * p.tiles_n), 1, 1), block(C::NW * 64);
  launch_one<E, AK, BK, CFG, EPI, OUT_BF16>(p, grid, block, smem_bytes<CFG>(), st);
}

void fill(bf16* d, size_t n, float scale, unsigned seed) {
  std::vector<unsigned short> h(n);
  std::mt19937 rng(seed);
  std::normal_distribution<float> dist(0.f, scale);
  for (size_t i = 0; i < n; ++i) {
    float f = dist(rng);
    unsigned u; memcpy(&u, &f, 4);
    h[i] = (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
  }
  hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
}

struct Shape { const char* name; int N, K, epi; bool bk; int shipped; };

}  // namespace

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 50432;
  const Shape shapes[] = {
      {"FF1 + GELU (two outputs)", 2048, 512, DVT_EPI_GELU, true, 3},
      {"FF2 dgrad + GELU' (aux read)", 2048, 512, DVT_EPI_DGELU, false, 3},
      {"QKV (plain)", 1536, 512, DVT_EPI_NONE, true, 5},
      {"FF2 + residual", 512, 2048, DVT_EPI_RESIDUAL, true, 5},
  };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  bf16 *A, *B, *R; float* bias;
  hipMalloc(&A, (size_t)M * 2048 * 2); hipMalloc(&B, (size_t)2048 * 2048 * 2); hipMalloc(&R, (size_t)M * 2048 * 2);
  hipMalloc(&bias, 2048 * 4);
  fill(A, (size_t)M * 2048, 1.0f, 1); fill(B, (size_t)2048 * 2048, 0.05f, 2); fill(R, (size_t)M * 2048, 1.0f, 3);
  { std::vector<float> hb(2048); for (int i = 0; i < 2048; ++i) hb[i] = 0.01f * (i % 17 - 8); hipMemcpy(bias, hb.data(), 2048 * 4, hipMemcpyHostToDevice); }
  constexpr int NBUF = 3;                       // rotating output buffers (back-to-back launches into the same lines run slower)
  bf16 *Cb[NBUF], *Ub[NBUF];
  for (int i = 0; i < NBUF; ++i) { hipMalloc(&Cb[i], (size_t)M * 2048 * 2); hipMalloc(&Ub[i], (size_t)M * 2048 * 2); }
  for (const Shape& s : shapes) {
    GemmParams p{};
    p.A = A; p.B = B; p.M = M; p.N = s.N; p.K = s.K; p.lda = s.K; p.ldb = s.bk ? s.K : s.N; p.ldc = s.N; p.ldaux = s.N; p.ldr = s.N;
    p.epilogue = s.epi; p.bias = s.epi == DVT_EPI_DGELU ? nullptr : bias; p.alpha = 1.f; p.k_per_split = s.K; p.elem = DVT_BF16;
    p.residual = s.epi == DVT_EPI_RESIDUAL ? R : nullptr;
    const double flops = 2.0 * M * s.N * s.K;
    const double obytes = (double)M * s.N * 2 * (s.epi == DVT_EPI_GELU ? 2 : 1);
    printf("== %s: [%d, %d] x [%d, %d], %.1f GF, %.0f MB of stores\n", s.name, M, s.K, s.N, s.K, flops * 1e-9, obytes * 1e-6);
    auto time_it = [&](auto&& fn) {
      for (int it = 0; it < 3; ++it) fn(it);
      hipDeviceSynchronize();
      std::vector<double> tt;
      for (int round = 0; round < 5; ++round) {
        hipEventRecord(e0, 0);
        for (int it = 0; it < 6; ++it) fn(it);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        tt.push_back(ms * 1e3 / 6);
      }
      std::sort(tt.begin(), tt.end());
      return tt[2];
    };
    auto with_bufs = [&](GemmParams q, int it) {
      q.C = Cb[it % NBUF];
      q.aux = s.epi == DVT_EPI_GELU ? (void*)Ub[it % NBUF] : (s.epi == DVT_EPI_DGELU ? (void*)R : nullptr);
      return q;
    };
    {   // shipped configuration through the product's own launcher (configuration 5 lives in gemm256_pp.hip)
      const double us = time_it([&](int it) { dvt_gemm_dma_launch(with_bufs(p, it), true, s.bk, 1, s.shipped, 0); });
      printf("  shipped (cfg %d)                         %7.1f us  %6.1f TF/s\n", s.shipped, us, flops / us * 1e-6);
    }
    for (int cfg : {0, 3, 1, 7, 9}) {
      double t[6];
      for (int mode : {0, 1, 2, 4, 5}) {
        t[mode] = time_it([&](int it) {
          GemmParams q = with_bufs(p, it);
#define RUN(CFG)                                                                                                           \
  do {                                                                                                                     \
    if (s.epi == DVT_EPI_GELU) run_cfg<bf16, true, true, CFG, DVT_EPI_GELU>(q, mode, 0);                                   \
    else if (s.epi == DVT_EPI_DGELU) run_cfg<bf16, true, false, CFG, DVT_EPI_DGELU>(q, mode, 0);                           \
    else if (s.epi == DVT_EPI_RESIDUAL) run_cfg<bf16, true, true, CFG, DVT_EPI_RESIDUAL>(q, mode, 0);                      \
    else run_cfg<bf16, true, true, CFG, DVT_EPI_NONE>(q, mode, 0);                                                         \
  } while (0)
          if (cfg == 0) RUN(0); else if (cfg == 3) RUN(3); else if (cfg == 1) RUN(1); else if (cfg == 7) RUN(7); else RUN(9);
#undef RUN
        });
      }
      const char* nm = cfg == 0 ? "256x256x64, 8 waves " : cfg == 3 ? "256x256x64, 16 waves" : cfg == 1 ? "256x128x32, 2 WG/CU" : cfg == 7 ? "256x128x64, 8 waves " : "128x128x64, 2 WG/CU";
      printf("  cfg %d %s  full %7.1f  no stores %7.1f  no main loop %7.1f us | register-transpose epilogue: full %7.1f  no stores %7.1f us\n",
             cfg, nm, t[0], t[1], t[2], t[4], t[5]);
    }
    {   // the register-transpose epilogue against the staged one, every element (configuration 3 and 0)
      for (int cfg : {3, 0}) {
        GemmParams a = with_bufs(p, 0), b = with_bufs(p, 1);
        hipMemset(a.C, 0xFF, (size_t)M * s.N * 2); hipMemset(b.C, 0xEE, (size_t)M * s.N * 2);
        if (s.epi == DVT_EPI_GELU) { hipMemset(a.aux, 0xFF, (size_t)M * s.N * 2); hipMemset(b.aux, 0xEE, (size_t)M * s.N * 2); }
#define RUN2(CFG, Q, MODE)                                                                                                 \
  do {                                                                                                                     \
    if (s.epi == DVT_EPI_GELU) run_cfg<bf16, true, true, CFG, DVT_EPI_GELU>(Q, MODE, 0);                                   \
    else if (s.epi == DVT_EPI_DGELU) run_cfg<bf16, true, false, CFG, DVT_EPI_DGELU>(Q, MODE, 0);                           \
    else if (s.epi == DVT_EPI_RESIDUAL) run_cfg<bf16, true, true, CFG, DVT_EPI_RESIDUAL>(Q, MODE, 0);                      \
    else run_cfg<bf16, true, true, CFG, DVT_EPI_NONE>(Q, MODE, 0);                                                         \
  } while (0)
        if (cfg == 3) { RUN2(3, a, 0); RUN2(3, b, 4); } else { RUN2(0, a, 0); RUN2(0, b, 4); }
#undef RUN2
        hipDeviceSynchronize();
        std::vector<unsigned short> ha((size_t)M * s.N), hb((size_t)M * s.N);
        hipMemcpy(ha.data(), a.C, ha.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b.C, hb.size() * 2, hipMemcpyDeviceToHost);
        size_t diff = 0; for (size_t i = 0; i < ha.size(); ++i) diff += ha[i] != hb[i];
        size_t diffu = 0;
        if (s.epi == DVT_EPI_GELU) {
          hipMemcpy(ha.data(), a.aux, ha.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b.aux, hb.size() * 2, hipMemcpyDeviceToHost);
          for (size_t i = 0; i < ha.size(); ++i) diffu += ha[i] != hb[i];
        }
        printf("  cfg %d: register-transpose vs staged epilogue, bitwise mismatches: C %zu, aux %zu of %zu\n", cfg, diff, diffu, ha.size());
      }
    }
  }
  // the store path by itself: the chip's streaming-store rate for the FF1 outputs
  return 0;
}
