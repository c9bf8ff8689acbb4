// Dev tool (not shipped): times gemm256_kernel variants on the metric shapes.
//   hipcc --offload-arch=gfx950 -O3 -I include tools/gemm_bench.hip <csrc>/gemm256_pp.hip <csrc>/runtime.hip -o gemm_bench
// (the compile-time ablations 1..6 and the s_memtime stamps of rounds 1-2 were removed from the product source in round 3; they live in commit feba3d0)
#include "../data-efficient-video-transformers_amd/csrc/gemm256.hip"
#include <vector>
#include <algorithm>
#include <string.h>
#include <random>

static void fill(bf16* d, size_t n, float scale) {
  std::vector<unsigned short> h(n);
  std::mt19937 rng(123);
  std::normal_distribution<float> dist(0.f, scale);
  for (size_t i = 0; i < n; ++i) {
    float f = dist(rng);
    unsigned u; memcpy(&u, &f, 4);
    h[i] = (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
  }
  hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
}

int main(int argc, char** argv) {
  struct Shape { const char* name; int M, N, K; bool ak, bk; int epi; int split; };
  const int Mt = argc > 1 ? atoi(argv[1]) : 50432;
  Shape shapes[] = {
    {"qkv_fwd   [M,512]x[1536,512]^T", Mt, 1536, 512, true, true, DVT_EPI_NONE, 1},
    {"ff1_fwd   [M,512]x[2048,512]^T gelu", Mt, 2048, 512, true, true, DVT_EPI_GELU, 1},
    {"ff2_fwd   [M,2048]x[512,2048]^T res", Mt, 512, 2048, true, true, DVT_EPI_RESIDUAL, 1},
    {"proj_fwd  [M,512]x[512,512]^T res", Mt, 512, 512, true, true, DVT_EPI_RESIDUAL, 1},
    {"ff2_dgrad [M,512]x[512,2048] dgelu", Mt, 2048, 512, true, false, DVT_EPI_DGELU, 1},
    {"ff1_dgrad [M,2048]x[2048,512]", Mt, 512, 2048, true, false, DVT_EPI_NONE, 1},
    {"qkv_dgrad [M,1536]x[1536,512]", Mt, 512, 1536, true, false, DVT_EPI_NONE, 1},
    {"proj_dgrad [M,512]x[512,512]", Mt, 512, 512, true, false, DVT_EPI_NONE, 1},
    {"ff1_wgrad [2048,M]x[M,512] split16", 2048, 512, Mt, false, false, DVT_EPI_NONE, 16},
    {"square 4096^3 k,k", 4096, 4096, 4096, true, true, DVT_EPI_NONE, 1},
    {"square 4096^3 k,mn", 4096, 4096, 4096, true, false, DVT_EPI_NONE, 1},
    {"square 4096^3 mn,mn", 4096, 4096, 4096, false, false, DVT_EPI_NONE, 1},
  };
  size_t maxA = (size_t)(Mt > 4096 ? Mt : 4096) * 2048, maxB = (size_t)Mt * 2048, maxC = (size_t)Mt * 2048;
  bf16 *A, *B, *C, *AUX, *RES; float* bias; float* slab;
  hipMalloc(&A, maxA * 2); hipMalloc(&B, maxB * 2); hipMalloc(&C, maxC * 2);
  hipMalloc(&AUX, maxC * 2); hipMalloc(&RES, maxC * 2); hipMalloc(&bias, 4096 * 4);
  hipMalloc(&slab, (size_t)32 * 2048 * 512 * 4 + (size_t)4096 * 4096 * 4);
  fill(A, maxA, 1.0f); fill(B, maxB, 0.05f); fill(AUX, maxC, 1.0f); fill(RES, maxC, 1.0f);
  hipMemset(bias, 0, 4096 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto& s : shapes) {
    GemmParams p{};
    p.A = A; p.B = B; p.C = C; p.M = s.M; p.N = s.N; p.K = s.K;
    p.lda = s.ak ? s.K : s.M; p.ldb = s.bk ? s.K : s.N; p.ldc = s.N;
    p.epilogue = s.epi; p.out_f32 = 0; p.accumulate = 0; p.bias = bias;
    p.residual = RES; p.ldr = s.N; p.aux = AUX; p.ldaux = s.N; p.alpha = 1.f;
    int kps = (int)(dvt_cdiv(dvt_cdiv(s.K, s.split), 64) * 64);
    int split = (int)dvt_cdiv(s.K, kps);
    p.k_per_split = kps; p.slab = split > 1 ? slab : nullptr;
    if (split > 1) { p.bias = nullptr; }
    // configurations interleaved over several rounds, median per configuration: the clock the chip holds drifts by a few
    // per cent within seconds, which is the size of the differences being measured
    const int cfgs[4] = {0, 3, 5, 8};
    std::vector<double> t[4];
    for (int c = 0; c < 4; ++c) { for (int it = 0; it < 3; ++it) dvt_gemm_dma_launch(p, s.ak, s.bk, split, cfgs[c], 0); }
    hipDeviceSynchronize();
    for (int round = 0; round < 7; ++round)
      for (int c = 0; c < 4; ++c) {
        const int reps = 10;
        hipEventRecord(e0, 0);
        for (int it = 0; it < reps; ++it) dvt_gemm_dma_launch(p, s.ak, s.bk, split, cfgs[c], 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        t[c].push_back(ms * 1e3 / reps);
      }
    printf("%-40s", s.name);
    for (int c = 0; c < 4; ++c) {
      std::sort(t[c].begin(), t[c].end());
      const double us = t[c][t[c].size() / 2];
      printf("  cfg%d %7.1f us %7.1f TF/s", cfgs[c], us, 2.0 * s.M * s.N * s.K / (us * 1e-6) / 1e12);
    }
    printf("\n");
  }
  return 0;
}
