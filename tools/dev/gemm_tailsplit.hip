// Dev probe: N = 512 launches (394 tiles of 256x256 = 1.54 rounds on 256 CUs).  A: one launch.  B: the first 128 row tiles
// (one full round) as 256x256 tiles, the remaining 17,664 rows as 256x128 tiles (276 workgroups, two may share a CU).
//   hipcc --offload-arch=gfx950 -O3 tools/dev/gemm_tailsplit.hip <csrc>/runtime.hip -o tools/_bin/gemm_tailsplit
#include "../../data-efficient-video-transformers_amd/csrc/gemm256.hip"
#include <vector>

int main() {
  const int Mt = 50432;
  struct Shape { const char* name; int N, K; bool ak, bk; int epi; };
  Shape shapes[] = {
    {"ff2_fwd   K=2048 res", 512, 2048, true, true, DVT_EPI_RESIDUAL},
    {"proj_fwd  K=512 res", 512, 512, true, true, DVT_EPI_RESIDUAL},
    {"ff1_dgrad K=2048", 512, 2048, true, false, DVT_EPI_NONE},
    {"qkv_dgrad K=1536", 512, 1536, true, false, DVT_EPI_NONE},
    {"proj_dgrad K=512", 512, 512, true, false, DVT_EPI_NONE},
  };
  bf16 *A, *B, *C, *RES; float* bias;
  hipMalloc(&A, (size_t)Mt * 2048 * 2); hipMalloc(&B, (size_t)2048 * 2048 * 2); hipMalloc(&C, (size_t)Mt * 512 * 2);
  hipMalloc(&RES, (size_t)Mt * 512 * 2); hipMalloc(&bias, 4096 * 4);
  hipMemset(A, 0x11, (size_t)Mt * 2048 * 2); hipMemset(B, 0x11, (size_t)2048 * 2048 * 2); hipMemset(RES, 0, (size_t)Mt * 512 * 2);
  hipMemset(bias, 0, 4096 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 30;
  for (auto& s : shapes) {
    GemmParams p{};
    p.A = A; p.B = B; p.C = C; p.M = Mt; p.N = s.N; p.K = s.K;
    p.lda = s.K; p.ldb = s.bk ? s.K : s.N; p.ldc = s.N;
    p.epilogue = s.epi; p.bias = bias; p.residual = RES; p.ldr = s.N; p.alpha = 1.f; p.k_per_split = s.K;
    auto time = [&](int mode, int m1rows) {
      float best = 1e9f;
      for (int w = 0; w < 2; ++w) {
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int it = 0; it < reps; ++it) {
          if (mode == 0) dvt_gemm_dma_launch(p, s.ak, s.bk, 1, 0, 0);
          else {
            GemmParams p1 = p, p2 = p;
            p1.M = m1rows;
            p2.M = Mt - m1rows; p2.A = A + (size_t)m1rows * s.K; p2.C = (bf16*)C + (size_t)m1rows * s.N;
            p2.residual = RES + (size_t)m1rows * s.N;
            dvt_gemm_dma_launch(p1, s.ak, s.bk, 1, 0, 0);
            dvt_gemm_dma_launch(p2, s.ak, s.bk, 1, mode == 1 ? 1 : 0, 0);
          }
        }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      return best * 1e3 / reps;
    };
    printf("%-22s one launch %7.1f us | 128 row tiles 256x256 + rest 256x128 %7.1f us | + rest 256x256 %7.1f us\n", s.name,
           time(0, 0), time(1, 32768), time(2, 32768));
  }
  return 0;
}
