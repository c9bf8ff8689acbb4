// Dev probe: does a split-K weight-gradient launch on a second stream fill the partial last round of a
// data-gradient launch?  Times A alone, B alone, and A||B on two streams.
//   hipcc --offload-arch=gfx950 -O3 tools/dev/gemm_overlap.hip <csrc>/runtime.hip -o tools/_bin/gemm_overlap
#include "../../data-efficient-video-transformers_amd/csrc/gemm256.hip"
#include <vector>
#include <string.h>

int main() {
  const int Mt = 50432;
  bf16 *A, *B, *C, *A2, *B2; float* slab; float* C2;
  hipMalloc(&A, (size_t)Mt * 2048 * 2); hipMalloc(&B, (size_t)2048 * 2048 * 2); hipMalloc(&C, (size_t)Mt * 2048 * 2);
  hipMalloc(&A2, (size_t)Mt * 2048 * 2); hipMalloc(&B2, (size_t)Mt * 512 * 2);
  hipMalloc(&slab, (size_t)16 * 2048 * 512 * 4); hipMalloc(&C2, (size_t)2048 * 512 * 4);
  hipMemset(A, 0, (size_t)Mt * 2048 * 2); hipMemset(B, 0, (size_t)2048 * 2048 * 2);
  hipMemset(A2, 0, (size_t)Mt * 2048 * 2); hipMemset(B2, 0, (size_t)Mt * 512 * 2);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  // A: ff1 dgrad [M,2048]x[2048,512] -> [M,512]  (394 tiles, 1.54 rounds)
  GemmParams pa{};
  pa.A = A; pa.B = B; pa.C = C; pa.M = Mt; pa.N = 512; pa.K = 2048; pa.lda = 2048; pa.ldb = 512; pa.ldc = 512;
  pa.epilogue = DVT_EPI_NONE; pa.alpha = 1.f; pa.k_per_split = 2048;
  // B: ff1 wgrad [2048,M]x[M,512], split 16 -> fp32 slabs (256 workgroups)
  GemmParams pb{};
  pb.A = A2; pb.B = B2; pb.C = C2; pb.M = 2048; pb.N = 512; pb.K = Mt; pb.lda = 2048; pb.ldb = 512; pb.ldc = 512;
  pb.epilogue = DVT_EPI_NONE; pb.alpha = 1.f; pb.out_f32 = 1;
  int kps = (int)(dvt_cdiv(dvt_cdiv(Mt, 16), 64) * 64); int split = (int)dvt_cdiv(Mt, kps);
  pb.k_per_split = kps; pb.slab = slab;
  hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
  const int reps = 20;
  auto run = [&](int mode) {
    for (int w = 0; w < 2; ++w) {
      hipDeviceSynchronize();
      hipEventRecord(e0, s1); hipStreamWaitEvent(s2, e0, 0);
      for (int it = 0; it < reps; ++it) {
        if (mode != 1) dvt_gemm_dma_launch(pa, true, false, 1, 0, s1);
        if (mode != 0) dvt_gemm_dma_launch(pb, false, false, split, 0, mode == 3 ? s1 : s2);
      }
      hipEventRecord(e2, s2); hipStreamWaitEvent(s1, e2, 0);
      hipEventRecord(e1, s1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / reps;
  };
  double a = run(0), b = run(1), ab = run(2), ser = run(3);
  printf("dgrad alone %.1f us, wgrad alone %.1f us, same stream %.1f us, two streams %.1f us\n", a, b, ser, ab);
  return 0;
}
