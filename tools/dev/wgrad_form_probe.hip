// Dev experiment: is the weight-gradient GEMM bound by its transposing LDS reads?  The shipped form reads BOTH operands
// mn-major (dy [rows][N_out] and x [rows][K_in], reduction over rows): ds_read_b64_tr_b16 for every fragment.  If a transposed
// copy x^T [K_in][rows] existed, dW^T = x^T dy would be the family's (A k-major, B mn-major) form: b128 reads on the 8-fragment
// side.  Times both forms at the metric shapes (split-K 16, fp32 slabs).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/dev/wgrad_form_probe.hip <csrc>/gemm256_pp.hip <csrc>/runtime.hip -o wgrad_form_probe
#include "../../data-efficient-video-transformers_amd/csrc/gemm256.hip"
#include <vector>
#include <algorithm>
int main() {
  const int R = 50432;
  struct Shape { const char* name; int M, N; bool ak; };
  Shape shapes[] = {
    {"ff1 wgrad  dy^T x   : [2048,R] x [R,512]  (mn,mn)  shipped", 2048, 512, false},
    {"ff1 wgrad  x^T dy   : [512,R]k x [R,2048] (k,mn)  with x^T", 512, 2048, true},
    {"qkv wgrad  dy^T x   : [1536,R] x [R,512]  (mn,mn)  shipped", 1536, 512, false},
    {"qkv wgrad  x^T dy   : [512,R]k x [R,1536] (k,mn)  with x^T", 512, 1536, true},
    {"ff2 wgrad  dy^T h   : [512,R] x [R,2048]  (mn,mn)  shipped", 512, 2048, false},
    {"ff2 wgrad  dy^T h   : [512,R]k x [R,2048] (k,mn)  with dy^T", 512, 2048, true},
    {"proj wgrad dy^T o   : [512,R] x [R,512]   (mn,mn)  shipped", 512, 512, false},
    {"proj wgrad          : [512,R]k x [R,512]  (k,mn)  with dy^T", 512, 512, true},
  };
  bf16 *A, *B; float* slab;
  hipMalloc(&A, (size_t)R * 2048 * 2); hipMalloc(&B, (size_t)R * 2048 * 2);
  hipMalloc(&slab, (size_t)32 * 2048 * 512 * 4);
  hipMemset(A, 0x3c, (size_t)R * 2048 * 2); hipMemset(B, 0x3c, (size_t)R * 2048 * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto& s : shapes) {
    GemmParams p{};
    p.A = A; p.B = B; p.C = nullptr; p.M = s.M; p.N = s.N; p.K = R;
    p.lda = s.ak ? R : s.M; p.ldb = s.N; p.ldc = s.N; p.alpha = 1.f; p.elem = DVT_BF16;
    const int split = 16;
    int kps = (int)(dvt_cdiv(dvt_cdiv(R, split), 64) * 64);
    p.k_per_split = kps; p.slab = slab;
    const int sp = (int)dvt_cdiv(R, kps);
    for (int cfg : {5, 0}) {
      for (int it = 0; it < 3; ++it) dvt_gemm_dma_launch(p, s.ak, false, sp, cfg, 0);
      hipDeviceSynchronize();
      std::vector<double> t;
      for (int round = 0; round < 5; ++round) {
        hipEventRecord(e0, 0);
        for (int it = 0; it < 10; ++it) dvt_gemm_dma_launch(p, s.ak, false, sp, cfg, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 100);
      }
      std::sort(t.begin(), t.end());
      printf("%-62s cfg%d %7.1f us %7.1f TF/s\n", s.name, cfg, t[2], 2.0 * s.M * s.N * R / (t[2] * 1e-6) / 1e12);
    }
  }
  return 0;
}
