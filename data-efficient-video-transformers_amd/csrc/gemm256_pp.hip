// gemm256_pp.hip -- configuration 5 of the LDS-DMA GEMM (gemm256.hip: the 256x256x64 tile with its two wave rows in
// antiphase) as a module of its own: dvt_gemm_dma_launch_pp.
#define DVT_GEMM256_UNIT 1
#include "gemm256.hip"
