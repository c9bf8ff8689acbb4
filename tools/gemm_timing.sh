#!/bin/bash
# Dev: build tools/_bin/libdvt_hip_gtiming.so = product objects with gemm256.hip recompiled under -DDVT_GEMM_TIMING
# (per-workgroup s_memtime stamps); then `python tools/gemm_timing_probe.py` on the GPU box.
set -e
cd "$(dirname "$0")/.."
PKG=data-efficient-video-transformers_amd
mkdir -p tools/_bin
python $PKG/build.py >/dev/null
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -DDVT_GEMM_TIMING -I include -c $PKG/csrc/gemm256.hip -o tools/_bin/gemm256_timing.o
OBJS=$(ls $PKG/csrc/_build/*.o | grep -v gemm256.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_bin/libdvt_hip_gtiming.so $OBJS tools/_bin/gemm256_timing.o
echo built tools/_bin/libdvt_hip_gtiming.so
