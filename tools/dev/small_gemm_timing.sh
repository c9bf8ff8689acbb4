#!/bin/bash
# Dev: tools/_bin/libdvt_hip_stiming.so = product objects with gemm_small.hip recompiled under -DDVT_SMALL_TIMING
set -e
cd "$(dirname "$0")/../.."
PKG=data-efficient-video-transformers_amd
mkdir -p tools/_bin
python $PKG/build.py >/dev/null
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -DDVT_SMALL_TIMING -I include -c $PKG/csrc/gemm_small.hip -o tools/_bin/gemm_small_timing.o
OBJS=$(ls $PKG/csrc/_build/*.o | grep -v gemm_small.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_bin/libdvt_hip_stiming.so $OBJS tools/_bin/gemm_small_timing.o -ldl
echo built tools/_bin/libdvt_hip_stiming.so
