#!/bin/bash
# Dev: build tools/_bin/libdvt_hip_timing.so = the product objects with attention.hip recompiled under -DDVT_ATTN_TIMING
# (per-workgroup s_memtime stamps of the dk/dv kernel into the backward workspace).  Run here (cross-compiles), then
# `python tools/attn_timing_probe.py` on the GPU box.
set -e
cd "$(dirname "$0")/.."
PKG=data-efficient-video-transformers_amd
mkdir -p tools/_bin
python $PKG/build.py >/dev/null
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -DDVT_ATTN_TIMING -I include -c $PKG/csrc/attention.hip -o tools/_bin/attention_timing.o
OBJS=$(ls $PKG/csrc/_build/*.o | grep -v attention.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_bin/libdvt_hip_timing.so $OBJS tools/_bin/attention_timing.o
echo built tools/_bin/libdvt_hip_timing.so
