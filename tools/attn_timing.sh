#!/bin/bash
# Dev: build tools/_bin/libdvt_hip_timing.so = the product objects with the INSTRUMENTED attention.hip of commit 0102df2
# (per-workgroup s_memtime stamps under -DDVT_ATTN_TIMING; the stamps were removed from the product source in round 4 and
# live in that commit).  Run here (cross-compiles), then `python tools/attn_timing_probe.py` on the GPU box.
set -e
cd "$(dirname "$0")/.."
PKG=data-efficient-video-transformers_amd
mkdir -p tools/_bin
python $PKG/build.py >/dev/null
git show 0102df2:$PKG/csrc/attention.hip > $PKG/csrc/_attention_timing_dev.hip     # beside common.h for its includes
trap "rm -f $PKG/csrc/_attention_timing_dev.hip" EXIT
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -DDVT_ATTN_TIMING -I include -x hip -c $PKG/csrc/_attention_timing_dev.hip -o tools/_bin/attention_timing.o
OBJS=$(ls $PKG/csrc/_build/*.o | grep -v "/attention.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_bin/libdvt_hip_timing.so $OBJS tools/_bin/attention_timing.o
echo built tools/_bin/libdvt_hip_timing.so
