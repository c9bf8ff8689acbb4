#!/bin/bash
# Dev: libdvt_hip_c3ablN.so = product objects with conv3x3.hip recompiled under -DDVT_C3_ABL=N
set -e
cd "$(dirname "$0")/../.."
PKG=data-efficient-video-transformers_amd
mkdir -p tools/_bin
python $PKG/build.py >/dev/null
for n in 1 2 3; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -DDVT_C3_ABL=$n -I include -c $PKG/csrc/conv3x3.hip -o tools/_bin/conv3_abl$n.o
  OBJS=$(ls $PKG/csrc/_build/*.o | grep -v conv3x3.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_bin/libdvt_hip_c3abl$n.so $OBJS tools/_bin/conv3_abl$n.o -ldl
done
echo built
