#!/bin/bash
# Dev: compact per-kernel resource table of one .hip file:  tools/dev/resusage.sh <file.hip> [grep pattern] [extra hipcc flags]
f=$1; pat=${2:-.}; shift; shift
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -I /root/repo/include "$@" -c $f -o /tmp/_res.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name:|    VGPRs:|ScratchSize|VGPRs Spill" | sed -E 's/.*remark: //; s/ \[-Rpass.*//' \
 | awk '/Function Name/{n=$3} /VGPRs:/{v=$2} /ScratchSize/{sc=$3} /VGPRs Spill/{print n, "vgpr="v, "scratch="sc, "spill="$3}' | grep -E "$pat" | c++filt -p 2>/dev/null | sed -E 's/\(anonymous namespace\):://'
