#!/bin/bash
# SQ counter passes over tools/dev/attn_cls_probe.py; output under gpurun_out/cls_pmc.
set -e
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/cls_pmc"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="$ROOT/tools/dev/attn_cls_probe.py"
python3 "$P" 20 > "$OUT/time.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d "$OUT/p1" -- python3 "$P" 2 > "$OUT/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p2" -- python3 "$P" 2 > "$OUT/p2.log" 2>&1
echo done
