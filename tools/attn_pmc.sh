#!/bin/bash
# SQ counter passes over tools/attn_probe.py (attention kernels of the metric shape only); output under gpurun_out/attn_pmc.
set -e
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/attn_pmc"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="$ROOT/tools/attn_probe.py"
python3 "$P" 197 20 > "$OUT/time.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d "$OUT/p1" -- python3 "$P" 197 3 > "$OUT/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p2" -- python3 "$P" 197 3 > "$OUT/p2.log" 2>&1
echo done
