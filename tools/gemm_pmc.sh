#!/bin/bash
# SQ counter pass over tools/fwd_gemm_probe.py; output under gpurun_out/gemm_pmc (then tools/pmc_by_grid.py).
set -e
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/gemm_pmc"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/p1" -- python3 "$ROOT/tools/fwd_gemm_probe.py" > "$OUT/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p2" -- python3 "$ROOT/tools/fwd_gemm_probe.py" > "$OUT/p2.log" 2>&1
echo done
