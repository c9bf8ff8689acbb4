#!/bin/bash
# SQ counter passes over tools/dev/win_probe.py (the LDS window / halo kernels of layer 1); output under gpurun_out/win_pmc.
# One rocprofv3 run per counter group, --pmc with --kernel-trace only; the program itself follows `--`.
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/win_pmc"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="$ROOT/tools/dev/win_probe.py"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t0" -- python3 "$P" 6 > "$OUT/t0.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d "$OUT/p1" -- python3 "$P" 3 > "$OUT/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p2" -- python3 "$P" 3 > "$OUT/p2.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_DATA_FIFO_FULL --output-format csv -d "$OUT/p3" -- python3 "$P" 3 > "$OUT/p3.log" 2>&1
python3 "$ROOT/tools/dev/win_pmc_table.py" "$OUT" > "$OUT/table.md" 2>&1
tail -5 "$OUT"/p*.log | tail -20
cat "$OUT/table.md"
