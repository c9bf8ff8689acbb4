#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the forward GEMM probe (separate passes); then tools/pmc_by_grid.py
set -e
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/gemm_fetch"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/f" -- python3 "$ROOT/tools/fwd_gemm_probe.py" > "$OUT/f.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/w" -- python3 "$ROOT/tools/fwd_gemm_probe.py" > "$OUT/w.log" 2>&1
echo done
