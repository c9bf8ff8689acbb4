#!/bin/bash
# Reproduces profiles/rNN_* on an MI355X box (run from the repo root through gpurun):
#   bash tools/run_profiles.sh && python3 tools/profile_summarize.py gpurun_out/prof profiles r04
# One rocprofv3 run per counter group (--pmc never together with other trace domains); the program itself follows `--`.
set -e
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/prof"
rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="$ROOT/bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$BENCH" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --detail-out '' > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$BENCH" --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --detail-out '' --no-graph > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$BENCH" --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --detail-out '' --no-graph > "$OUT/write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/mfma" -- python3 "$BENCH" --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --detail-out '' --no-graph > "$OUT/mfma.log" 2>&1
# the per-frame CNN encoder's kernels (secondary.pyramid.roofline of the bench line): traffic of the pyramid workload
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_pyramid" -- python3 "$BENCH" --workload pyramid --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --detail-out '' --no-graph > "$OUT/fetch_pyramid.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write_pyramid" -- python3 "$BENCH" --workload pyramid --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --detail-out '' --no-graph > "$OUT/write_pyramid.log" 2>&1
# the R(2+1)D encoder's kernels (secondary.frametransformer.roofline): traffic of the frametransformer workload
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_frametransformer" -- python3 "$BENCH" --workload frametransformer --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --detail-out '' --no-graph > "$OUT/fetch_frametransformer.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write_frametransformer" -- python3 "$BENCH" --workload frametransformer --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --detail-out '' --no-graph > "$OUT/write_frametransformer.log" 2>&1
# secondary workloads: kernel-time breakdown
for wl in pyramid frametransformer longclip; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$wl" -- python3 "$BENCH" --workload $wl --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --detail-out '' > "$OUT/stats_$wl.log" 2>&1
done
echo "profiles collected under $OUT"
