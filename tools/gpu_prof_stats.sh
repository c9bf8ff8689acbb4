#!/bin/bash
# kernel-trace stats of the headline bench only -> gpurun_out/prof/stats (cleared first)
set -e
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/prof"
rm -rf "$OUT/stats"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary > "$OUT/stats.log" 2>&1
cd "$ROOT" && python3 tools/profile_summarize.py gpurun_out/prof gpurun_out/prof_summary r03 > gpurun_out/prof_sum.log 2>&1
grep -o '"ms_per_step": [0-9.]*' "$OUT/stats.log" | head -1
