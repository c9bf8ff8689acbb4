#!/bin/bash
# kernel durations of tools/dev/win_probe.py (rocprofv3 kernel trace), optionally against another build: win_time.sh [lib.so ...]
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
for lib in "${@:-in-tree}"; do
  OUT="$ROOT/gpurun_out/win_time"; rm -rf "$OUT"; mkdir -p "$OUT"
  if [ "$lib" != "in-tree" ]; then export DVT_LIB_PATH="$ROOT/$lib"; else unset DVT_LIB_PATH; fi
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/t0" -- python3 "$ROOT/tools/dev/win_probe.py" 8 > "$OUT/t0.log" 2>&1
  echo "== $lib"
  python3 - "$OUT" <<'PY'
import collections, csv, glob, re, sys
dur = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/t0/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::|^void |_ZN12_GLOBAL__N_1\d+", "", r["Kernel_Name"]).split("(")[0][:70]
        dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, d in sorted(dur.items()):
    if "conv3x" in n:
        d = sorted(d[2:]) or [0]
        print(f"{d[len(d)//2]:8.1f} us (min {d[0]:.1f}, n={len(d)})  {n}")
PY
  rm -rf "$OUT"
done
