#!/bin/bash
# Dev: per-step total and per-launch median of the kernels matching a pattern, inside a workload's captured step, under one or more
# builds of the library.  usage: kern_ab.sh WORKLOAD PATTERN lib.so|in-tree ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; wl=$1; pat=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  out=$R/gpurun_out/kern_ab; rm -rf $out
  if [ "$lib" != "in-tree" ]; then export DVT_LIB_PATH=$R/$lib; else unset DVT_LIB_PATH; fi
  rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $R/bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --detail-out '' > $out.log 2>&1
  echo "== $lib"
  python3 - $out "$pat" $R <<'PY'
import collections, glob, re, sys
sys.path.insert(0, sys.argv[3] + "/tools/dev")
import trace_steps
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows, steps = trace_steps.load_steps(f, 3)
d = collections.defaultdict(list)
span = 0.0
for a, b in steps:
    span += (int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
    for r in rows[a:b]:
        if re.search(sys.argv[2], r["Kernel_Name"]):
            d[trace_steps.short(r["Kernel_Name"])[:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
n = len(steps)
tot = 0.0
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    tot += sum(v) / n
    v = sorted(v)
    print(f"{sum(v)/n:8.1f} us/step {v[len(v)//2]:7.1f} us median  n/step={len(v)/n:5.1f}  {k}")
print(f"{tot:8.1f} us/step in all matching; step span {span/n/1e3:.3f} ms ({n} steps)")
PY
  rm -rf $out
done
