#!/bin/bash
# kernel trace of tools/dev/l4_probe.py: per launch name / grid / duration, in launch order of the timed loops
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/l4_probe; rm -rf $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $R/tools/dev/l4_probe.py 4 > $out.log 2>&1
grep "us/launch" $out.log
python3 - $out <<'PY'
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
agg = collections.OrderedDict()
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::|^void |_ZN12_GLOBAL__N_1\d+", "", r["Kernel_Name"]).split("(")[0][:64]
    if "gemm" not in n and "splitk" not in n and "reduce" not in n: continue
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // int(r["Workgroup_Size_X"])
    k = (n, g, int(r["Workgroup_Size_X"]), r["LDS_Block_Size"])
    agg.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    v = sorted(v)
    print(f"{v[len(v)//2]:7.1f} us  n={len(v):2d}  wgs={k[1]:5d} x{k[2]:4d} lds={k[3]:>6}  {k[0]}")
PY
