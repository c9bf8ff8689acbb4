"""Summarise a rocprofv3 kernel trace CSV by (kernel, grid, block): calls, average and total time.
    python3 tools/dev/trace_by_grid.py <kernel_trace.csv> [filter-substring]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(list)
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)[:64]
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    agg[(n, g, int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
out = sorted(((sum(v), len(v), sum(v) / len(v), k) for k, v in agg.items()), reverse=True)
tot = sum(t for t, *_ in out)
for t, c, a, k in out:
    if flt in k[0]:
        print(f"{t:10.1f} us {100 * t / tot:5.1f}%  {c:5d} calls {a:8.1f} avg  {k}")
print(f"total {tot:.1f} us")
