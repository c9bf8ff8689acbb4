"""Per-STEP kernel table from a rocprofv3 kernel trace: steps are the intervals between consecutive optimizer launches
(`adamw_fused_kernel`), so one-time set-up work (parameter copies, warm-up allocations) is not smeared over the steps.
The last <keep> full steps are averaged, split by (kernel, grid, block).

    python3 tools/dev/trace_steps.py <kernel_trace.csv> [keep=3] [top=60] [ordered listing of the last step -> file]
"""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    n = re.sub(r"^void ", "", n)
    return n[:72]


def load_steps(path, keep=3):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
    # a step = the kernels after one optimizer launch up to and including the next, if there are enough of them
    steps = [(a + 1, b + 1) for a, b in zip(marks, marks[1:]) if b - a > 20]
    return rows, steps[-keep:]


def table(path, keep=3, top=60, out=sys.stdout, order_file=None):
    rows, steps = load_steps(path, keep)
    if not steps:
        out.write("no steps found (no adamw launches?)\n")
        return
    agg = collections.defaultdict(lambda: [0, 0.0])
    span = 0.0
    for a, b in steps:
        span += (int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e6
        for r in rows[a:b]:
            g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            k = (short(r["Kernel_Name"]), g, int(r["Workgroup_Size_X"]))
            agg[k][0] += 1
            agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = len(steps)
    tot = sum(v[1] for v in agg.values()) / n
    out.write(f"# {n} steps; kernel time {tot / 1e3:.3f} ms/step, wall span {span / n:.3f} ms/step, "
              f"{sum(v[0] for v in agg.values()) / n:.0f} launches/step\n")
    out.write("| kernel | grid | block | calls/step | ms/step | avg us | % |\n|---|---|---|---|---|---|---|\n")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        out.write(f"| `{k[0]}` | {k[1]} | {k[2]} | {v[0] / n:.2f} | {v[1] / n / 1e3:.4f} | {v[1] / v[0]:.1f} | {100 * v[1] / n / tot:.1f} |\n")
    byname = collections.defaultdict(lambda: [0, 0.0])
    for k, v in agg.items():
        byname[k[0]][0] += v[0]
        byname[k[0]][1] += v[1]
    out.write("\n# by kernel name\n")
    for k, v in sorted(byname.items(), key=lambda kv: -kv[1][1])[:40]:
        out.write(f"{v[1] / n / 1e3:8.4f} ms {v[0] / n:7.2f} calls  {k}\n")
    # the last step in launch order (small-launch zone work: which launches, how long, how far apart)
    if order_file:
        a, b = steps[-1]
        t0 = int(rows[a]["Start_Timestamp"])
        prev_end = t0
        with open(order_file, "w") as fh:
            for i, r in enumerate(rows[a:b]):
                s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
                fh.write(f"{i:4d} t={(s - t0) / 1e3:9.1f} us  dur={(e - s) / 1e3:7.1f}  gap={(s - prev_end) / 1e3:6.1f}  "
                         f"grid={g // int(r['Workgroup_Size_X']):6d}x{r['Workgroup_Size_X']:>4}  {short(r['Kernel_Name'])}\n")
                prev_end = max(prev_end, e)


if __name__ == "__main__":
    table(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3, int(sys.argv[3]) if len(sys.argv) > 3 else 60,
          order_file=sys.argv[4] if len(sys.argv) > 4 else None)
