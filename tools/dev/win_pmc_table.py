"""Per-kernel means of the counter passes of tools/dev/win_pmc.sh, one markdown table (profiles/r06_conv3x1_bound.md)."""
import collections, csv, glob, re, sys
out = sys.argv[1]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    return n.split("(")[0][:64]


dur = collections.defaultdict(list)
for f in glob.glob(f"{out}/t0/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = [k for k in cnt if any(s in k for s in ("conv3x1", "conv3x3"))]
cols = sorted({c for k in names for c in cnt[k]})
print("| kernel | us | " + " | ".join(cols) + " |")
print("|---|---|" + "---|" * len(cols))
for k in sorted(names):
    d = sorted(dur.get(k, [0]))
    med = d[len(d) // 2]
    print(f"| `{k}` | {med:.1f} | " + " | ".join(f"{sum(cnt[k][c]) / max(1, len(cnt[k][c])):.4g}" if cnt[k].get(c) else "-" for c in cols) + " |")
