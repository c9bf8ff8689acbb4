#!/bin/bash
# SQ / TCC counter passes over tools/dev/l4_probe.py (implicit-GEMM convolutions of R(2+1)D layers 3 - 4), per (kernel, grid).
# One rocprofv3 run per counter group, --pmc with --kernel-trace only; the program itself follows `--`.
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/l4_pmc"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="$ROOT/tools/dev/l4_probe.py"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/t0" -- python3 "$P" 4 > "$OUT/t0.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d "$OUT/p1" -- python3 "$P" 2 > "$OUT/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_WAVES --output-format csv -d "$OUT/p2" -- python3 "$P" 2 > "$OUT/p2.log" 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/p3" -- python3 "$P" 2 > "$OUT/p3.log" 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum --output-format csv -d "$OUT/p4" -- python3 "$P" 2 > "$OUT/p4.log" 2>&1
python3 - "$OUT" <<'PY'
import collections, csv, glob, re, sys
out = sys.argv[1]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|^void |_ZN12_GLOBAL__N_1\d+", "", n)
    return n.split("(")[0][:58]
def key(r):
    if "Grid_Size_X" in r:
        g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // int(r["Workgroup_Size_X"])
    else:
        g = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
    return (short(r["Kernel_Name"]), g)
dur = collections.defaultdict(list)
for f in glob.glob(f"{out}/t0/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[key(r)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[key(r)][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = [k for k in cnt if "gemm_dma" in k[0]]
cols = sorted({c for k in names for c in cnt[k]})
print("| kernel | workgroups | us | " + " | ".join(cols) + " |")
print("|---|---|---|" + "---|" * len(cols))
for k in sorted(names):
    d = sorted(dur.get(k, [0])); med = d[len(d) // 2]
    print(f"| `{k[0]}` | {k[1]} | {med:.1f} | " + " | ".join(f"{sum(cnt[k][c]) / max(1, len(cnt[k][c])):.4g}" if cnt[k].get(c) else "-" for c in cols) + " |")
PY
for f in "$OUT"/p*.log; do grep -il "error\|invalid\|not found" $f; done 2>/dev/null | head
