#!/bin/bash
# LDS counters (bank-conflict cycles against all LDS-array cycles) of every kernel of a bench workload:
#   tools/dev/lds_pmc.sh <workload> <outname>      -> gpurun_out/<outname>.md
wl=$1; out=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/$out
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$out -- python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-graph --detail-out '' > $R/gpurun_out/$out.log 2>&1
python3 - "$R/gpurun_out/$out" > $R/gpurun_out/$out.md <<'PY'
import collections, csv, glob, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::|^void |_ZN12_GLOBAL__N_1\d+", "", r["Kernel_Name"]).split("(")[0][:70]
        acc[(n, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    rows.append((m.get("SQ_LDS_BANK_CONFLICT", 0) * len(d["SQ_LDS_BANK_CONFLICT"]), k, m, len(d["SQ_LDS_BANK_CONFLICT"])))
print("| kernel | grid | launches | LDS_BANK_CONFLICT | LDS_IDX_ACTIVE | conflict share | IDX_ACTIVE / (GUI_ACTIVE x 256 CUs) |")
print("|---|---|---|---|---|---|---|")
for tot, k, m, n in sorted(rows, key=lambda t: -t[0])[:60]:
    ia, gui = m.get("SQ_LDS_IDX_ACTIVE", 0), m.get("GRBM_GUI_ACTIVE", 1)
    print(f"| `{k[0]}` | {k[1]} | {n} | {m.get('SQ_LDS_BANK_CONFLICT', 0):.3g} | {ia:.3g} | {m.get('SQ_LDS_BANK_CONFLICT', 0) / max(ia, 1):.2f} | {ia / max(gui * 256, 1):.2f} |")
PY
rm -rf $R/gpurun_out/$out
head -30 $R/gpurun_out/$out.md
