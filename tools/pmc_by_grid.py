"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter values per (kernel family, grid size)."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "gemm_dma" not in kn and "attn" not in kn:
                continue
            fam = kn.split("(")[0].split("<")[0].split("::")[-1]
            acc[(fam, r["Grid_Size"], r["Workgroup_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, dd in sorted(acc.items()):
            print(k, {c: round(sum(v) / len(v)) for c, v in dd.items()})
