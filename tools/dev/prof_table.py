import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 9
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step: {tot/1e6/steps:.3f} ms ({steps:g} steps in trace)")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f'{float(r["TotalDurationNs"])/1e6/steps:8.3f} ms/step {int(r["Calls"])/steps:7.1f} calls {float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][:100]}')
