#!/bin/bash
# per-launch time of the FF1 + GELU GEMM (gemm_dma_kernel<.., 3, 1, ..>) in the vivit step under two builds of the library
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  out=$R/gpurun_out/gelu_ab; rm -rf $out
  DVT_LIB_PATH=$R/$lib rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --detail-out '' > $out.log 2>&1
  echo "== $lib"
  python3 - $out <<'PY'
import collections, csv, glob, sys
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm_dma" in n:
            d[(n[n.find("gemm_dma"):][:70], str(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v[len(v)//3:])
    if len(v) >= 6: print(f"{v[len(v)//2]:8.1f} us median (n={len(v)})  {k}")
PY
  rm -rf $out
done
