#!/bin/bash
# Dev: rocprofv3 kernel trace of one bench workload, summarised per step.  usage: tools/dev/prof_wl.sh <workload> <outname> [bench args]
wl=$1; out=$2; shift; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/$out
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$out -- python3 $R/bench.py --workload $wl --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary "$@" > $R/gpurun_out/$out.log 2>&1
f=$(find $R/gpurun_out/$out -name "*kernel_trace.csv" | head -1)
python3 $R/tools/dev/trace_steps.py $f 3 70 $R/gpurun_out/${out}_order.txt > $R/gpurun_out/$out.md 2>&1
rm -rf $R/gpurun_out/$out
head -3 $R/gpurun_out/$out.md
