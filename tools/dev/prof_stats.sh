#!/bin/bash
# Dev: rocprofv3 kernel stats of the default bench (vivit), printed as a per-step table.  usage: tools/dev/prof_stats.sh <outname> [bench args]
out=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary "$@" > $GRAFT_REPO_ROOT/gpurun_out/$out.log 2>&1
