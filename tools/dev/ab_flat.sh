#!/bin/bash
# Dev: same-box runs of the CNN workloads (this tree); window forward / virtual BatchNorm switches on the frametransformer one
R=$GRAFT_REPO_ROOT
run() { wl=$1; shift; env "$@" timeout -k 10 200 python $R/bench.py --workload $wl --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run frametransformer DVT_WINDOW_FWD=0 DVT_WINDOW_VIRTUAL_BN=0
run frametransformer DVT_WINDOW_FWD=1 DVT_WINDOW_VIRTUAL_BN=0
run frametransformer DVT_WINDOW_FWD=1 DVT_WINDOW_VIRTUAL_BN=1
run pyramid A=1
run crossmodal A=1
done
