#!/bin/bash
# Dev: same-box A/B of the fused mid-BatchNorm backward on the frametransformer workload
R=$GRAFT_REPO_ROOT
run() { wl=$1; shift; env "$@" timeout -k 10 200 python $R/bench.py --workload $wl --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run frametransformer DVT_FUSED_MID_BN_BWD=0
run frametransformer DVT_FUSED_MID_BN_BWD=1
done
