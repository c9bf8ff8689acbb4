#!/bin/bash
# Dev: same-box A/B of one environment switch (VAR=0 / VAR=1) on the CNN workloads:  ab_env2.sh VAR [workloads...]
R=$GRAFT_REPO_ROOT
V=$1; shift
WLS=${@:-frametransformer}
run() { wl=$1; shift; env "$@" timeout -k 10 200 python $R/bench.py --workload $wl --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for wl in $WLS; do
run $wl $V=0
run $wl $V=1
done
done
