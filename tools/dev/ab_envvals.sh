#!/bin/bash
# Dev: same-box A/B of one environment variable over several values:  ab_envvals.sh VAR "v1 v2 .." [workloads...]
R=$GRAFT_REPO_ROOT
V=$1; VALS=$2; shift; shift
WLS=${@:-frametransformer}
run() { wl=$1; shift; env "$@" timeout -k 10 200 python $R/bench.py --workload $wl --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for wl in $WLS; do
for v in $VALS; do run $wl $V=$v; done
done
done
