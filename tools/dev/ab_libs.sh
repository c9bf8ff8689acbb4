#!/bin/bash
# Dev: same-box A/B of several builds of the library:  ab_libs.sh "a.so b.so .." [workloads...]   (paths relative to the repo)
R=$GRAFT_REPO_ROOT
LIBS=$1; shift
WLS=${@:-frametransformer}
run() { wl=$1; lib=$2; timeout -k 10 200 python $R/tools/dev/bench_with_lib.py $R/$lib --workload $wl --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --no-secondary --detail-out '' 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $lib', d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
for wl in $WLS; do
for lib in $LIBS; do run $wl $lib; done
done
done
