#!/bin/bash
# Dev: same-box A/B of the window forward and the virtual BatchNorm on the frametransformer workload
R=$GRAFT_REPO_ROOT
run() { env "$@" timeout -k 10 200 python $R/bench.py --workload frametransformer --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run DVT_WINDOW_FWD=0 DVT_WINDOW_VIRTUAL_BN=0
run DVT_WINDOW_FWD=1 DVT_WINDOW_VIRTUAL_BN=0
run DVT_WINDOW_FWD=1 DVT_WINDOW_VIRTUAL_BN=1
done
