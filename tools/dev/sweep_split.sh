#!/bin/bash
# Dev: sweep of the convolution split-K planner knobs on the frametransformer workload
R=$GRAFT_REPO_ROOT
run() { env "$@" timeout -k 10 200 python $R/bench.py --workload frametransformer --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run DVT_CONV_SPLIT_FILL=0
run DVT_CONV_SPLIT_FILL=80 DVT_CONV_SPLIT_TARGET=10
run DVT_CONV_SPLIT_FILL=60 DVT_CONV_SPLIT_TARGET=10
run DVT_CONV_SPLIT_FILL=80 DVT_CONV_SPLIT_TARGET=15
run DVT_CONV_SPLIT_FILL=80 DVT_CONV_SPLIT_TARGET=20
run DVT_CONV_SPLIT_FILL=80 DVT_CONV_SPLIT_TARGET=10 DVT_CONV_SPLIT_MINK=12
done
