#!/bin/bash
# Dev: same-box A/B of another checkout of the repo (built; e.g. `git archive HEAD | tar -x -C tools/_bin/headtree`) against the
# working tree:  ab_tree.sh tools/_bin/headtree [workloads...]
R=$GRAFT_REPO_ROOT
T=$1; shift
WLS=${@:-vivit}
run() { (cd $1 && timeout -k 10 200 python bench.py --workload $2 --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$3', '$2', d['value'], d['ms_per_step'])"); }
for rep in 1 2 3; do
for wl in $WLS; do
run $R/$T $wl base
run $R $wl work
done
done
