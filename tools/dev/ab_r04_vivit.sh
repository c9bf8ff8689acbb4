#!/bin/bash
# Dev: same-box A/B of the round-4 tree against the working tree on the headline workload, alternating, 4 repetitions
R=$GRAFT_REPO_ROOT
for rep in 1 2 3 4; do
  (cd $R/tools/_bin/r04tree && timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('r04 ', d['value'], d['ms_per_step'])")
  (cd $R && timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('r05 ', d['value'], d['ms_per_step'])")
done
