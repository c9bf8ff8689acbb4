#!/bin/bash
# Dev: same-box A/B of the round-4 tree (tools/_bin/r04tree, `git archive d323b1d` + build) against the working tree, per workload.
R=$GRAFT_REPO_ROOT
for wl in vivit pyramid crossmodal frametransformer longclip; do
  for rep in 1 2; do
    (cd $R/tools/_bin/r04tree && timeout -k 10 200 python bench.py --workload $wl --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('r04 ', '$wl', d['value'], d['ms_per_step'])")
    (cd $R && timeout -k 10 200 python bench.py --workload $wl --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('r05 ', '$wl', d['value'], d['ms_per_step'])")
  done
done
