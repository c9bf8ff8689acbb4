#!/bin/bash
# Dev: same-box A/B of a bench.py switch over workloads.  usage: ab_arg.sh --switch "wl1 wl2"
R=$GRAFT_REPO_ROOT; sw=$1; shift
for wl in $1; do
  for rep in 1 2; do
    for v in "" "$sw"; do
      timeout -k 10 200 python $R/bench.py --workload $wl --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --no-secondary $v 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', '$wl', d['value'], d['ms_per_step'])"
    done
  done
done
