#!/bin/bash
# Dev: same-box A/B of one environment variable between two VALUES over workloads.  usage: ab_envval.sh VAR valA valB "wl1 wl2"
R=$GRAFT_REPO_ROOT; var=$1; a=$2; b=$3
for wl in $4; do
  for rep in 1 2; do
    for v in $a $b; do
      env $var=$v timeout -k 10 200 python $R/bench.py --workload $wl --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --no-secondary --detail-out '' 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v', '$wl', d['value'], d['ms_per_step'])"
    done
  done
done
