#!/bin/bash
# Dev: same-box A/B of one environment variable (set / unset) on bench workloads.  usage: ab_env_ft.sh VAR "wl1 wl2"
R=$GRAFT_REPO_ROOT; var=$1
for wl in $2; do
  for rep in 1 2; do
    for v in "" 1; do
      if [ -z "$v" ]; then unset $var; else export $var=$v; fi
      timeout -k 10 200 python $R/bench.py --workload $wl --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v', '$wl', d['value'], d['ms_per_step'])"
    done
  done
done
