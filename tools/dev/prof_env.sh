#!/bin/bash
# Dev: tools/dev/prof_wl.sh with one environment variable set for the profiled run.  usage: prof_env.sh VAR=VALUE <workload> <outname>
export "$1"; shift
exec bash "$(dirname "$0")/prof_wl.sh" "$@"
