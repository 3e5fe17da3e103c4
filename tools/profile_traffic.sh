#!/bin/bash
# The two traffic passes of tools/profile_bench.sh alone (FETCH_SIZE, WRITE_SIZE), for profiles/traffic.json after a change
# that left the kernels as they were.   usage: tools/profile_traffic.sh <tag> [bench args...]
set -u
TAG=${1:?usage: profile_traffic.sh <tag> [bench args...]}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT="gpurun_out/prof_$TAG"; mkdir -p "$OUT"
run_pmc() { n=$1; ctrs=$2; shift 2; rm -rf "$OUT/pmc_$n"; rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/pmc_$n" -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-side-lines --serial 1 "$@" > "$OUT/bench_$n.log" 2>&1; echo "pmc $n rc=$?"; }
run_pmc fetch "FETCH_SIZE" "$@"
run_pmc write "WRITE_SIZE" "$@"
