#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + two separate PMC passes of the same bench command.
# Usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" > $OUT/bench_trace.log 2>&1
# (counter passes run one kernel at a time: --serial 1 lets the host order the two streams, see vrg.h "serial_streams")
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --serial 1 "$@" > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --serial 1 "$@" > $OUT/bench_write.log 2>&1
python3 bench.py --steps 500 --warmup 20 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-400
find $OUT -name "*.csv" | head -20
