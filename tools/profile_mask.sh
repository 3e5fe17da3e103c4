#!/bin/bash
# Run on the GPU box (via gpurun): kernel stats + two PMC passes of the stage-1 voxel passes (tools/bench_mask.py).
# Usage: tools/profile_mask.sh <tag> [shape]
set -u
TAG=$1; SHAPE=${2:-880x880x640}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_mask_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_mask.py $SHAPE > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/bench_mask.py $SHAPE > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/bench_mask.py $SHAPE > $OUT/bench_write.log 2>&1
tail -2 $OUT/bench_trace.log
find $OUT -name "*.csv" | head
