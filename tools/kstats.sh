#!/bin/bash
# Per-kernel time of one command under rocprofv3 (kernel trace + stats only), top kernels printed and the stats CSV kept.
# usage: tools/kstats.sh <tag> <program> [args...]      -> gpurun_out/kstats_<tag>/
set -u
TAG=${1:?usage: kstats.sh <tag> <program> [args...]}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT="gpurun_out/kstats_$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- "$@" > "$OUT/run.log" 2>&1
echo "rc=$?" >> "$OUT/run.log"
f=$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp "$f" "$OUT/kernel_stats.csv"; python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print('%-60s calls %6s  avg %9.2f us  total %9.3f ms  %5s %%' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, r['Percentage']))
PY
fi
find "$OUT/trace" -name '*kernel_trace.csv' -size +20M -delete
grep '^{' "$OUT/run.log" | cut -c1-400
