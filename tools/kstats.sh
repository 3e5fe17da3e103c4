#!/bin/bash
# Per-kernel time of one command under rocprofv3 (kernel trace + stats only), top kernels printed and the stats CSV kept.
# usage: tools/kstats.sh <tag> python3 <script.py> [args...]   (or an ELF binary)   -> gpurun_out/kstats_<tag>/
# The program right behind `rocprofv3 --` must be the interpreter or binary itself: a script with a `#!/usr/bin/env` line, `env`, `bash -c`
# or any other launcher that re-execs is an exec from a process the profiler's preloaded library has already initialised the GPU in -
# forbidden on this pool (it takes the machine down).  So: refuse anything that is not python3 / an ELF file.
set -u
TAG=${1:?usage: kstats.sh <tag> python3 <script.py> [args...]}; shift
prog=${1:?program}
case "$(basename "$prog")" in
  python3|python3.*) ;;
  *) if ! head -c 4 "$prog" 2>/dev/null | grep -q ELF; then echo "kstats.sh: '$prog' is neither python3 nor an ELF binary - run scripts as: kstats.sh <tag> python3 script.py ..." >&2; exit 2; fi ;;
esac
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
