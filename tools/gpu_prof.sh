#!/bin/bash
# per-kernel durations of a bench command: tools/gpu_prof.sh <tag> [bench args...]
set -u
tag=${1:?usage: gpu_prof.sh <tag> [bench args...]}; shift
out="gpurun_out/$tag"
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline "$@" > $out/bench.log 2>&1
grep '^{' $out/bench.log | cut -c1-300
f=$(find $out/trace -name '*kernel_stats.csv' | head -1)
cp $f $out/kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open('$out/kernel_stats.csv')))
for r in rows[:14]:
    print('%-70s calls %6s avg %10.1f ns  total %6.2f %%' % (r['Name'][:70], r['Calls'], float(r['AverageNs']), float(r['Percentage'])))
PY
python3 tools/timeline.py $out/trace 100 2>/dev/null | head -20
find $out/trace -name '*kernel_trace.csv' -size +30M -delete
