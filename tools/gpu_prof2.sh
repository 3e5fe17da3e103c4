#!/bin/bash
# per-kernel durations of the band chain alone (dense pass off) and with it: tools/gpu_prof2.sh <tag> <shape>
tag=$1; shape=${2:-512x512x170}
out=gpurun_out/$tag
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
for d in 1 0; do
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace$d -- python3 tools/notorch_bench.py $shape 300 $d > $out/bench$d.log 2>&1
tail -1 $out/bench$d.log | cut -c1-200
f=$(find $out/trace$d -name '*kernel_stats.csv' | head -1)
python3 - <<PY
import csv
rows = list(csv.DictReader(open('$f')))
for r in rows[:7]:
    print('%-60s calls %6s avg %8.2f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
python3 tools/timeline.py $out/trace$d 150 2>/dev/null | head -12
done
rm -rf $out/trace0 $out/trace1
