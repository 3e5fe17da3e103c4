#!/bin/bash
# round 3, first GPU session: full GPU suite (new tests included) + baseline timelines of the band chain
set -u
tag=${1:?usage: gpu_r3a.sh <tag>}
out="gpurun_out/$tag"
mkdir -p "$out"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > "$out/build_smoke.log" 2>&1; echo "build+smoke rc=$?" >> "$out/build_smoke.log"
tail -2 "$out/build_smoke.log"
timeout 3000 python -m pytest tests -m gpu -x -q --durations=12 > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
tail -22 "$out/pytest_gpu.log"
for cfg in "512x512x170 --steps 200" "880x880x80 --steps 300 --force-dist"; do
  set -- $cfg
  shp=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$shp" -- python3 bench.py --no-cpu-baseline --shape $shp "$@" > "$out/bench_$shp.log" 2>&1
  grep '^{' "$out/bench_$shp.log" | cut -c1-900
  python3 tools/timeline.py "$out/trace_$shp" 100 2>/dev/null | head -14
  f=$(find "$out/trace_$shp" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -12 "$f" > "$out/kernel_stats_$shp.csv"
  find "$out/trace_$shp" -name '*kernel_trace.csv' -size +20M -delete
done
