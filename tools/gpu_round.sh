#!/bin/bash
# One GPU-box session: GPU test suite, bench lines (headline, config 2, storage16), rocprofv3 kernel stats.
# usage: tools/gpu_round.sh <tag> [quick]
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1
if [ "$2" != "quick" ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $out/pytest_gpu.log
fi
timeout 900 python bench.py > $out/bench_880.json 2> $out/bench_880.err
timeout 600 python bench.py --shape 512x512x170 --steps 200 > $out/bench_512.json 2> $out/bench_512.err
timeout 600 python bench.py --storage16 --no-cpu-baseline > $out/bench_880_s16.json 2> $out/bench_880_s16.err
# what one rank of 2 / 4 / 8 does per sweep (a 1-rank RCCL communicator on the one GPU; projection, not a scaling run)
for nz in 320 160 80; do
  timeout 600 python bench.py --force-dist --shape 880x880x$nz --no-cpu-baseline --steps 300 > $out/bench_dist1_880x880x$nz.json 2> $out/bench_dist1_880x880x$nz.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --no-cpu-baseline --steps 200 > $out/prof_bench.log 2>&1
find $out/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/prof
tail -3 $out/pytest_gpu.log 2>/dev/null; cat $out/bench_880.json $out/bench_512.json | cut -c1-600
