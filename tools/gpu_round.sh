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
for n in 2 4 8; do
  timeout 600 python bench.py --force-dist --slab-of $n --no-cpu-baseline --steps 300 > $out/bench_slab_of_$n.json 2> $out/bench_slab_of_$n.err
done
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/$out/prof -o bench -- python3 $OLDPWD/bench.py --no-cpu-baseline --steps 200 > $OLDPWD/$out/prof_bench.log 2>&1)
find $out/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out/prof -name '*.db' -delete; find $out/prof -name '*kernel_trace.csv' -size +20M -delete
tail -3 $out/pytest_gpu.log 2>/dev/null; cat $out/bench_880.json $out/bench_512.json | cut -c1-600
