#!/bin/bash
# quick GPU check of a new build: smoke, a few parity tests, short bench lines
tag=${1:-q}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log
tail -3 $out/smoke.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q ${PYTEST_ARGS:-} > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $out/pytest_gpu.log
tail -25 $out/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline --steps 300 > $out/bench_880.json 2> $out/bench_880.err; tail -2 $out/bench_880.err; cut -c1-1500 $out/bench_880.json
timeout 600 python bench.py --no-cpu-baseline --shape 512x512x170 --steps 200 > $out/bench_512.json 2> $out/bench_512.err; tail -2 $out/bench_512.err; cut -c1-1500 $out/bench_512.json
