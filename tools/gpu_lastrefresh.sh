#!/bin/bash
# Closing refresh after a change that left the dense kernels as they were: the whole GPU suite, the two traffic passes
# (profiles/traffic.json carries the source hash), the headline bench lines.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3last3
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r3last3/pytest.log 2>&1; grep "passed\|failed" gpurun_out/r3last3/pytest.log
bash tools/profile_traffic.sh r03; bash tools/profile_traffic.sh r03_slab80 --force-dist --shape 880x880x80
python bench.py 2>/dev/null | grep "^{" > gpurun_out/r3last3/bench_880.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep "^{" > gpurun_out/r3last3/bench_880_driver_args.json
python bench.py --shape 512x512x170 --steps 200 2>/dev/null | grep "^{" > gpurun_out/r3last3/bench_512.json
python bench.py --shape 512x512x170 --steps 200 --levels 4095 --integer-values --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/r3last3/bench_512_levels4095_integer.json
python bench.py --force-dist --shape 880x880x80 --steps 300 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/r3last3/bench_dist1_880x880x80.json
python3 -c "
import json
for n in ('880','880_driver_args','512','512_levels4095_integer','dist1_880x880x80'):
    d=json.load(open('gpurun_out/r3last3/bench_%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['valid'])"
