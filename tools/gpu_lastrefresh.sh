#!/bin/bash
# After a change that touched only the 16-bit pass's launch shape: its parity tests, the traffic passes, its bench lines.
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "storage16 or skip_excluded or config3 or size_properties or slab or ranks" 2>&1 | tail -2
bash tools/profile_traffic.sh r03; bash tools/profile_traffic.sh r03_slab80 --force-dist --shape 880x880x80
mkdir -p gpurun_out/r3last2
python bench.py --storage16 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/r3last2/bench_880_s16.json
python bench.py --storage16 --shape 1024x1024x1024 --steps 200 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/r3last2/bench_1024_s16.json
python bench.py --storage16 --shape 512x512x170 --steps 200 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/r3last2/bench_512_s16.json
python bench.py --storage16 --no-cpu-baseline --force-dist --shape 880x880x80 --steps 300 2>/dev/null | grep "^{" > gpurun_out/r3last2/bench_dist1_880x880x80_s16.json
python3 -c "
import json
for n in ('880_s16','1024_s16','512_s16','dist1_880x880x80_s16'):
    d=json.load(open('gpurun_out/r3last2/bench_%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['dense_workgroups'])"
