#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "storage16 or skip_excluded or config3 or size_properties or dense_pipe or goldens" 2>&1 | tail -2
timeout 900 python tests/full_size_check.py --config5 2>&1 | tail -3
one() { timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print('ms/step %.4f dense %.4f frac %.3f %s wgs %s' % (d['ms_per_step'], c['dense_ms'], r['frac'], r['kernel'], c.get('dense_workgroups')))"; }
for rep in 1 2; do
  printf "880 s16:  "; one --storage16 --steps 200
  printf "1024 s16: "; one --storage16 --shape 1024x1024x1024 --steps 100
  printf "512 s16:  "; one --storage16 --shape 512x512x170 --steps 200
  printf "512 s16 4095 levels:  "; one --storage16 --shape 512x512x170 --steps 200 --levels 4095
done
