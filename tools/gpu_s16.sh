#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "storage16 or skip_excluded or goldens" 2>&1 | tail -2
one() { timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print('ms/step %.4f dense %.4f frac %.3f %s' % (d['ms_per_step'], c['dense_ms'], r['frac'], r['kernel']))"; }
for rep in 1 2; do for p in 0 1; do
  printf "pipe %d 880 s16:   " $p; one --storage16 --dense-pipe $p --steps 200
  printf "pipe %d 1024 s16:  " $p; one --storage16 --dense-pipe $p --shape 1024x1024x1024 --steps 100
  printf "pipe %d 512 s16:   " $p; one --storage16 --dense-pipe $p --shape 512x512x170 --steps 200
done; done
