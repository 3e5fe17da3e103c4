#!/bin/bash
set -u
export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('ms/step %.4f dense %.4f value %.0f' % (d['ms_per_step'], c['dense_ms'], d['value']))"; done
