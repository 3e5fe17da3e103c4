#!/bin/bash
set -u
export TMPDIR=/tmp
one() { timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('ms/step %.4f dense %.4f chain alone %s beside %s wgs %s' % (d['ms_per_step'], c['dense_ms'], c.get('band_chain_ms'), c.get('band_chain_beside_dense_ms'), c.get('dense_workgroups')))"; }
for rep in 1 2; do
for wb in 0 192 256 320; do printf "512 blocks %3d: " $wb; one --shape 512x512x170 --steps 200 --sweep-blocks $wb; done
for wb in 0 256 320; do printf "nz 80 blocks %3d: " $wb; one --force-dist --shape 880x880x80 --steps 300 --sweep-blocks $wb; done
for wb in 320 384 448 512; do printf "nz 160 blocks %3d: " $wb; one --force-dist --shape 880x880x160 --steps 300 --sweep-blocks $wb; done
for wb in 512 640 768; do printf "nz 320 blocks %3d: " $wb; one --force-dist --shape 880x880x320 --steps 300 --sweep-blocks $wb; done
done
