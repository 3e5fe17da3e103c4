#!/bin/bash
set -u
tag=${1:?tag}; out="gpurun_out/$tag"; mkdir -p "$out"; export TMPDIR=/tmp
one() { timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('  ms/step %.4f dense %.4f chain alone %s beside %s' % (d['ms_per_step'], c['dense_ms'], c.get('band_chain_ms'), c.get('band_chain_beside_dense_ms')))"; }
for rep in 1 2 3; do
  for lib in prev new; do
    if [ $lib = prev ]; then export VRG_HIP_LIB=$PWD/arterynetwork_amd/csrc/libvrg_hip_prev.so; else unset VRG_HIP_LIB; fi
    echo "$lib 512:"; one --shape 512x512x170 --steps 200
    echo "$lib slab80:"; one --shape 880x880x80 --steps 300 --force-dist
  done
done
