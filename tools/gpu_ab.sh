#!/bin/bash
# every csrc/libvrg_hip_ab_*.so in turn, three rounds: the band chain alone / beside the recount on the two small shapes
set -u
export TMPDIR=/tmp
one() { timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('ms/step %.4f dense %.4f chain alone %s beside %s' % (d['ms_per_step'], c['dense_ms'], c.get('band_chain_ms'), c.get('band_chain_beside_dense_ms')))"; }
for rep in 1 2 3; do
  for lib in arterynetwork_amd/csrc/libvrg_hip_ab_*.so; do
    export VRG_HIP_LIB=$PWD/$lib
    n=$(basename $lib .so); n=${n#libvrg_hip_ab_}
    printf "%-14s 512:    " $n; one --shape 512x512x170 --steps 200
    printf "%-14s slab80: " $n; one --shape 880x880x80 --steps 300 --force-dist
  done
done
