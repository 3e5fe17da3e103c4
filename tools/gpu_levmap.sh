#!/bin/bash
set -u
export TMPDIR=/tmp
one() { timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('ms/step %.4f dense %.4f chain alone %s beside %s init %s nseg_end %s valid %s' % (d['ms_per_step'], c['dense_ms'], c.get('band_chain_ms'), c.get('band_chain_beside_dense_ms'), c.get('init_seconds'), c.get('nseg_end'), d['valid']))"; }
for rep in 1 2; do for lib in arterynetwork_amd/csrc/libvrg_hip_ab_*.so; do
  export VRG_HIP_LIB=$PWD/$lib; n=$(basename $lib .so); n=${n#libvrg_hip_ab_}
  printf "%-6s 512 int 4095:  " $n; one --shape 512x512x170 --steps 200 --levels 4095 --integer-values
  printf "%-6s 512 int 65535: " $n; one --shape 512x512x170 --steps 100 --levels 65535 --integer-values
  printf "%-6s 880 int 4095:  " $n; one --steps 200 --levels 4095 --integer-values
done; done
printf "ref   512 frac 4095:  "; one --shape 512x512x170 --steps 200 --levels 4095
