#!/bin/bash
set -u
tag=${1:?tag}; out="gpurun_out/$tag"; mkdir -p "$out"; export TMPDIR=/tmp
export VRG_CHAIN_KERNEL=1
for m in 8 16 64 128; do
  export VRG_CHAIN_MEMBERS=$m
  timeout 300 python bench.py --no-cpu-baseline --shape 512x512x170 --steps 200 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('members $m: ms/step %.4f chain alone %s beside %s valid %s' % (d['ms_per_step'], c.get('band_chain_ms'), c.get('band_chain_beside_dense_ms'), d['valid']))"
done
export VRG_HIP_LIB=$PWD/arterynetwork_amd/csrc/libvrg_hip_stamps.so
VRG_CHAIN_MEMBERS=64 python tools/chain_stamps.py 512x512x170 1 40 2>&1 | grep -v amdgpu.ids | tee -a "$out/chain_stamps_ck1_m64.log"
