#!/bin/bash
# One development round trip for a change to the band chain: build + smoke, the quick GPU parity tests, the two small bench shapes twice, in-kernel stamps.
set -u
tag=${1:?tag}; out="gpurun_out/$tag"; mkdir -p "$out"; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "goldens or random or medium or config2_full or repeatable or race_free" > "$out/pytest.log" 2>&1; tail -3 "$out/pytest.log" | cut -c1-200
one() { timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('  ms/step %.4f dense %.4f chain alone %s beside %s' % (d['ms_per_step'], c['dense_ms'], c.get('band_chain_ms'), c.get('band_chain_beside_dense_ms')))"; }
for rep in 1 2; do echo "512:"; one --shape 512x512x170 --steps 200; echo "slab80:"; one --shape 880x880x80 --steps 300 --force-dist; done
export VRG_HIP_LIB=$PWD/arterynetwork_amd/csrc/libvrg_hip_stamps.so
python tools/chain_stamps.py 512x512x170 1 60 2>&1 | grep -v amdgpu.ids | tee "$out/chain_stamps.log"
