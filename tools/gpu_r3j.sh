#!/bin/bash
set -u
tag=${1:?tag}; out="gpurun_out/$tag"; mkdir -p "$out"; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > "$out/build_smoke.log" 2>&1; echo "build+smoke rc=$?" >> "$out/build_smoke.log"; tail -2 "$out/build_smoke.log"
timeout 3000 python -m pytest tests -m gpu -x -q --durations=5 > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"; tail -10 "$out/pytest_gpu.log" | cut -c1-300
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" 2> "$out/bench_$name.err" | grep '^{' > "$out/bench_$name.json"; python3 - "$out/bench_$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d['config']; r = d['roofline']
    print('%-34s value %10.1f  ms/step %.4f  dense %.4f  chain alone %s beside %s  frac %s  wgs %s' % (sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], c.get('dense_ms', 0), c.get('band_chain_ms'), c.get('band_chain_beside_dense_ms'), r.get('frac'), c.get('dense_workgroups')))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
run 512 --shape 512x512x170 --steps 200
run slab80 --shape 880x880x80 --steps 300 --force-dist
run 880 --steps 300
export VRG_HIP_LIB=$PWD/arterynetwork_amd/csrc/libvrg_hip_stamps.so
for off in 1 0; do python tools/chain_stamps.py 512x512x170 $off 60 2>&1 | grep -v amdgpu.ids | tee -a "$out/chain_stamps.log"; done
