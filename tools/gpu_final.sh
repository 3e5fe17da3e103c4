#!/bin/bash
# Round-end GPU session: the driver's checks (build, smoke, pytest -m gpu), the bench lines for profiles/, the rocprofv3 passes.
set -u
tag=${1:?tag}; out="gpurun_out/$tag"; mkdir -p "$out"; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > "$out/build_smoke.log" 2>&1; echo "build+smoke rc=$?" >> "$out/build_smoke.log"; tail -2 "$out/build_smoke.log"
( time timeout 3000 python -m pytest tests -m gpu -x -q --durations=8 ) > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"; tail -16 "$out/pytest_gpu.log" | cut -c1-300
run() { name=$1; shift; timeout 900 python bench.py "$@" 2> "$out/bench_$name.err" | grep '^{' > "$out/bench_$name.json"; python3 - "$out/bench_$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d['config']; r = d['roofline']
    print('%-34s value %10.1f  ms/step %.4f  dense %.4f  chain alone %s beside %s  frac %s  valid %s' % (sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], c.get('dense_ms', 0), c.get('band_chain_ms'), c.get('band_chain_beside_dense_ms'), r.get('frac'), d['valid']))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
( time run 880 ) 2>&1 | grep -v "^$\|user\|sys"
run 880_driver --steps 20 --warmup 5
run 512 --shape 512x512x170 --steps 200
run 880_s16 --storage16 --no-cpu-baseline
run 1024_s16 --shape 1024x1024x1024 --storage16 --no-cpu-baseline --steps 200
run 880_nomask --no-brain-mask --no-cpu-baseline --steps 300
for lv in 4095 65535 0; do run 512_levels$lv --no-cpu-baseline --shape 512x512x170 --steps 100 --levels $lv; done
for nz in 320 160 80; do run dist1_880x880x$nz --force-dist --shape 880x880x$nz --no-cpu-baseline --steps 300; done
# rare-race hunt: 8 seeds x 500 stepwise parity runs of one random case each (labels, lists, histograms after every sweep)
for sd in 3 11 19 27 42 77 101 202; do timeout 600 python tools/repeat_case.py $sd 500 2>&1 | grep -v amdgpu.ids | tail -1; done > "$out/repeat_case.log" 2>&1; cat "$out/repeat_case.log"
( export VRG_HIP_LIB=$PWD/arterynetwork_amd/csrc/libvrg_hip_stamps.so
  python tools/chain_stamps.py 512x512x170 1 60 2>&1 | grep -v amdgpu.ids > "$out/chain_stamps.log"
  python tools/chain_stamps.py 512x512x170 0 60 2>&1 | grep -v amdgpu.ids >> "$out/chain_stamps.log"
  python tools/chain_stamps.py 880x880x80 0 60 2>&1 | grep -v amdgpu.ids >> "$out/chain_stamps.log" )
tail -3 "$out/chain_stamps.log"
bash tools/profile_r3.sh r03 2>&1 | tail -12
bash tools/profile_r3.sh r03_slab80 --force-dist --shape 880x880x80 2>&1 | tail -6
