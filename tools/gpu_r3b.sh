#!/bin/bash
# round 3, GPU session: full GPU suite + bench lines + xcdbench (placement / barrier / litmus)
set -u
tag=${1:?usage: gpu_r3b.sh <tag> [notests]}
out="gpurun_out/$tag"
mkdir -p "$out"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > "$out/build_smoke.log" 2>&1; echo "build+smoke rc=$?" >> "$out/build_smoke.log"
tail -2 "$out/build_smoke.log"
if [ "${2:-}" != "notests" ]; then
  timeout 3000 python -m pytest tests -m gpu -x -q --durations=6 > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
  tail -14 "$out/pytest_gpu.log"
fi
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" > "$out/bench_$name.json" 2> "$out/bench_$name.err"; python3 - "$out/bench_$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d['config']; r = d['roofline']
    print('%-30s value %10.1f  ms/step %.4f  dense %.4f  chain %s  frac %s  bytes %s' % (sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], c.get('dense_ms', 0), c.get('band_chain_ms'), r.get('frac'), r.get('bytes_per_launch')))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
run 880 --steps 500
run 880_driver --steps 20 --warmup 5
run 880_nomask --steps 300 --no-brain-mask
run 880_s16 --steps 300 --storage16
run 880_stream --steps 200 --skip-excluded 0
run 512 --shape 512x512x170 --steps 200
run dist1_880x880x80 --shape 880x880x80 --steps 300 --force-dist
run dist1_880x880x160 --shape 880x880x160 --steps 300 --force-dist
[ -x tools/xcdbench.bin ] && timeout 300 tools/xcdbench.bin 20000 > "$out/xcdbench.log" 2>&1; cat "$out/xcdbench.log"
