#!/bin/bash
set -u
tag=${1:?tag}; out="gpurun_out/$tag"; mkdir -p "$out"; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > "$out/build_smoke.log" 2>&1; echo "build+smoke rc=$?" >> "$out/build_smoke.log"; tail -2 "$out/build_smoke.log"
timeout 3000 python -m pytest tests -m gpu -x -q --durations=5 > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"; tail -12 "$out/pytest_gpu.log"
timeout 900 python tools/sweep_recount.py 880x880x640 0 3 0,640,768,896,1024 40 > "$out/sweep_880.log" 2>&1; cat "$out/sweep_880.log"
timeout 600 python tools/sweep_recount.py 512x512x170 0 3 0,384,512,768,1024 40 > "$out/sweep_512.log" 2>&1; cat "$out/sweep_512.log"
timeout 600 python tools/sweep_recount.py 880x880x80 0 3 0,384,512,768,1024 40 > "$out/sweep_slab80.log" 2>&1; cat "$out/sweep_slab80.log"
timeout 600 python tools/sweep_recount.py 880x880x640 0 3 0,1024,2048 40 --storage16 > "$out/sweep_880_s16.log" 2>&1; cat "$out/sweep_880_s16.log"
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
run 880_stream --steps 200 --skip-excluded 0
run 512 --shape 512x512x170 --steps 200
run dist1_880x880x80 --shape 880x880x80 --steps 300 --force-dist
tail -3 "$out/bench_dist1_880x880x80.err"
