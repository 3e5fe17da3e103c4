#!/bin/bash
set -u
tag=${1:?tag}; out="gpurun_out/$tag"; mkdir -p "$out"; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > "$out/build_smoke.log" 2>&1; echo "build+smoke rc=$?" >> "$out/build_smoke.log"; tail -2 "$out/build_smoke.log"
echo "== chain kernel: smoke + goldens"
VRG_CHAIN_KERNEL=1 timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
VRG_CHAIN_KERNEL=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "goldens or medium or determinism or config2_full" > "$out/pytest_chain.log" 2>&1; echo "pytest chain rc=$?" >> "$out/pytest_chain.log"; tail -6 "$out/pytest_chain.log" | cut -c1-400
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" 2> "$out/bench_$name.err" | grep '^{' > "$out/bench_$name.json"; python3 - "$out/bench_$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d['config']; r = d['roofline']
    print('%-34s value %10.1f  ms/step %.4f  dense %.4f  chain alone %s beside %s  frac %s  valid %s' % (sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], c.get('dense_ms', 0), c.get('band_chain_ms'), c.get('band_chain_beside_dense_ms'), r.get('frac'), d['valid']))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for ck in 0 1; do
  export VRG_CHAIN_KERNEL=$ck
  run 512_ck$ck --shape 512x512x170 --steps 200
  run slab80_ck$ck --shape 880x880x80 --steps 300 --force-dist
  run 880_ck$ck --steps 300
done
tail -3 "$out/bench_512_ck1.err"
