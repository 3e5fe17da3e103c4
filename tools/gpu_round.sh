#!/bin/bash
# One GPU-box session: GPU test suite, bench lines (headline, config 2, storage16, level-table regimes, slab sizes),
# rocprofv3 kernel stats + PMC passes.   usage: tools/gpu_round.sh <tag> [notests]
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $out/build_smoke.log 2>&1; echo "build+smoke rc=$?" >> $out/build_smoke.log
if [ "$2" != "notests" ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $out/pytest_gpu.log
fi
timeout 900 python bench.py > $out/bench_880.json 2> $out/bench_880.err
timeout 600 python bench.py --shape 512x512x170 --steps 200 > $out/bench_512.json 2> $out/bench_512.err
timeout 600 python bench.py --storage16 --no-cpu-baseline > $out/bench_880_s16.json 2> $out/bench_880_s16.err
timeout 600 python bench.py --shape 1024x1024x1024 --storage16 --no-cpu-baseline --steps 200 > $out/bench_1024_s16.json 2> $out/bench_1024_s16.err
for lv in 4095 65535 0; do
  timeout 600 python bench.py --no-cpu-baseline --shape 512x512x170 --steps 100 --levels $lv > $out/bench_512_levels$lv.json 2> $out/bench_512_levels$lv.err
done
# what one rank of 2 / 4 / 8 does per sweep (a 1-rank RCCL communicator on the one GPU; projection, not a scaling run)
for nz in 320 160 80; do
  timeout 600 python bench.py --force-dist --shape 880x880x$nz --no-cpu-baseline --steps 300 2> $out/bench_dist1_880x880x$nz.err | grep '^{' > $out/bench_dist1_880x880x$nz.json
done
bash tools/profile_gpu.sh $tag > $out/profile.log 2>&1
grep -E "passed|failed|rc=" $out/pytest_gpu.log $out/build_smoke.log | tail -4
for f in $out/bench_*.json; do python3 - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d['config']; r = d['roofline']
    print('%-44s value %10.1f  ms/step %.4f  dense %.4f  chain %s  frac %s' % (sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], c.get('dense_ms', 0), c.get('band_chain_ms'), r.get('frac')))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
