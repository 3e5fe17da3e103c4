#!/bin/bash
set -u
tag=${1:?tag}; out="gpurun_out/$tag"; mkdir -p "$out"; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > "$out/build_smoke.log" 2>&1; echo "build+smoke rc=$?" >> "$out/build_smoke.log"; tail -2 "$out/build_smoke.log"
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "goldens or skip_excluded or 16bit or zslabs or float64 or config2_full or handle_reuse or medium" > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"; tail -6 "$out/pytest_gpu.log"
timeout 900 python tools/sweep_recount.py 880x880x640 0,8,14,20,28 3,6 768 > "$out/sweep_880.log" 2>&1; cat "$out/sweep_880.log"
timeout 600 python tools/sweep_recount.py 880x880x640 14 3,6 512,1024,1280 > "$out/sweep_880_blocks.log" 2>&1; cat "$out/sweep_880_blocks.log"
timeout 600 python tools/sweep_recount.py 880x880x640 14 3 768 --no-brain-mask > "$out/sweep_880_nomask.log" 2>&1; cat "$out/sweep_880_nomask.log"
timeout 600 python tools/sweep_recount.py 512x512x170 8,14,20 3,6 0 > "$out/sweep_512.log" 2>&1; cat "$out/sweep_512.log"
timeout 600 python tools/sweep_recount.py 880x880x80 8,14,20 3,6 0 > "$out/sweep_slab80.log" 2>&1; cat "$out/sweep_slab80.log"
timeout 600 python bench.py --no-cpu-baseline --steps 200 --skip-excluded 0 2>/dev/null | cut -c1-400
