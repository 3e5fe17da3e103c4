#!/bin/bash
set -u
tag=${1:?tag}; out="gpurun_out/$tag"; mkdir -p "$out"; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > "$out/build_smoke.log" 2>&1; echo "build+smoke rc=$?" >> "$out/build_smoke.log"; tail -2 "$out/build_smoke.log"
timeout 1500 python -m pytest tests/test_litmus.py tests/test_float32_input.py tests/test_gpu_parity.py -m gpu -x -q -k "litmus or float32 or skip_excluded or goldens or 16bit or random" > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"; tail -5 "$out/pytest_gpu.log" | cut -c1-300
grep -E "^TICKET|^ELECTED|negative|^LITMUS" gpurun_out/litmus.log | cut -c1-220
( time timeout 900 python bench.py ) > "$out/bench_880.json" 2> "$out/bench_880.err"; tail -4 "$out/bench_880.err"; cut -c1-3000 "$out/bench_880.json"
bash tools/profile_r3.sh r03 2>&1 | tail -30
