#!/bin/bash
set -u
tag=${1:?tag}; out="gpurun_out/$tag"; mkdir -p "$out"; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > "$out/build_smoke.log" 2>&1; echo "build+smoke rc=$?" >> "$out/build_smoke.log"; tail -2 "$out/build_smoke.log"
ORDERS=0,1 timeout 900 python tools/sweep_recount.py 880x880x640 0 3 768,1024 40 > "$out/sweep_880.log" 2>&1; cat "$out/sweep_880.log"
ORDERS=0,1 timeout 900 python tools/sweep_recount.py 880x880x640 0 3 768 40 --no-brain-mask > "$out/sweep_880_nomask.log" 2>&1; cat "$out/sweep_880_nomask.log"
ORDERS=0,1 timeout 600 python tools/sweep_recount.py 512x512x170 0 3 0,256 40 > "$out/sweep_512.log" 2>&1; cat "$out/sweep_512.log"
timeout 300 tools/xcdbench.bin 20000 20000 > "$out/xcdbench.log" 2>&1; grep -E "ELECTED" "$out/xcdbench.log"
