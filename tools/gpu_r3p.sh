#!/bin/bash
set -u
tag=${1:?tag}; cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; out="gpurun_out/$tag"; rm -rf "$out"; mkdir -p "$out"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$out/pmc_icache" -- python3 bench.py --shape 512x512x170 --steps 30 --warmup 10 --no-cpu-baseline --serial 1 > "$out/bench.log" 2>&1; echo rc=$?
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/pmc_icache/*/*counter_collection.csv')[0]
acc = {}
for r in csv.DictReader(open(f)):
    k = next((n for n in ('k_recount', 'k_band', 'k_order', 'k_mark_relabel', 'k_close', 'k_gate') if n in r['Kernel_Name']), None)
    if k: acc.setdefault((k, r['Counter_Name']), []).append(float(r['Counter_Value']))
ks = sorted({k for k, _ in acc})
for k in ks:
    g = lambda c: (lambda v: sum(sorted(v)[len(v)//4:]) / max(1, len(v) - len(v)//4))(acc.get((k, c), [0]))
    print('%-16s icache req %9.0f hits %9.0f misses %8.0f dup %8.0f  ifetch %9.0f  waves %6.0f wave_cycles %10.0f' % (k, g('SQC_ICACHE_REQ'), g('SQC_ICACHE_HITS'), g('SQC_ICACHE_MISSES'), g('SQC_ICACHE_MISSES_DUPLICATE'), g('SQ_IFETCH'), g('SQ_WAVES'), g('SQ_WAVE_CYCLES')))
PY
