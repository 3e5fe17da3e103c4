#!/bin/bash
# SQ counters of the EDT kernels (tools/edt_bench.py, one repetition) - development tool, run on the GPU box.
set -u
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d "$out/pmc$i" -- python3 tools/edt_bench.py ${EDT_SHAPE:-880x880x640} 1 > "$out/pmc$i.log" 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/pmc*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'k_(edt|transpose)\w*(<[^>(]*>)?', r['Kernel_Name'])
        if m: acc[m.group(0)][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    # calls alternate ellipsoid (first half) / tube (second half): report the ellipsoid half
    print(k, ' '.join('%s=%.4g' % (n, sum(v[:len(v) // 2]) / max(1, len(v) // 2)) for n, v in sorted(c.items())))
PY
