#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/edtpmc; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/edt_only.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
sys.argv = ['x']
import importlib.util
spec = importlib.util.spec_from_file_location('bm', 'tools/bench_mask.py')
src = open('tools/bench_mask.py').read().split("for shape in")[0]
exec(src)
from arterynetwork_amd import generateVesselVolume as G
brain, ves = vol((880, 880, 640))
for _ in range(2): G.distance_transform_edt(brain)
PY
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq1 -- python3 /tmp/edt_only.py > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq2 -- python3 /tmp/edt_only.py > $OUT/sq2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ('sq1', 'sq2'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('gpurun_out/edtpmc/%s/*/*counter_collection.csv' % d):
        for r in csv.DictReader(open(f)):
            if 'k_edt_envelope' in r['Kernel_Name']:
                acc[r['Counter_Name']]['v'].append(float(r['Counter_Value']))
    for k, v in acc.items():
        print(d, k, 'mean per launch %.4g over %d' % (sum(v['v']) / len(v['v']), len(v['v'])))
PY
tail -3 $OUT/sq2.log
