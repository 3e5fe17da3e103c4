#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/edtab; rm -rf $OUT; mkdir -p $OUT
for lib in arterynetwork_amd/csrc/libvrg_hip_ab_*.so; do
  export VRG_HIP_LIB=$PWD/$lib
  n=$(basename $lib .so); n=${n#libvrg_hip_ab_}
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n/trace -- python3 tools/edt_only.py > $OUT/$n.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$n/fetch -- python3 tools/edt_only.py >> $OUT/$n.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$n/write -- python3 tools/edt_only.py >> $OUT/$n.log 2>&1
  python3 - $OUT/$n $n <<'PY'
import csv, glob, sys
d, n = sys.argv[1], sys.argv[2]
t = [r for r in csv.DictReader(open(glob.glob(d + '/trace/*/*kernel_stats.csv')[0])) if 'k_edt_envelope' in r['Name']][0]
def pmc(sub, c):
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(glob.glob(d + '/%s/*/*counter_collection.csv' % sub)[0])) if r['Counter_Name'] == c and 'k_edt_envelope' in r['Kernel_Name']]
    return sum(v) / len(v) / 1e6
f, w = pmc('fetch', 'FETCH_SIZE'), pmc('write', 'WRITE_SIZE')
print('%-10s envelope %.3f ms  fetch raw %.3f GB (x2 = %.3f)  write %.3f GB  total %.2f GB' % (n, float(t['AverageNs']) / 1e6, f, 2 * f, w, 2 * f + w))
PY
done
