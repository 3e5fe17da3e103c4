#!/bin/bash
set -u
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "goldens or random or layouts or config1 or kats or skip_excluded or float64" 2>&1 | tail -2
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('init_seconds', c['init_seconds'], 'reinit', c['reinit_seconds'], 'ms/step', d['ms_per_step'])"; done
cd /tmp; cd $GRAFT_REPO_ROOT; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/initprof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1; grep -h "k_init_voxel\|k_cls_build\|k_hist_lds\|rocprim" gpurun_out/initprof/*/*kernel_stats.csv | cut -c1-60,100-200 | sed 's/.*(VrgCtx)"//' | head
grep -h "k_init_voxel\|k_cls_build" gpurun_out/initprof/*/*kernel_stats.csv | awk -F, '{print $1, $(NF-6), $(NF-5), $(NF-4)}' | cut -c1-120
