#!/bin/bash
# Kernel times of tools/edt_bench.py per mask (rocprofv3 kernel trace) - development tool, run on the GPU box.
#   tools/edt_prof.sh <outdir> [lib]     -> prints the mean duration of every EDT kernel, ellipsoid and tube separately
set -u
out=$1; lib=${2:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
[ -n "$lib" ] && export VRG_HIP_LIB="$lib"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -- python3 tools/edt_bench.py ${EDT_SHAPE:-880x880x640} 3 > "$out/bench.log" 2>&1
grep -E "ellipsoid|tube" "$out/bench.log"
python3 - "$out" <<'PY'
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + '/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_edt' in r['Kernel_Name'] or 'k_transpose' in r['Kernel_Name']]
per = len(rows) // 2                    # first half: ellipsoid, second half: tube (warm-up call + reps each)
for name, part in (('ellipsoid', rows[:per]), ('tube', rows[per:])):
    acc = collections.OrderedDict()
    seq = collections.Counter()
    for i, r in enumerate(part):
        k = re.search(r'k_\w+(<[^>(]*>)?', r['Kernel_Name']).group(0)
        if 'envelope' in k:
            k += ' pass %d' % (1 + seq[i // 5] ); seq[i // 5] += 1
        acc.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
    print(name, ' '.join('%s %.3f' % (k, sum(v) / len(v)) for k, v in acc.items()), ' sum %.2f ms' % sum(sum(v) / len(v) for v in acc.values()))
PY
