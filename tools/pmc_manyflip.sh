#!/bin/bash
# Counter passes over the many-flip workload (run on the GPU box through gpurun): what bounds the four-launch chain at thousands of flips per sweep.
# usage: tools/pmc_manyflip.sh <tag> SHAPE TUBES      -> gpurun_out/pmcmf_<tag>/*.csv (one rocprofv3 --pmc run per counter set; option serial_streams:
# counter collection runs one kernel at a time)
set -u
TAG=${1:?tag}; SHAPE=${2:-512x512x170}; TUBES=${3:-128}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT="gpurun_out/pmcmf_$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
run_pmc() { n=$1; ctrs=$2; timeout 240 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/pmc_$n" -- python3 tools/manyflip.py $SHAPE $TUBES --sweeps 20 --warmup 6 --opt serial_streams=1 > "$OUT/run_$n.log" 2>&1; echo "pmc $n rc=$?"; }
run_pmc sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
run_pmc sq2 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_LDS_ATOMIC"
# (a pass with TCC_ATOMIC_sum + TCC_EA0_* aborted inside rocprofv3 and then hung for 25 minutes: SQ counters only, and every pass under its own timeout)
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/pmc_*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
rows = []
for k, cs in sorted(acc.items()):
    if not k.startswith('k_'): continue
    # (a kernel that runs as a no-op after the run's last sweep counts almost nothing: the upper half of the launches describes the real ones)
    rec = {'kernel': k}
    for c, v in sorted(cs.items()):
        v = sorted(v); rec[c] = sum(v[len(v) // 2:]) / max(1, len(v) - len(v) // 2)
    rows.append(rec)
cols = ['kernel'] + sorted({c for r in rows for c in r if c != 'kernel'})
with open(out + '/summary.csv', 'w') as fh:
    w = csv.DictWriter(fh, cols); w.writeheader()
    for r in rows: w.writerow({c: (round(r[c], 1) if isinstance(r.get(c), float) else r.get(c, '')) for c in cols})
for r in rows:
    if r['kernel'].split('<')[0] in ('k_mark_relabel', 'k_band', 'k_close', 'k_rank_wide'):
        wc = r.get('SQ_WAVE_CYCLES') or 1
        print('%-22s waves %6.0f wait_any %.2f wait_inst %.2f active_any %.2f valu %.2f lds %.2f vmem %.2f | VMEM rd %d wr %d LDS insts %d (atomic %d) | TCC req %d hit %.2f atomic %d EA rd %d wr %d | TCP stall %d atomics ret %d noret %d tagconflict %d' % (
            r['kernel'][:22], r.get('SQ_WAVES', 0), r.get('SQ_WAIT_ANY', 0) / wc, r.get('SQ_WAIT_INST_ANY', 0) / wc, r.get('SQ_ACTIVE_INST_ANY', 0) / wc, r.get('SQ_ACTIVE_INST_VALU', 0) / wc,
            r.get('SQ_ACTIVE_INST_LDS', 0) / wc, r.get('SQ_ACTIVE_INST_VMEM', 0) / wc, r.get('SQ_INSTS_VMEM_RD', 0), r.get('SQ_INSTS_VMEM_WR', 0), r.get('SQ_INSTS_LDS', 0), r.get('SQ_INSTS_LDS_ATOMIC', 0),
            r.get('TCC_REQ_sum', 0), r.get('TCC_HIT_sum', 0) / max(1, r.get('TCC_HIT_sum', 0) + r.get('TCC_MISS_sum', 0)), r.get('TCC_ATOMIC_sum', 0), r.get('TCC_EA0_RDREQ_sum', 0), r.get('TCC_EA0_WRREQ_sum', 0),
            r.get('TCP_PENDING_STALL_CYCLES_sum', 0), r.get('TCP_TCC_ATOMIC_WITH_RET_REQ_sum', 0), r.get('TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum', 0), r.get('TCP_ATOMIC_TAGCONFLICT_STALL_CYCLES_sum', 0)))
PY
find "$OUT" -name '*.csv' -size +8M -delete
