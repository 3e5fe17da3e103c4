#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag> (tools/profile_bench.sh) into profiles/<name>_kernel_stats.csv, profiles/<name>_pmc.csv and
an entry of profiles/traffic.json.   usage: summarize_profile.py <tag> [name] [shape] [planes] [n_gpus] [storage16]"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
tag = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else tag
src = 'gpurun_out/prof_' + tag


def profiled_shape():
    """The volume the profiled bench lines were run on (their `metric` names it): the default when no shape is given, so that
    a call with the tag alone can never file one volume's bytes under another's shape."""
    import re
    for log in ('bench_fetch.log', 'bench.json', 'bench_trace.log'):
        try:
            line = [l for l in open(os.path.join(src, log)) if l.startswith('{')][-1]
            return re.search(r'(\d+x\d+x\d+) volume', json.loads(line)['metric']).group(1)
        except Exception:
            continue
    return '880x880x640'


shape = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else profiled_shape()).split('x')]
planes = int(sys.argv[4]) if len(sys.argv) > 4 else shape[2]
n_gpus = int(sys.argv[5]) if len(sys.argv) > 5 else 1
storage16 = bool(int(sys.argv[6])) if len(sys.argv) > 6 else False
os.makedirs('profiles', exist_ok=True)
KERNELS = ('k_recount_pipe', 'k_recount_bits', 'k_band', 'k_sweep', 'k_memo', 'k_order', 'k_mark_relabel', 'k_close', 'k_gate')


def short(n):
    if 'rocprim' in n:
        return 'rocprim (init only)'
    if 'at::native' in n or 'elementwise' in n:
        return 'torch elementwise (synthetic volume generation)'
    return n.replace('(anonymous namespace)::', '').replace('void ', '')


st = sorted(glob.glob(src + '/trace/*/*kernel_stats.csv'), key=os.path.getmtime)
if st:
    agg = {}
    for r in csv.DictReader(open(st[-1])):
        a = agg.setdefault(short(r['Name']), [0, 0])
        a[0] += int(r['Calls']); a[1] += int(r['TotalDurationNs'])
    tot = sum(a[1] for a in agg.values())
    with open('profiles/%s_kernel_stats.csv' % name, 'w') as f:
        f.write('# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline (%s)\n' % 'x'.join(map(str, shape)))
        f.write('Name,Calls,TotalDurationNs,AverageNs,Percentage\n')
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write('"%s",%d,%d,%.1f,%.3f\n' % (k, a[0], a[1], a[1] / a[0], 100.0 * a[1] / tot))

# every PMC pass: mean per launch and kernel (the bulk of the launches: first quarter and last tenth dropped)
rows = []
per = {}
for d in sorted(glob.glob(src + '/pmc_*')):
    files = sorted(glob.glob(d + '/*/*counter_collection.csv'), key=os.path.getmtime)
    if not files:
        continue
    vals = {}
    for r in csv.DictReader(open(files[-1])):
        kn = next((k for k in KERNELS if k in r['Kernel_Name']), None)
        if kn:
            vals.setdefault((kn, r['Counter_Name']), []).append(float(r['Counter_Value']))
    for (kn, cn), v in sorted(vals.items()):
        v = sorted(v)
        core = v[len(v) // 4: len(v) - max(1, len(v) // 10)] or v
        m = sum(core) / len(core)
        rows.append((os.path.basename(d)[4:], kn, cn, m, len(core)))
        per[(kn, cn)] = m
with open('profiles/%s_pmc.csv' % name, 'w') as f:
    f.write('# rocprofv3 --pmc <set> -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --serial 1 ...: one run per counter set; mean per launch\n')
    f.write('pass,kernel,counter,mean_per_launch,launches_used\n')
    for r in rows:
        f.write('%s,%s,%s,%.1f,%d\n' % r)

# the dense pass of the sweeps: the recount kernel with the most launches (the other one runs once, at init)
used = {kn: max((r[4] for r in rows if r[1] == kn and r[2] == 'FETCH_SIZE'), default=0) for kn in ('k_recount_pipe', 'k_recount_bits')}
dk = max(used, key=used.get)
fk, wk = per.get((dk, 'FETCH_SIZE')), per.get((dk, 'WRITE_SIZE'))
if fk is not None and wk is not None:
    entry = {'shape': shape, 'planes': planes, 'n_gpus': n_gpus, 'storage16': storage16, 'kernel': dk, 'launches_averaged': used[dk], 'src_sha': bench.device_source_sha(),
             'FETCH_SIZE_KB_per_launch_raw': fk, 'WRITE_SIZE_KB_per_launch_raw': wk,
             'note': 'gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads (MI355X_MICROARCH.md, HBM); corrected fetch = 2 x raw. WRITE_SIZE is exact.',
             'hbm_bytes_per_launch': int((2 * fk + wk) * 1024)}
    try:
        line = [l for l in open(src + '/bench_fetch.log') if l.startswith('{')][-1]
        entry['design_bytes_per_launch_counted_on_device'] = json.loads(line)['roofline']['bytes_per_launch']
        entry['hbm_over_design'] = round(entry['hbm_bytes_per_launch'] / entry['design_bytes_per_launch_counted_on_device'], 4)
    except Exception:
        pass
    path = 'profiles/traffic.json'
    try:
        t = json.load(open(path))
        entries = t.get('entries', [])
    except Exception:
        entries = []
    entries = [e for e in entries if not (e.get('shape') == shape and e.get('planes', e.get('shape', [0, 0, 0])[2]) == planes and e.get('n_gpus', 1) == n_gpus and bool(e.get('storage16')) == storage16)]
    entries.append(entry)
    json.dump({'src_sha_note': 'sha256[:16] of csrc/vrg_device.hip + vrg_items.h + vrg_types.h at profiling time; bench.py reports a traffic figure only while those sources are unchanged',
               'entries': entries}, open(path, 'w'), indent=1)
    print(json.dumps(entry, indent=1))
# derived: what the SQ counters say about the recount
g = lambda k, c: per.get((k, c))
for k in KERNELS:
    wc, busy, valu, wait, act = g(k, 'SQ_WAVE_CYCLES'), g(k, 'SQ_BUSY_CYCLES'), g(k, 'SQ_INSTS_VALU'), g(k, 'SQ_WAIT_ANY'), g(k, 'SQ_ACTIVE_INST_ANY')
    if wc:
        print('%-16s waves %s  VALU insts %s  wait_any/wave_cycles %.2f  active_inst/wave_cycles %.2f  valu_active/wave_cycles %s  L2 hit %s' % (
            k, g(k, 'SQ_WAVES'), valu, (wait or 0) / wc, (act or 0) / wc,
            ('%.2f' % (g(k, 'SQ_ACTIVE_INST_VALU') / wc)) if g(k, 'SQ_ACTIVE_INST_VALU') else None,
            ('%.3f' % (g(k, 'TCC_HIT_sum') / (g(k, 'TCC_HIT_sum') + g(k, 'TCC_MISS_sum')))) if g(k, 'TCC_HIT_sum') is not None and g(k, 'TCC_MISS_sum') else None))
