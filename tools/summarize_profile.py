#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag> (tools/profile_gpu.sh) into profiles/<tag>_*.csv + profiles/traffic.json."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
tag = sys.argv[1]
shape = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '880x880x640').split('x')]
src = 'gpurun_out/prof_' + tag
os.makedirs('profiles', exist_ok=True)

def short(name):
    if 'rocprim' in name:
        return 'rocprim::' + ('radix_sort' if 'radix_sort' in name else 'unique' if 'partition' in name else 'other') + ' (init only)'
    if 'at::native' in name:
        return 'torch elementwise (synthetic volume generation)'
    return name.replace('(anonymous namespace)::', '').replace('void ', '')

st = max(glob.glob(src + '/trace/*/*kernel_stats.csv'), key=os.path.getmtime)   # gpurun merges runs: newest wins
agg = {}
for r in csv.DictReader(open(st)):
    k = short(r['Name'])
    a = agg.setdefault(k, [0, 0])
    a[0] += int(r['Calls']); a[1] += int(r['TotalDurationNs'])
tot = sum(a[1] for a in agg.values())
with open('profiles/%s_kernel_stats.csv' % tag, 'w') as f:
    f.write('# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline (%s)\n' % 'x'.join(map(str, shape)))
    f.write('Name,Calls,TotalDurationNs,AverageNs,Percentage\n')
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write('"%s",%d,%d,%.1f,%.3f\n' % (k, a[0], a[1], a[1] / a[0], 100.0 * a[1] / tot))

def pmc(dirname, counter):
    f = max(glob.glob(src + '/' + dirname + '/*/*counter_collection.csv'), key=os.path.getmtime)
    vals = []
    for r in csv.DictReader(open(f)):
        if 'k_recount' in r['Kernel_Name'] and r['Counter_Name'] == counter:
            vals.append(float(r['Counter_Value']))
    return vals

fetch = pmc('pmc_fetch', 'FETCH_SIZE')
write = pmc('pmc_write', 'WRITE_SIZE')
# drop the init launch and no-op launches (first one, and the ones after the stop flag): keep the bulk
fetch = sorted(fetch)[len(fetch) // 4: -max(1, len(fetch) // 10)]
write = sorted(write)[len(write) // 4: -max(1, len(write) // 10)]
fk = sum(fetch) / len(fetch); wk = sum(write) / len(write)
V = shape[0] * shape[1] * shape[2]
PX = (shape[0] + 2 + 15) // 16 * 16
streamed = shape[2] * (shape[1] + 4) * PX          # padded interior voxels the kernel actually reads
out = {
    'shape': shape, 'n_gpus': 1, 'storage16': False, 'kernel': 'k_recount_bits', 'src_sha': bench.device_source_sha(),
    'FETCH_SIZE_KB_per_launch_raw': fk, 'WRITE_SIZE_KB_per_launch_raw': wk,
    'note': 'gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads (MI355X_MICROARCH.md, HBM); '
            'corrected fetch = 2 x raw. WRITE_SIZE is exact.',
    'hbm_bytes_per_launch': int((2 * fk + wk) * 1024),
    'algorithmic_bytes_per_launch': 6 * V, 'bytes_if_every_voxel_streamed': int(4.25 * streamed),
}
# the bytes the kernel requests by its design, counted on the device in the PMC pass itself (bench.py roofline.bytes_per_launch)
try:
    line = [l for l in open(src + '/bench_fetch.log') if l.startswith('{')][-1]
    out['design_bytes_per_launch_counted_on_device'] = json.loads(line)['roofline']['bytes_per_launch']
    out['hbm_over_design'] = round(out['hbm_bytes_per_launch'] / out['design_bytes_per_launch_counted_on_device'], 4)
except Exception:
    pass
out['src_sha_note'] = 'sha256[:16] of csrc/vrg_device.hip + vrg_items.h + vrg_types.h at profiling time; bench.py reports this traffic figure only while those sources are unchanged'
json.dump(out, open('profiles/traffic.json', 'w'), indent=1)
with open('profiles/%s_pmc.csv' % tag, 'w') as f:
    f.write('# separate passes: rocprofv3 --pmc FETCH_SIZE ... ; rocprofv3 --pmc WRITE_SIZE ... (k_recount rows, per launch)\n')
    f.write('counter,mean_KB_per_launch,launches_used\nFETCH_SIZE,%.1f,%d\nWRITE_SIZE,%.1f,%d\n' % (fk, len(fetch), wk, len(write)))
print(json.dumps(out, indent=1))
