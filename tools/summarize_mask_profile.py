#!/usr/bin/env python3
"""Condense gpurun_out/prof_mask_<tag> (tools/profile_mask.sh) into profiles/<tag>_mask_kernel_stats.csv and
profiles/<tag>_mask_pmc.csv (development tool).   usage: tools/summarize_mask_profile.py <tag> [out-tag]"""
import csv, glob, os, re, sys
tag = sys.argv[1]; out = sys.argv[2] if len(sys.argv) > 2 else tag
src = 'gpurun_out/prof_mask_' + tag

def short(name):
    if 'rocprim' in name: return 'rocprim::exclusive_scan ' + ('(state init)' if 'init_lookback' in name or 'state_kernel' in name else '(lookback kernel)')
    m = re.search(r'(k_[a-z0-9_]+(?:<\w+>)?)', name)
    return m.group(1) if m else name[:40]

newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)      # (gpurun merges into gpurun_out/: an older run's files may still be there)
f = newest(src + '/trace/*/*kernel_stats.csv')
with open('profiles/%s_mask_kernel_stats.csv' % out, 'w') as o:
    o.write('# rocprofv3 --kernel-trace --stats -- python3 tools/bench_mask.py 880x880x640  (EDT x4, label x4, stage-1 pipeline x4 calls)\n')
    o.write('Name,Calls,TotalDurationNs,AverageNs\n')
    for r in csv.DictReader(open(f)):
        o.write('"%s",%s,%s,%s\n' % (short(r['Name']), r['Calls'], r['TotalDurationNs'], r['AverageNs']))
rows = {}
for d, c in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
    g = newest(src + '/%s/*/*counter_collection.csv' % d)
    for r in csv.DictReader(open(g)):
        if r['Counter_Name'] == c and ('k_' in r['Kernel_Name'] or 'rocprim' in r['Kernel_Name']):
            rows.setdefault(short(r['Kernel_Name']), {}).setdefault(c, []).append(float(r['Counter_Value']))
with open('profiles/%s_mask_pmc.csv' % out, 'w') as o:
    o.write('# separate passes: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python3 tools/bench_mask.py 880x880x640; mean KB per launch, raw counters\n')
    o.write('# (gfx950: FETCH_SIZE reports half of the bytes of wide coalesced reads, MI355X_MICROARCH.md; WRITE_SIZE is exact)\n')
    o.write('kernel,launches,FETCH_SIZE_KB_raw,WRITE_SIZE_KB\n')
    for k, v in sorted(rows.items()):
        fs, ws = v.get('FETCH_SIZE', [0]), v.get('WRITE_SIZE', [0])
        o.write('%s,%d,%.1f,%.1f\n' % (k, len(fs), sum(fs) / len(fs), sum(ws) / len(ws)))
print(open('profiles/%s_mask_kernel_stats.csv' % out).read()); print(open('profiles/%s_mask_pmc.csv' % out).read())
