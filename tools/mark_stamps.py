#!/usr/bin/env python3
"""Development tool: a per-workgroup timeline of k_mark_relabel (the four-launch trip's relabel kernel) on a many-flip workload.

usage: VRG_HIP_LIB=arterynetwork_amd/csrc/libvrg_hip_stamps.so python3 tools/mark_stamps.py SHAPE TUBES [samples]
(the library: tools/build_stamps.sh).  Every workgroup's thread 0 stamps: entry, state loaded, the end of each of its rounds, filing
started / ended, exit, and the hardware id of the CU it ran on.  Printed: when workgroups start and end relative to the first entry,
how long rounds and filings take, how many workgroups shared a CU."""
import ctypes as C, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from arterynetwork_amd import phantoms
from arterynetwork_amd._capi import Session

shape = tuple(int(v) for v in sys.argv[1].split('x'))
tubes = int(sys.argv[2]) if len(sys.argv) > 2 else 128
samples = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device('cuda', 0)
I, vm = phantoms.bench_volume_torch(shape, dev, levels=255, tubes=tubes)
torch.cuda.synchronize()
s = Session(shape)
s.set_option('batch', 16)
for o in os.environ.get('VRG_OPTS', '').split(','):
    if o: k, v = o.split('='); s.set_option(k, int(v))
s.set_volume_ptr(I.data_ptr(), np.float32, list(I.stride())); s.set_labels_ptr(vm.data_ptr(), np.uint8, list(vm.stride()))
s.init(2.25)
done = s.run(12, 10 ** 15, None).sweeps
NW, PER = 1024, 32
buf = (C.c_uint64 * (NW * PER))()
pct = lambda v, q: float(np.percentile(v, q)) if len(v) else float('nan')
rows = []
for smp in range(samples):
    done += s.run(done + 16, 10 ** 15, None).sweeps
    s._check(s.lib.debug_stamps_wide(s._h, buf, NW * PER))
    a = np.frombuffer(buf, dtype=np.uint64).reshape(NW, PER).astype(np.int64)
    a = a[a[:, 0] > 0]
    if not len(a):
        raise SystemExit('no stamps: not the -DVRG_STAMPS build (set VRG_HIP_LIB), or no four-launch trip ran')
    t0 = a[:, 0].min()
    us = lambda col: (col - t0) * 0.01
    entry, loaded, fl0, fl1, ex = us(a[:, 0]), us(a[:, 1]), us(a[:, 12]), us(a[:, 13]), us(a[:, 14])
    worked = a[:, 12] > 0
    rounds = a[:, 2:12]
    nround = (rounds > 0).sum(axis=1)
    # duration of each round: from the stamp before (state loaded for the first)
    prev = np.concatenate([a[:, 1:2], rounds[:, :-1]], axis=1)
    rd = np.where(rounds > 0, (rounds - prev) * 0.01, np.nan)
    hw = a[:, 15]
    xcc = (hw >> 32) & 0xf; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    cuid = xcc * 1000 + se * 100 + sh * 10 + cu
    per_cu = np.bincount(np.unique(cuid, return_inverse=True)[1])
    mid = a[:, 16] > 0
    rec = {'mid_flush_us_p50_p90 (compact kernel: the filing after two rounds)': [pct(((a[:, 17] - a[:, 16]) * 0.01)[mid], 50), pct(((a[:, 17] - a[:, 16]) * 0.01)[mid], 90), int(mid.sum())],
           'workgroups': int(len(a)), 'with_flips': int(worked.sum()), 'rounds_mean': float(nround[worked].mean()),
           'entry_us_p50_p90_max': [pct(entry, 50), pct(entry, 90), float(entry.max())],
           'exit_us_p10_p50_p90_max': [pct(ex[worked], 10), pct(ex[worked], 50), pct(ex[worked], 90), float(ex[worked].max())],
           'round_us_first_p50_p90': [pct(rd[worked, 0], 50), pct(rd[worked, 0], 90)],
           'round_us_later_p50_p90_max': [float(np.nanpercentile(rd[worked, 1:], 50)), float(np.nanpercentile(rd[worked, 1:], 90)), float(np.nanmax(rd[worked, 1:]))] if (nround[worked] > 1).any() else None,
           'flush_us_p50_p90_max': [pct((fl1 - fl0)[worked], 50), pct((fl1 - fl0)[worked], 90), float((fl1 - fl0)[worked].max())],
           'round0_phases_us_p50 (tile back -> marks back -> own stencils done -> round end)': [pct(((a[:, 16] - a[:, 1]) * 0.01)[worked], 50), pct(((a[:, 17] - a[:, 16]) * 0.01)[worked], 50), pct(((a[:, 18] - a[:, 17]) * 0.01)[worked], 50), pct(((a[:, 2] - a[:, 18]) * 0.01)[worked], 50)],
           'round2_phases_us_p50 (tile back -> marks back -> own stencils done -> round end)': [pct(((a[:, 20] - a[:, 3]) * 0.01)[a[:, 20] > 0], 50), pct(((a[:, 21] - a[:, 20]) * 0.01)[a[:, 20] > 0], 50), pct(((a[:, 22] - a[:, 21]) * 0.01)[a[:, 20] > 0], 50), pct(((a[:, 4] - a[:, 22]) * 0.01)[a[:, 20] > 0], 50)],
           'tail_after_flush_us_p50_max': [pct((ex - fl1)[worked], 50), float((ex - fl1)[worked].max())],
           'cus_used': int(len(per_cu)), 'wgs_per_cu_max': int(per_cu.max()), 'xccs': sorted(set(int(v) for v in xcc))}
    rows.append(rec)
    if smp == samples - 1:
        # late starters: workgroups that entered after the first workgroup had already left
        first_exit = ex[worked].min()
        rec['entered_after_first_exit'] = int((entry > first_exit).sum())
        order = np.argsort(entry)
        rec['entry_us_sorted_every_32nd'] = [round(float(entry[i]), 1) for i in order[::32]]
        rec['exit_us_sorted_every_32nd'] = [round(float(v), 1) for v in np.sort(ex[worked])[::32]]
print(json.dumps({'shape': sys.argv[1], 'tubes': tubes, 'samples': rows[-3:]}, indent=1))
s.close()
