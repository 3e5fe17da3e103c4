#!/usr/bin/env python3
"""Measurement tool: ms per sweep against flips per sweep (the many-flip regime: several disjoint tubes, whole-mask seeding).

usage: python tools/manyflip.py SHAPE TUBES[,TUBES...] [--whole] [--sweeps N] [--warmup W] [--opt name=value ...]
One JSON line per workload: flips per sweep (mean / max), band size, ms per sweep, how the trips ran (fused / four-launch /
host-driven), dense pass time."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from arterynetwork_amd import phantoms
from arterynetwork_amd._capi import Session

ap = argparse.ArgumentParser()
ap.add_argument('shape'); ap.add_argument('tubes')
ap.add_argument('--whole', action='store_true'); ap.add_argument('--sweeps', type=int, default=100); ap.add_argument('--warmup', type=int, default=10)
ap.add_argument('--opt', action='append', default=[]); ap.add_argument('--levels', type=int, default=255)
ap.add_argument('--verify-every', type=int, default=1)
a = ap.parse_args()
shape = tuple(int(v) for v in a.shape.split('x'))
dev = torch.device('cuda', 0)
for tubes in [int(t) for t in a.tubes.split(',')]:
    I, vm = phantoms.bench_volume_torch(shape, dev, levels=a.levels, tubes=tubes, seed_mode='whole' if a.whole else 'planes')
    torch.cuda.synchronize()
    s = Session(shape)
    s.set_option('batch', 64); s.set_option('events', 4)
    if a.verify_every != 1: s.set_option('verify_every', a.verify_every)
    for o in a.opt:
        k, v = o.split('='); s.set_option(k, int(v))
    s.set_volume_ptr(I.data_ptr(), np.float32, list(I.stride())); s.set_labels_ptr(vm.data_ptr(), np.uint8, list(vm.stride()))
    t0 = time.perf_counter(); s.init(2.25); t_init = time.perf_counter() - t0
    r0 = s.run(a.warmup, 10 ** 15, None)
    st0 = s.stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = s.run(a.warmup + a.sweeps, 10 ** 15, None)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    tr = s.trace(); st = s.stats()
    nfl = tr['nflip'][a.warmup + 1:]
    out = {'shape': a.shape, 'tubes': tubes, 'seed_mode': 'whole' if a.whole else 'planes', 'sweeps': int(r.sweeps), 'stop': int(r.stop_reason),
           'warmup_sweeps': int(r0.sweeps), 'ms_per_sweep': round(dt / max(1, r.sweeps) * 1e3, 4), 'flips_mean': round(float(nfl.mean()), 1) if len(nfl) else None,
           'flips_max': int(nfl.max()) if len(nfl) else None, 'flips_first': [int(v) for v in tr['nflip'][1:6]], 'band_end': int(tr['ni'][-1] + tr['no'][-1]), 'nseg_end': int(tr['nseg'][-1]),
           'dense_ms': round(r.sweep_kernel_ms / max(1, r.sweep_launches), 4), 'init_s': round(t_init, 3),
           'fused_trips': st['fused_trips'] - st0['fused_trips'], 'host_driven_trips': st['host_driven_trips'] - st0['host_driven_trips'],
           'bail_fuse': st['bail_fuse'] - st0['bail_fuse'], 'bail_flips': st['bail_flips'] - st0['bail_flips'], 'grow_pool': st['grow_pool'], 'grow_marks': st['grow_marks'],
           'ties': int(r.ties), 'verify_every': a.verify_every, 'opts': a.opt}
    print(json.dumps(out), flush=True)
    s.close()
    del I, vm
    torch.cuda.empty_cache()
