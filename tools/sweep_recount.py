#!/usr/bin/env python3
"""Development tool: time the dense recount for a grid of (dense_cost_floor, dense_units, sweep_blocks) in ONE process.
usage: tools/sweep_recount.py SHAPE floors units blocks [sweeps] [--no-brain-mask] [--storage16]   (comma-separated lists)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from arterynetwork_amd import phantoms
from arterynetwork_amd._capi import Session

shape = tuple(int(v) for v in sys.argv[1].split('x'))
floors = [int(v) for v in sys.argv[2].split(',')]
units = [int(v) for v in sys.argv[3].split(',')]
blocks = [int(v) for v in sys.argv[4].split(',')]
sweeps = int(sys.argv[5]) if len(sys.argv) > 5 and sys.argv[5].isdigit() else 60
orders = [int(v) for v in os.environ.get('ORDERS', '0').split(',')]
dev = torch.device('cuda', 0)
I, vm = phantoms.bench_volume_torch(shape, dev, brain_mask='--no-brain-mask' not in sys.argv)
torch.cuda.synchronize()
s = Session(shape)
if '--storage16' in sys.argv:
    s.set_option('storage16', 1)
s.set_option('events', 1); s.set_option('batch', 64)
s.set_volume_ptr(I.data_ptr(), np.float32, list(I.stride()))
s.set_labels_ptr(vm.data_ptr(), np.uint8, list(vm.stride()))
s.init(2.25)
done = 0
r = s.run(20, 10 ** 15, None); done += r.sweeps
for od in orders:
  for fl in floors:
    for un in units:
        for bl in blocks:
            s.set_option('sweep_blocks', bl); s.set_option('dense_pipe', od)
            r = s.run(done + 10, 10 ** 15, None); done += r.sweeps           # (re-split + warm-up)
            t0 = time.perf_counter()
            r = s.run(done + sweeps, 10 ** 15, None); done += r.sweeps
            dt = time.perf_counter() - t0
            db = s.stats()['dense_bytes']
            k = r.sweep_kernel_ms / max(1, r.sweep_launches)
            if not k:
                print('run stopped', r.stop_reason); break
            print('order %d' % od, 'floor %2d units %d blocks %4d: dense %.4f ms  step %.4f ms  %.0f GB/s (%.3f of peak) bytes %d' % (fl, un, bl, k, dt / max(1, r.sweeps) * 1e3, db / k / 1e6, db / k / 1e6 / 8000, db), flush=True)
s.close()
