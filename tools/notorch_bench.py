#!/usr/bin/env python3
"""ms/sweep WITHOUT PyTorch in the process (system ROCm runtime) - compare with bench.py (PyTorch's bundled runtime).
args: [shape] [sweeps] [dense_off 0/1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arterynetwork_amd import phantoms
from arterynetwork_amd._capi import Session
shape = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '512x512x170').split('x'))
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dense_off = int(sys.argv[3]) if len(sys.argv) > 3 else 0
data, vmap = phantoms.bench_volume(shape, seed=3, radius=4.0)
data = np.asfortranarray(data); vm = np.asfortranarray(vmap.astype(np.uint8))
s = Session(shape)
s.set_option('batch', 64); s.set_option('events', 0 if dense_off else 1)
s.set_volume(data); s.set_labels(vm); s.init(2.25)
s.run(20, 10 ** 12, None)
if dense_off:
    s.set_option('dense_off', 1)
r = s.run(20 + sweeps, 10 ** 12, None)
print('no-torch %s dense_off=%d: %.4f ms/sweep (%d sweeps, dense kernel %.4f ms), stats %s' % (
    'x'.join(map(str, shape)), dense_off, r.seconds / max(1, r.sweeps) * 1e3, r.sweeps,
    r.sweep_kernel_ms / max(1, r.sweep_launches), s.stats()))
