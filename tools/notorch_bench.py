#!/usr/bin/env python3
"""ms/sweep WITHOUT PyTorch in the process (system ROCm runtime) - compare with bench.py (PyTorch's bundled runtime)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arterynetwork_amd import phantoms
from arterynetwork_amd._capi import Session
shape = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '512x512x170').split('x'))
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
graph = int(sys.argv[3]) if len(sys.argv) > 3 else 0
data, vmap = phantoms.bench_volume(shape, seed=3, radius=4.0)
data = np.asfortranarray(data); vm = np.asfortranarray(vmap.astype(np.uint8))
s = Session(shape)
s.set_option('batch', 64); s.set_option('events', 0 if graph else 1); s.set_option('graph', graph)
s.set_volume(data); s.set_labels(vm); s.init(2.25)
s.run(20, 10 ** 12, None)
r = s.run(20 + sweeps, 10 ** 12, None)
tr = s.trace()
print('no-torch %s graph=%d: %.4f ms/sweep (%d sweeps, kernel %.4f ms), flips/sweep %.0f, nseg %d' % (
    'x'.join(map(str, shape)), graph, r.seconds / max(1, r.sweeps) * 1e3, r.sweeps,
    r.sweep_kernel_ms / max(1, r.sweep_launches), tr['nflip'][21:].mean(), tr['nseg'][-1]))
