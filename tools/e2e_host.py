#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) timing of the drop-in function with HOST numpy inputs (DESIGN.md note)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arterynetwork_amd import variationalRegionGrowing, phantoms
shape = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '512x512x170').split('x'))
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
data, vmap = phantoms.bench_volume(shape, seed=2)
data = np.asfortranarray(data); vm = np.asfortranarray(vmap.astype(np.uint8))
variationalRegionGrowing(data[:64, :64, :32].copy(order='F'), np.asfortranarray(vm[:64, :64, :32]), iterMax=2, maxSegmentSize=10**12, quiet=True) if (vm[:64,:64,:32]==0).any() else None
t0 = time.perf_counter()
seg, segMap, vm2 = variationalRegionGrowing(data, vm, iterMax=sweeps, maxSegmentSize=10 ** 12, maxTime=None, quiet=True)
dt = time.perf_counter() - t0
V = data.size
print('shape %s, %d sweeps, host arrays in / out: %.3f s total -> %.0f Mvoxel-iter/s PCIe-inclusive (bytes H2D %.2f GB)' % (
    'x'.join(map(str, shape)), sweeps, dt, V * sweeps / dt / 1e6, (data.nbytes + vm.nbytes) / 1e9))
