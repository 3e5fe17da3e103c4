#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) timing of the drop-in function with HOST numpy inputs (DESIGN.md note).

usage: tools/e2e_host.py [SHAPE] [SWEEPS] [--ref-dtypes]
--ref-dtypes: the reference's own calling convention - int64 valueMap (np.full(shape, 3), :288), float64 dataArray, C order - instead of
the friendly one (fp32 data, uint8 labels, Fortran order = the device layout).  Prints where the call's time goes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arterynetwork_amd import variationalRegionGrowing, phantoms
from arterynetwork_amd import _capi

args = [a for a in sys.argv[1:] if not a.startswith('--')]
ref = '--ref-dtypes' in sys.argv
shape = tuple(int(x) for x in (args[0] if args else '512x512x170').split('x'))
sweeps = int(args[1]) if len(args) > 1 else 200
data, vmap = phantoms.bench_volume(shape, seed=2)
if ref:
    data = np.ascontiguousarray(data, dtype=np.float64); vm = np.ascontiguousarray(vmap, dtype=np.int64)
else:
    data = np.asfortranarray(data); vm = np.asfortranarray(vmap.astype(np.uint8))
# a small call first: the library is loaded, the code objects are on the device (a fresh process pays ~0.3 s for that once)
sm = (slice(0, 64), slice(0, 64), slice(0, 32))
if (vm[sm] == 0).any():
    variationalRegionGrowing(np.ascontiguousarray(data[sm]), np.ascontiguousarray(vm[sm]), iterMax=2, maxSegmentSize=10 ** 12, quiet=True)
# where the time goes: the Session methods, timed
times = {}
def timed(name, f):
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            times[name] = times.get(name, 0.0) + time.perf_counter() - t0
    return g
S = _capi.Session
for m, label in (('set_volume', 'dataArray in (narrow + H2D + pack)'), ('set_labels', 'valueMap in (narrow + H2D + pack)'), ('init', 'init'), ('run', 'sweeps'),
                 ('labels', 'valueMap out (unpack + D2H + widen)'), ('segmented_map', 'segmentedMap out'), ('segmented', 'segmented list'), ('stats', 'stats')):
    setattr(S, m, timed(label, getattr(S, m)))
t0 = time.perf_counter()
seg, segMap, vm2 = variationalRegionGrowing(data, vm, iterMax=sweeps, maxSegmentSize=10 ** 12, maxTime=None, quiet=True)
dt = time.perf_counter() - t0
V = data.size
assert vm2 is vm and segMap.dtype == np.int64 and int(segMap.sum()) == len(seg)
print('shape %s, %d sweeps, %s host arrays in / out: %.3f s total -> %.0f Mvoxel-iter/s PCIe-inclusive (host bytes in %.2f GB, out %.2f GB)' % (
    'x'.join(map(str, shape)), sweeps, 'int64 / float64 C-order (the reference\'s own dtypes)' if ref else 'uint8 / fp32 Fortran-order', dt, V * sweeps / dt / 1e6,
    (data.nbytes + vm.nbytes) / 1e9, (vm.nbytes + segMap.nbytes) / 1e9))
for k, v in times.items():
    print('    %-40s %.3f s' % (k, v))
print('    %-40s %.3f s' % ('everything else (Session create / close, Python)', dt - sum(times.values())))
