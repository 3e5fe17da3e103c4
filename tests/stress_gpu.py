#!/usr/bin/env python3
"""Large adversarial parity runs (development tool): noise volumes whose narrow band is most of the volume."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import parity
from arterynetwork_amd import phantoms
from arterynetwork_amd._capi import product_lib
lib = product_lib()
for shape, seed, levels, sweeps in (((96, 96, 96), 1, 6, 12), ((128, 120, 112), 2, 10, 8), ((160, 160, 128), 3, 4, 6)):
    data, vmap = phantoms.noise_volume(shape, seed, p_seed=0.2, p_excl=0.3, levels=levels)
    t = time.time()
    res, k = parity.run_stepwise(lib, data, vmap, 2.25, None, sweeps, density_mode=1, check_hist=True, rtol=1e-8)
    print('%s levels=%d: %s sweeps compared, %s, nseg=%s band=%s  (%.0f s)' % (
        shape, levels, k, 'tie-ambiguous' if res is None else 'OK stop=%d' % res.stop_reason,
        None if res is None else res.nseg, None if res is None else res.ni + res.no, time.time() - t), flush=True)

# the same kind of volume in ONE vrg_run call (sweeps batched, dense pass trailing the band kernels by up to two sweeps)
for shape, seed, levels, sweeps, opts in (((96, 96, 96), 11, 6, 12, {'batch': 4}), ((128, 120, 112), 12, 10, 8, {'batch': 8, 'storage16': 1}),
                                          ((200, 180, 150), 13, 5, 5, {'batch': 8})):
    data, vmap = phantoms.noise_volume(shape, seed, p_seed=0.2, p_excl=0.3, levels=levels)
    t = time.time()
    res, k = parity.run_batched(lib, data, vmap, 2.25, None, sweeps, density_mode=1, rtol=1e-8, options=opts)
    print('%s levels=%d one call: %s sweeps, %s, nseg=%s band=%s  (%.0f s)' % (
        shape, levels, k, 'tie-ambiguous' if res is None else 'OK stop=%d' % res.stop_reason,
        None if res is None else res.nseg, None if res is None else res.ni + res.no, time.time() - t), flush=True)
