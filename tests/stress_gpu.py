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
