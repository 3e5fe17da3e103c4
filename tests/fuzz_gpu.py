#!/usr/bin/env python3
"""Large randomized parity run of the HIP path against the oracle (development tool; the committed tests run a subset)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import parity
from test_hostmodel import random_case
from arterynetwork_amd._capi import product_lib
lib = product_lib()
n_small, n_med = int(sys.argv[1]), int(sys.argv[2])
off = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # shifts every seed range: other cases
t = time.time(); sweeps = 0; amb = 0; fails = 0
for sd in range(100000 + off, 100000 + off + n_small):
    I, vm, H, variant, dmode = random_case(sd)
    try:
        res, k = parity.run_stepwise(lib, I, vm, H, None, 40, density_mode=dmode, check_hist=True, options={'sweep_variant': variant})
        sweeps += k; amb += res is None
    except Exception as e:
        fails += 1; print('FAIL small', sd, type(e).__name__, str(e)[:200].replace('\n', ' '))
for sd in range(200000 + off, 200000 + off + n_med):
    I, vm, H, variant, dmode = random_case(sd, 10, 40)
    try:
        res, k = parity.run_stepwise(lib, I, vm, H, None, 30, density_mode=1, check_hist=True, options={'sweep_variant': variant})
        sweeps += k; amb += res is None
    except Exception as e:
        fails += 1; print('FAIL medium', sd, type(e).__name__, str(e)[:200].replace('\n', ' '))
for sd in range(300000 + off, 300000 + off + n_small):
    I, vm, H, variant, dmode = random_case(sd)
    try:
        res, k = parity.run_batched(lib, I, vm, H, None, 40, density_mode=dmode,
                                    options={'sweep_variant': variant, 'batch': 1 + sd % 7, 'small_flips': (4096, 0, 8)[sd % 3]})
        sweeps += k; amb += res is None
    except Exception as e:
        fails += 1; print('FAIL batched', sd, type(e).__name__, str(e)[:200].replace('\n', ' '))
print('fuzz: %d small + %d medium cases, %d sweeps compared, %d tie-ambiguous, %d FAILED, %.0f s' % (n_small, n_med, sweeps, amb, fails, time.time() - t))
