#!/usr/bin/env python3
"""Repeat one random parity case many times (development tool: hunting rare races)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import parity
from test_hostmodel import random_case
from arterynetwork_amd._capi import product_lib
lib = product_lib()
sd, reps = int(sys.argv[1]), int(sys.argv[2])
nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 3
I, vm, H, variant, dmode = random_case(sd, 10, 40)
print('seed', sd, 'shape', I.shape, 'variant', variant, 'levels', len(np.unique(I)), flush=True)
fails = 0
for r in range(reps):
    try:
        parity.run_stepwise(lib, I, vm, H, None, nmax, density_mode=1, check_hist=True, options={'sweep_variant': variant})
    except AssertionError as e:
        fails += 1
        if fails <= 5: print('FAIL rep', r, str(e)[:120].replace('\n', ' '), flush=True)
print('reps', reps, 'fails', fails)
