#!/usr/bin/env python3
"""Repeat one random case in whole-run mode (development tool: hunting rare races). args: seed reps small_flips batch [lo hi]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import parity
from test_hostmodel import random_case
from arterynetwork_amd._capi import product_lib
lib = product_lib()
sd, reps, small, batch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
lo, hi = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (1, 13)
I, vm, H, variant, dmode = random_case(sd, lo, hi)
if max(I.shape) > 16: dmode = 1
print('seed', sd, 'shape', I.shape, 'variant', variant, 'levels', len(np.unique(I)), 'dmode', dmode, flush=True)
extra = {k: int(v) for k, v in (kv.split('=') for kv in os.environ.get('VRG_REPEAT_OPTS', '').split(',') if kv)}      # e.g. VRG_REPEAT_OPTS=open_sweeps=0
fails = 0
for r in range(reps):
    try:
        res, k = parity.run_batched(lib, I, vm, H, None, 40, density_mode=dmode, options=dict({'sweep_variant': variant, 'batch': batch, 'small_flips': small}, **extra))
        if r == 0: print('sweeps', k, 'res', None if res is None else res.stop_reason)
    except AssertionError as e:
        fails += 1
        if fails <= 5: print('FAIL rep', r, str(e)[:120].replace('\n', ' '), flush=True)
print('small_flips', small, 'batch', batch, 'reps', reps, 'fails', fails)
