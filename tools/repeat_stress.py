import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import parity
from arterynetwork_amd import phantoms
from arterynetwork_amd._capi import product_lib
lib = product_lib()
cases = (((96, 96, 96), 11, 6, 12, {'batch': 4}), ((128, 120, 112), 12, 10, 8, {'batch': 8, 'storage16': 1}), ((200, 180, 150), 13, 5, 5, {'batch': 8}))
datas = [phantoms.noise_volume(sh, seed, p_seed=0.2, p_excl=0.3, levels=lv) for sh, seed, lv, sw, op in cases]
n = int(sys.argv[1]); fails = 0
for it in range(n):
    for (sh, seed, lv, sw, op), (data, vmap) in zip(cases, datas):
        try:
            parity.run_batched(lib, data, vmap, 2.25, None, sw, density_mode=1, rtol=1e-8, options=op)
        except AssertionError as e:
            fails += 1; print('FAIL it', it, sh, str(e)[:150].replace('\n', ' '), flush=True)
print('done', n, 'iterations', fails, 'failures')
