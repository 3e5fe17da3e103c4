#!/usr/bin/env python3
"""Development tool: the distance transform of tools/bench_mask.py's brain mask alone (for profiler passes).  usage: edt_only.py [shape] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arterynetwork_amd import generateVesselVolume as G
shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '880x880x640').split('x'))
x = np.arange(shape[0], dtype=np.float32)[:, None, None]; y = np.arange(shape[1], dtype=np.float32)[None, :, None]; z = np.arange(shape[2], dtype=np.float32)[None, None, :]
c = [(n - 1) / 2.0 for n in shape]
brain = ((((x - c[0]) / (0.45 * shape[0])) ** 2 + ((y - c[1]) / (0.45 * shape[1])) ** 2 + ((z - c[2]) / (0.45 * shape[2])) ** 2) <= 1.0).astype(np.uint8)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    G.distance_transform_edt(brain)
