#!/usr/bin/env python3
"""Timing of the stage-1 voxel passes on MI355X next to scipy on the host (development tool).
Host arrays in / out, so the GPU figures include PCIe transfers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy import ndimage as ndi
from arterynetwork_amd import generateVesselVolume as G

def vol(shape, seed=0):
    rng = np.random.default_rng(seed)
    x = np.arange(shape[0], dtype=np.float32)[:, None, None]; y = np.arange(shape[1], dtype=np.float32)[None, :, None]; z = np.arange(shape[2], dtype=np.float32)[None, None, :]
    c = [(n - 1) / 2.0 for n in shape]
    brain = (((x - c[0]) / (0.45 * shape[0])) ** 2 + ((y - c[1]) / (0.45 * shape[1])) ** 2 + ((z - c[2]) / (0.45 * shape[2])) ** 2) <= 1.0
    tube = ((y - c[1] - 0.2 * shape[1] * np.sin(2 * np.pi * x / shape[0])) ** 2 + (z - c[2]) ** 2) <= 9.0
    ves = tube.astype(np.float32) + 0.25 * rng.random(shape, dtype=np.float32)
    return brain.astype(np.uint8), ves

def t(f, *a, reps=3):
    f(*a); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(*a); best = min(best, time.perf_counter() - t0)
    return best, r

for shape in [tuple(int(v) for v in s.split('x')) for s in (sys.argv[1:] or ['512x512x170'])]:
    brain, ves = vol(shape)
    V = brain.size
    fg = (ves > 0.9).astype(np.uint8)
    te, edt = t(G.distance_transform_edt, brain)
    tl, (lab, res) = t(G.labelVolume, fg)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        tp, m = t(G.vesselVolumeMask, brain, ves)
    print('%s (%.1f Mvoxel)  GPU incl. PCIe: edt %.3f s (%.0f Mvox/s)  label26 %.3f s (%.0f Mvox/s, %d comps)  stage-1 pipeline %.3f s (%.0f Mvox/s)' % (
        'x'.join(map(str, shape)), V / 1e6, te, V / te / 1e6, tl, V / tl / 1e6, len(res) - 1, tp, V / tp / 1e6), flush=True)
    if V <= 60e6:
        t0 = time.perf_counter(); e2 = ndi.distance_transform_edt(brain); ce = time.perf_counter() - t0
        t0 = time.perf_counter(); l2, n2 = ndi.label(fg, structure=np.ones((3, 3, 3))); cl = time.perf_counter() - t0
        print('   scipy on 1 host core: edt %.2f s (%.1f Mvox/s, equal=%s)  label26 %.2f s (%.1f Mvox/s, equal=%s)' % (
            ce, V / ce / 1e6, np.array_equal(e2, edt), cl, V / cl / 1e6, np.array_equal(l2, lab)))
