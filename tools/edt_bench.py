#!/usr/bin/env python3
"""Device-resident timing of the exact distance transform - development tool.
    python3 tools/edt_bench.py [shape] [reps]          (run under `rocprofv3 --kernel-trace --stats` for per-kernel times)
The mask is the brain-sized ellipsoid of tools/bench_mask.py (a large solid region: deep envelope stacks) and, second,
a thin tube (a vessel mask: nearly every voxel background), both built in HBM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arterynetwork_amd import generateVesselVolume as G

shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '880x880x640').split('x'))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device('cuda:0')
x = torch.arange(shape[0], dtype=torch.float32, device=dev)[:, None, None]
y = torch.arange(shape[1], dtype=torch.float32, device=dev)[None, :, None]
z = torch.arange(shape[2], dtype=torch.float32, device=dev)[None, None, :]
c = [(n - 1) / 2.0 for n in shape]
masks = {
    'ellipsoid': ((((x - c[0]) / (0.45 * shape[0])) ** 2 + ((y - c[1]) / (0.45 * shape[1])) ** 2 + ((z - c[2]) / (0.45 * shape[2])) ** 2) <= 1.0).to(torch.uint8),
    'tube': (((y - c[1] - 0.2 * shape[1] * torch.sin(2 * torch.pi * x / shape[0])) ** 2 + (z - c[2]) ** 2) <= 9.0).expand(shape).to(torch.uint8).contiguous(),
}
V = shape[0] * shape[1] * shape[2]
for name, m in masks.items():
    out = G.distance_transform_edt(m)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = G.distance_transform_edt(m)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print('%s %s: %.2f ms per transform incl. scratch allocation (%.0f Mvoxel/s), max distance %.3f, checksum %.6f' % (
        'x'.join(map(str, shape)), name, dt * 1e3, V / dt / 1e6, float(out.max()), float(out.double().sum())), flush=True)
