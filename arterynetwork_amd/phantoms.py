"""Synthetic MRA-like inputs for the VRG path (numpy only, deterministic).

These are the recipes SURVEY.md §8(d) fixes for BASELINE.json's configs, plus the
small adversarial volumes the parity tests use.  Every function returns
``(dataArray, valueMap)`` with ``valueMap`` following the reference's label
contract (variationalRegionGrowing.py:21): 0 = seed/inside, 3 = outside, 4 = excluded.

The same recipes are used by tests/golden/make_goldens.py (which runs the real
reference on them in the build container) and by the GPU parity tests/bench, so
goldens and live runs see byte-identical inputs.
"""
from __future__ import annotations

import numpy as np


def straight_line():
    """Reference KAT, variationalRegionGrowing.py:284-289 (inputs only)."""
    volume = np.zeros((50, 50, 150), dtype=int)
    volume[20:22, 20:22, 20:40] = 1
    valueMap = np.full(volume.shape, 3)
    valueMap[20:22, 20:22, 22:25] = 0
    return volume, valueMap


def sphere():
    """Reference KAT, variationalRegionGrowing.py:300-305 (inputs only)."""
    x, y, z = np.mgrid[:50, :50, :50]
    volume = ((x - 25) ** 2 + (y - 25) ** 2 + (z - 25) ** 2 <= 100).astype(int)
    valueMap = np.full(volume.shape, 3)
    valueMap[25:27, 25:27, 25:27] = 0
    return volume, valueMap


def _salted(volume, valueMap, p, seed):
    """Salt noise on a binary / few-level integer volume: a fraction p of the voxels (never a seed) is set to the top level."""
    rng = np.random.default_rng(seed)
    salt = (rng.random(volume.shape) < p) & (valueMap != 0)
    out = volume.copy()
    out[salt] = volume.max()
    return out, valueMap


def straight_line_salt(p=0.05, seed=41):
    """The reference's straight-line KAT (:284-289) cropped to 30x30x60, with 5 % salt noise (integer volume)."""
    volume = np.zeros((30, 30, 60), dtype=int)
    volume[12:14, 12:14, 10:30] = 1
    valueMap = np.full(volume.shape, 3)
    valueMap[12:14, 12:14, 12:15] = 0
    return _salted(volume, valueMap, p, seed)


def sphere_salt(p=0.05, seed=42):
    """The reference's sphere KAT (:300-305) at 30^3 / radius 7, with 5 % salt noise (integer volume)."""
    x, y, z = np.mgrid[:30, :30, :30]
    volume = ((x - 15) ** 2 + (y - 15) ** 2 + (z - 15) ** 2 <= 49).astype(int)
    valueMap = np.full(volume.shape, 3)
    valueMap[15:17, 15:17, 15:17] = 0
    return _salted(volume, valueMap, p, seed)


def two_touching_tubes():
    """Binary volume: two straight tubes (radius 2.5) along x and along y that touch where they cross; seeds in the first."""
    n = 36
    x, y, z = np.mgrid[:n, :n, :n]
    t1 = ((y - 14) ** 2 + (z - 16) ** 2 <= 6.25) & (x >= 3) & (x < 33)
    t2 = ((x - 20) ** 2 + (z - 20) ** 2 <= 6.25) & (y >= 3) & (y < 33)
    volume = (t1 | t2).astype(int)
    valueMap = np.full(volume.shape, 3)
    valueMap[t1 & (x < 6)] = 0
    return volume, valueMap


def torus():
    """Binary torus (R = 11, r = 3) in a 36x36x16 volume; seeds = a short arc of it."""
    nx, ny, nz = 36, 36, 16
    x, y, z = np.mgrid[:nx, :ny, :nz]
    rho = np.sqrt((x - 17.5) ** 2 + (y - 17.5) ** 2)
    tor = (rho - 11.0) ** 2 + (z - 7.5) ** 2 <= 9.0
    volume = tor.astype(int)
    valueMap = np.full(volume.shape, 3)
    valueMap[tor & (x > 26) & (np.abs(y - 17.5) < 2)] = 0
    return volume, valueMap


def three_level(seed=43):
    """Three integer levels: background 0, a tissue ellipsoid 1, a vessel tube 2 inside it, with 3 % of the voxels moved
    one level up or down; excluded (4) outside a larger ellipsoid; seeds = the first planes of the tube."""
    nx, ny, nz = 40, 32, 24
    x, y, z = np.mgrid[:nx, :ny, :nz]
    tissue = ((x - 19.5) / 17.0) ** 2 + ((y - 15.5) / 13.0) ** 2 + ((z - 11.5) / 9.0) ** 2 <= 1.0
    cy = 15.5 + 5.0 * np.sin(2 * np.pi * x / nx)
    tube = ((y - cy) ** 2 + (z - 11.5) ** 2 <= 6.25) & (x >= 4) & (x < 36)
    volume = tissue.astype(int) + tube.astype(int)
    rng = np.random.default_rng(seed)
    u = rng.random(volume.shape)
    volume = np.clip(volume + (u < 0.015).astype(int) - (u > 0.985).astype(int), 0, 2)
    valueMap = np.full(volume.shape, 3)
    valueMap[((x - 19.5) / 19.0) ** 2 + ((y - 15.5) / 15.0) ** 2 + ((z - 11.5) / 11.0) ** 2 > 1.0] = 4
    valueMap[tube & (x < 7)] = 0
    return volume, valueMap


def tube_phantom(shape=(128, 128, 64), radius=3.5, noise=0.1, seed=2024,
                 seed_planes=4, amp_y=20.0, amp_z=8.0, levels=None,
                 brain_mask=False, dtype=np.float64, noise_dtype=np.float64):
    """Sinusoid tube along x (SURVEY.md §8(d) "Config 1" recipe when called with defaults).

    centreline cy = ny/2 + amp_y*sin(2*pi*x/nx), cz = nz/2 + amp_z*cos(2*pi*x/nx);
    tube (y-cy)^2 + (z-cz)^2 <= radius^2; I = tube + noise*N(0,1) drawn with
    default_rng(seed).standard_normal in C order, cast through float32.
    ``levels``: if given, quantise I to round(I*levels)/levels (integer-level volume).
    ``brain_mask``: label 4 (excluded) outside a centred ellipsoid.
    Seeds (label 0) = tube voxels with x < seed_planes.
    """
    nx, ny, nz = shape
    x = np.arange(nx, dtype=np.float64)[:, None, None]
    y = np.arange(ny, dtype=np.float64)[None, :, None]
    z = np.arange(nz, dtype=np.float64)[None, None, :]
    cy = ny / 2.0 + amp_y * np.sin(2 * np.pi * x / nx)
    cz = nz / 2.0 + amp_z * np.cos(2 * np.pi * x / nx)
    tube = ((y - cy) ** 2 + (z - cz) ** 2) <= radius ** 2
    rng = np.random.default_rng(seed)
    if noise_dtype == np.float64:   # survey recipe: float64 draw, then one cast through float32
        I = (tube + noise * rng.standard_normal(shape)).astype(np.float32)
    else:                           # large volumes: draw float32 directly (half the host memory)
        I = tube.astype(np.float32) + np.float32(noise) * rng.standard_normal(shape, dtype=np.float32)
    if levels is not None:
        I = (np.round(I * np.float32(levels)) / np.float32(levels)).astype(np.float32)
    valueMap = np.full(shape, 3, dtype=np.int64)
    if brain_mask:
        ell = (((x - (nx - 1) / 2.0) / (0.48 * nx)) ** 2 + ((y - (ny - 1) / 2.0) / (0.48 * ny)) ** 2
               + ((z - (nz - 1) / 2.0) / (0.48 * nz)) ** 2) <= 1.0
        valueMap[~np.broadcast_to(ell, shape)] = 4
    seeds = tube & (np.arange(nx)[:, None, None] < seed_planes)
    valueMap[seeds] = 0
    return I.astype(dtype), valueMap


def config1():
    """BASELINE.json configs[0]: 128x128x64 tube phantom (continuous-valued, seed 2024)."""
    return tube_phantom()


def noise_volume(shape, seed, p_seed=0.2, p_excl=0.3, levels=None):
    """Pure-noise adversarial volume: every voxel independently seed / excluded / outside."""
    rng = np.random.default_rng(seed)
    I = rng.standard_normal(shape).astype(np.float32)
    if levels is not None:
        I = (np.round(I * np.float32(levels)) / np.float32(levels)).astype(np.float32)
    u = rng.random(shape)
    valueMap = np.full(shape, 3, dtype=np.int64)
    valueMap[u < p_seed] = 0
    valueMap[u > 1.0 - p_excl] = 4
    return I.astype(np.float64), valueMap


def scattered_seeds(shape=(24, 24, 24), seed=7, n_seeds=40, p_excl=0.35):
    """Single-voxel seeds scattered in a two-blob intensity field with excluded voxels adjacent
    to seeds: exercises ghost outer-boundary voxels and skipped flips (SURVEY.md §8c iii)."""
    rng = np.random.default_rng(seed)
    nx, ny, nz = shape
    x, y, z = np.mgrid[:nx, :ny, :nz]
    blob = (((x - nx * 0.35) ** 2 + (y - ny * 0.4) ** 2 + (z - nz * 0.5) ** 2) <= (0.22 * nx) ** 2) | \
           (((x - nx * 0.7) ** 2 + (y - ny * 0.65) ** 2 + (z - nz * 0.45) ** 2) <= (0.18 * nx) ** 2)
    I = blob.astype(np.float32) + np.float32(0.35) * rng.standard_normal(shape).astype(np.float32)
    valueMap = np.full(shape, 3, dtype=np.int64)
    valueMap[rng.random(shape) < p_excl] = 4
    flat = rng.choice(nx * ny * nz, size=n_seeds, replace=False)
    valueMap.reshape(-1)[flat] = 0
    return I.astype(np.float64), valueMap


def shell_with_holes(n=20, seed=11, hole_frac=0.08):
    """Ball whose seed set is a thick shell with random pin-holes and a hollow core:
    exercises hole filling (ghost inner-boundary voxels, omitted density terms)."""
    rng = np.random.default_rng(seed)
    x, y, z = np.mgrid[:n, :n, :n]
    c = (n - 1) / 2.0
    r2 = (x - c) ** 2 + (y - c) ** 2 + (z - c) ** 2
    ball = r2 <= (0.42 * n) ** 2
    shell = ball & (r2 >= (0.2 * n) ** 2)
    I = ball.astype(np.float32) + np.float32(0.25) * rng.standard_normal((n, n, n)).astype(np.float32)
    valueMap = np.full((n, n, n), 3, dtype=np.int64)
    seeds = shell & (rng.random((n, n, n)) > hole_frac)
    valueMap[seeds] = 0
    return I.astype(np.float64), valueMap


def bench_volume(shape, seed, levels=255, radius=4.0, noise=0.1, seed_planes=3):
    """Configs 2-4 family: helix/sinusoid tube longer than the sweep count,
    integer-level intensities stored fp32, excluded voxels outside a centred ellipsoid."""
    nx, ny, nz = shape
    return tube_phantom(shape=shape, radius=radius, noise=noise, seed=seed, seed_planes=seed_planes,
                        amp_y=0.18 * ny, amp_z=0.18 * nz, levels=levels, brain_mask=True,
                        dtype=np.float32, noise_dtype=np.float32)


def tube_lattice(shape, tubes):
    """Centres (cy, cz), radius and wiggle amplitude of `tubes` disjoint tubes running along x (SURVEY.md 8(d), config 5:
    "several disjoint tubes"): a gy x gz lattice over the central 60 % of the (y, z) cross-section - inside the brain
    ellipsoid for the central third of the x range.  Radius 4 where the pitch allows it (at least 3 voxels stay between
    two tubes: their one-voxel bands never touch), smaller on a dense lattice; the wiggle keeps that gap."""
    import math
    nx, ny, nz = shape
    gy = max(1, int(round(math.sqrt(tubes * (0.6 * ny) / (0.6 * nz)))))
    gz = (tubes + gy - 1) // gy
    py, pz = 0.6 * ny / gy, 0.6 * nz / gz
    radius = min(4.0, (min(py, pz) - 3.0) / 2.0)
    if radius < 1.5:
        raise ValueError('{} tubes do not fit a {}x{} cross-section'.format(tubes, ny, nz))
    amp = max(0.0, (min(py, pz) - 3.0 - 2.0 * radius) / 2.0)
    cen = [(0.2 * ny + (j % gy + 0.5) * py, 0.2 * nz + (j // gy + 0.5) * pz) for j in range(tubes)]
    return cen, radius, min(amp, 0.18 * min(ny, nz))


def bench_volume_torch(shape, device, seed=3, levels=255, radius=4.0, noise=0.1, seed_planes=3, brain_mask=True, integer_values=False,
                       tubes=1, seed_mode='planes', brain_scale=0.48):
    """The configs 2-4 recipe generated directly in HBM with torch (plumbing only), x-fastest
    layout.  Returns (I, vm) as torch tensors of logical shape (nx,ny,nz) with element strides (1,nx,nx*ny).
    ``levels=None`` keeps the continuous float32 noise (one distinct value per voxel, nearly).
    ``tubes`` > 1: that many disjoint tubes on a lattice (tube_lattice), seeded at the central x planes so that every tube
    grows both ways - about 100 flips per tube and sweep: the many-flip regime.  ``seed_mode='whole'``: every tube voxel
    is a seed (what refine() does with a stage-1 mask: all vessels at once).  ``seed_mode='noisy-mask'``: the seed is a
    PERTURBED vessel mask, what a vesselness threshold hands to this stage (generateVesselVolume.py:187-199; README.md:69-71: VRG
    smooths an existing mask) - the tube mask with a random half of its surface voxels (26-connected erosion) taken off, plus
    everything within two voxels of 0.6 % of them ("bites"), plus salt: isolated false positives, 2 % of the mask's voxel count (a tenth of
    them 3x3x3 blobs), anywhere in the volume - inside the brain mask or out of it (a seed overrides the excluded label).  The first
    sweep then flips 10^4-10^5 voxels and the count decays over a few sweeps to convergence."""
    import math
    import torch
    nx, ny, nz = shape
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    xs = torch.arange(nx, device=device, dtype=torch.float32)[None, None, :]
    ys = torch.arange(ny, device=device, dtype=torch.float32)[None, :, None]
    zs = torch.arange(nz, device=device, dtype=torch.float32)[:, None, None]
    if tubes <= 1:
        cy = ny / 2.0 + 0.18 * ny * torch.sin(2 * math.pi * xs / nx)
        cz = nz / 2.0 + 0.18 * nz * torch.cos(2 * math.pi * xs / nx)
        tube = ((ys - cy) ** 2 + (zs - cz) ** 2) <= radius ** 2            # (nz,ny,nx)
        seeds = tube & (xs < seed_planes)
    else:
        cen, rad, amp = tube_lattice(shape, tubes)
        tube = torch.zeros((nz, ny, nx), dtype=torch.bool, device=device)
        wy, wz = amp * torch.sin(2 * math.pi * xs / nx * 3.0), amp * torch.cos(2 * math.pi * xs / nx * 3.0)
        r1 = int(math.ceil(rad + amp)) + 1
        for (c_y, c_z) in cen:                                            # (each tube only touches its own box of the volume)
            y0, y1 = max(0, int(c_y) - r1), min(ny, int(c_y) + r1 + 2)
            z0, z1 = max(0, int(c_z) - r1), min(nz, int(c_z) + r1 + 2)
            tube[z0:z1, y0:y1, :] |= ((ys[:, y0:y1] - (c_y + wy)) ** 2 + (zs[z0:z1] - (c_z + wz)) ** 2) <= rad ** 2
        seeds = tube & (xs >= nx // 2 - (seed_planes + 1) // 2) & (xs < nx // 2 + seed_planes // 2)
    if seed_mode == 'whole':
        seeds = tube
    elif seed_mode == 'noisy-mask':
        import torch.nn.functional as F
        outside = (~tube).to(torch.float32)[None, None]
        eroded = F.max_pool3d(outside, kernel_size=3, stride=1, padding=1)[0, 0] < 0.5      # no non-tube voxel among the 26 neighbours (the volume's faces count as tube)
        surface = tube & ~eroded
        u = torch.rand((nz, ny, nx), generator=g, device=device, dtype=torch.float32)
        salt_p = 0.02 * float(tube.sum().item()) / float(nx * ny * nz)
        # bites: everything within two voxels of 0.6 % of the surface voxels is missing too (a threshold loses stretches of a thin vessel, not single
        # voxels), and a tenth of the salt comes as 3x3x3 blobs - so the mask is several sweeps away from its fixed point, not one
        bite = F.max_pool3d((surface & (u > 0.5) & (u < 0.506)).to(torch.float32)[None, None], kernel_size=5, stride=1, padding=2)[0, 0] > 0.5
        blob = F.max_pool3d((~tube & (u > 1.0 - 0.1 * salt_p)).to(torch.float32)[None, None], kernel_size=3, stride=1, padding=1)[0, 0] > 0.5
        seeds = (tube & ~(surface & (u < 0.5)) & ~bite) | (~tube & ((u > 1.0 - salt_p) | blob))
        del outside, eroded, surface, u, bite, blob
    I = torch.randn((nz, ny, nx), generator=g, device=device, dtype=torch.float32)
    I.mul_(noise).add_(tube.to(torch.float32))
    if levels:
        I = torch.round(I * levels)                 # integer_values: what a scanner delivers (the caller scales H by 1 / levels^2: same run)
        if not integer_values:
            I = I / levels
    # (brain_scale: the ellipsoid's half axes as a fraction of the volume's - smaller than the tubes' lattice and the vessels cross the brain mask's edge:
    # excluded voxels right beside the flips, the 4 -> 3 inclusion rule :166-168, :177-179 at work in every sweep)
    ell = (((xs - (nx - 1) / 2.0) / (brain_scale * nx)) ** 2 + ((ys - (ny - 1) / 2.0) / (brain_scale * ny)) ** 2
           + ((zs - (nz - 1) / 2.0) / (brain_scale * nz)) ** 2) <= 1.0
    vm = torch.full((nz, ny, nx), 3, dtype=torch.uint8, device=device)
    if brain_mask:
        vm[~ell.expand(nz, ny, nx)] = 4
    vm[seeds] = 0
    return I.permute(2, 1, 0), vm.permute(2, 1, 0)
