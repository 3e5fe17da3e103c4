"""Hand-off to the external skeletoniser.

The reference's ``skeletonization.analyze()`` (skeletonization.py:97-146) prepares three files for A. Tabb's
curve-skeleton tool before it starts the Docker image (:148-162, host plumbing outside this path).  This module
produces the same three files from a vessel mask; only their *format* is the contract:

``BB.txt``      three lines - ``1``, the lower corner ``0 0 0``, the upper corner = the volume's shape in the tool's
                axis order (z, y, x: the reference swaps axes 0 and 2 first); no trailing newline
``xyz.txt``     first line the number of vessel voxels, then one ``z y x`` row per voxel in raster order of the
                swapped volume, unsigned integers
``vesselVolumeMaskLabelInfo.npz``
                the 26-connected component labels of the swapped volume and the per-component (label, size) table,
                under the reference's key names (its later stages read them back)

Component labelling runs on the GPU (``vmask_label``).
"""
from __future__ import annotations

import os

import numpy as np

from .generateVesselVolume import labelVolume

RESULT_DIR = 'skeletonizationResult'
LABEL_CACHE = 'vesselVolumeMaskLabelInfo.npz'


def to_tool_axes(vesselVolumeMask):
    """Binary uint8 volume in the skeletoniser's (z, y, x) axis order."""
    return np.swapaxes((np.asarray(vesselVolumeMask) != 0).astype(np.uint8), 0, 2)


def write_bb(path, shape):
    """Bounding-box file: one box, from the origin to `shape`."""
    lines = ['1', '0 0 0', ' '.join(str(int(n)) for n in shape)]
    with open(path, 'w') as f:
        f.write('\n'.join(lines))


def write_xyz(path, mask):
    """Voxel list: count, then the coordinates of every non-zero voxel of `mask` in raster order."""
    coords = np.argwhere(mask)
    with open(path, 'w') as f:
        f.write('%d\n' % len(coords))
        np.savetxt(f, coords, fmt='%1u')
    return len(coords)


def write_label_cache(path, labeled, label_result):
    np.savez_compressed(path, vesselVolumeMaskLabeled=labeled, vesselVolumeMaskLabelResult=label_result)


def analyze_export(vesselVolumeMask, baseFolder, device=0):
    """Write BB.txt, xyz.txt and the label cache into <baseFolder>/skeletonizationResult; returns that directory."""
    mask = to_tool_axes(vesselVolumeMask)
    out_dir = os.path.join(baseFolder, RESULT_DIR)
    if not os.path.isdir(out_dir):
        os.makedirs(out_dir)
        print('Directory {} created.'.format(out_dir))
    labeled, label_result = labelVolume(mask, minSize=1, device=device)
    cache = os.path.join(out_dir, LABEL_CACHE)
    write_label_cache(cache, labeled, label_result)
    print('{} saved to {}.'.format(LABEL_CACHE, cache))
    write_bb(os.path.join(out_dir, 'BB.txt'), mask.shape)
    write_xyz(os.path.join(out_dir, 'xyz.txt'), mask)
    return out_dir
