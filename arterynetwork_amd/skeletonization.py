"""Hand-off to the external skeletoniser: the export half of the reference's
skeletonization.analyze() (skeletonization.py:97-146) - SURVEY.md section 8 row f3.

Writes exactly the files the curve-skeleton Docker tool reads (`BB.txt`, `xyz.txt`) and the label cache
(`vesselVolumeMaskLabelInfo.npz`) into <baseFolder>/skeletonizationResult; it does not start Docker
(:148-162 is host plumbing outside this path).  Component labelling runs on the GPU.
"""
from __future__ import annotations

import os

import numpy as np

from .generateVesselVolume import labelVolume


def analyze_export(vesselVolumeMask, baseFolder, device=0):
    vesselVolumeMask = np.asarray(vesselVolumeMask).astype(np.uint8)
    vesselVolumeMask[vesselVolumeMask != 0] = 1                                   # :103-104
    vesselVolumeMask = np.swapaxes(vesselVolumeMask, 0, 2)                        # :105
    shape = vesselVolumeMask.shape
    vesselVolumeMaskLabeled, vesselVolumeMaskLabelResult = labelVolume(vesselVolumeMask, minSize=1, device=device)   # :108
    directory = os.path.join(baseFolder, 'skeletonizationResult')
    if not os.path.exists(directory):
        os.makedirs(directory)
        print('Directory {} created.'.format(directory))
    name = 'vesselVolumeMaskLabelInfo.npz'
    path = os.path.join(directory, name)
    np.savez_compressed(path, vesselVolumeMaskLabeled=vesselVolumeMaskLabeled,
                        vesselVolumeMaskLabelResult=vesselVolumeMaskLabelResult)   # :116
    print('{} saved to {}.'.format(name, path))
    with open(os.path.join(directory, 'BB.txt'), 'w') as f1:                       # :128-133
        f1.write('1\n')
        f1.write('{} {} {}\n'.format(0, 0, 0))
        f1.write('{} {} {}'.format(*shape))
    vesselCoords = np.array(np.where(vesselVolumeMask)).T                          # :135
    with open(os.path.join(directory, 'xyz.txt'), 'w') as f2:                      # :136-146: count line + '%1u' rows
        f2.write('{}\n'.format(len(vesselCoords)))
        np.savetxt(f2, vesselCoords, fmt='%1u')
    return directory
