"""NIfTI-in / NIfTI-out driver for the VRG stage: sits between generateVesselVolume.py (stage 1) and
skeletonization.py (stage 3) of the reference pipeline (README.md:201-219).

The reference never wires variationalRegionGrowing into a script (nothing calls it),
so WHICH volume plays `dataArray` and HOW `valueMap` is seeded is this driver's own, documented policy,
following the hints left in the reference (variationalRegionGrowing.py:41-43: `valueMap = full(3)`,
`valueMap[dataArray <= threshold] = 4`):

    dataArray  = brainVolume.nii.gz            masked raw MRA intensities (generateVesselVolume.py:164-166)
    valueMap   = 3 everywhere
                 4 where brainVolumeMask.nii.gz == 0 (outside the brain), or dataArray <= exclude_below
                 0 where vesselVolumeMask.nii.gz != 0  (stage-1 mask = the seeds to be refined)
    output     = segmentedMap as uint8 with the input's affine, written like saveVolume does
                 (generateVesselVolume.py:213-216); skeletonization.py:750-752 loads
                 'vesselVolumeMask.nii.gz' and binarises it with != 0 (:103-104).
"""
from __future__ import annotations

import os

import numpy as np

from .nifti import loadVolume, saveVolume
from .variationalRegionGrowing import variationalRegionGrowing


def build_value_map(dataArray, seedMask, brainMask=None, exclude_below=None):
    valueMap = np.full(dataArray.shape, 3, dtype=np.uint8, order='F')
    if brainMask is not None:
        valueMap[np.asarray(brainMask) == 0] = 4
    if exclude_below is not None:
        valueMap[np.asarray(dataArray) <= exclude_below] = 4
    valueMap[np.asarray(seedMask) != 0] = 0
    return valueMap


def refine(baseFolder, dataName='brainVolume.nii.gz', seedName='vesselVolumeMask.nii.gz',
           brainMaskName='brainVolumeMask.nii.gz', outName='vesselVolumeMaskRefined.nii.gz',
           H=2.25, maxSegmentSize=None, iterMax=200, maxTime=None, exclude_below=None, device=0, quiet=False):
    """Load the stage-1 files from `baseFolder`, run VRG on the GPU, save the refined mask.

    Pass outName='vesselVolumeMask.nii.gz' to overwrite the stage-1 mask so that skeletonization.py picks
    the refined one up unchanged.  Returns (segmented, segmentedMap, valueMap) like the reference function.
    """
    dataArray, affine = loadVolume(baseFolder, dataName)
    seedMask, _ = loadVolume(baseFolder, seedName)
    brainMask = None
    if brainMaskName and os.path.exists(os.path.join(baseFolder, brainMaskName)):
        brainMask, _ = loadVolume(baseFolder, brainMaskName)
    # (any dtype goes through as it is: the library keeps a volume whose values float32 cannot hold - e.g. a NIfTI with a
    # scl_slope, which loadVolume returns as float64 - as float64 on the device, include/vrg.h vrg_set_volume)
    valueMap = build_value_map(dataArray, seedMask, brainMask, exclude_below)
    if maxSegmentSize is None:
        maxSegmentSize = int(dataArray.size) + 1
    segmented, segmentedMap, valueMap = variationalRegionGrowing(
        dataArray, valueMap, H=H, maxSegmentSize=maxSegmentSize, iterMax=iterMax, maxTime=maxTime,
        device=device, quiet=quiet)
    saveVolume(segmentedMap, affine, os.path.join(baseFolder, outName), astype=np.uint8)
    return segmented, segmentedMap, valueMap
