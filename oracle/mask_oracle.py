"""CPU restatement of the stage-1 voxel passes (generateVesselVolume.py, skeletonization.analyze export).

TEST INFRASTRUCTURE ONLY (same rule as vrg_oracle.c).  The algorithms live in the reference's third-party
dependencies: scipy.ndimage.distance_transform_edt (called at generateVesselVolume.py:183 and
manualCorrectionGUI.py:248; scipy 1.15.3 is installed here, so the oracle IS that call - pinned) and
skimage.measure.label (generateVesselVolume.py:129; scikit-image is NOT installed - scipy.ndimage.label with
the matching structuring element stands in: both label in raster order of a component's first voxel.
PARITY UNPINNED for that numbering convention against skimage itself).
"""
import numpy as np
from scipy import ndimage as ndi


def distance_transform_edt(mask):
    return ndi.distance_transform_edt(np.asarray(mask) != 0)


def labelVolume(volume, minSize=1, maxHop=3):
    """generateVesselVolume.py:107-136 with scipy's label as the skimage stand-in."""
    structure = ndi.generate_binary_structure(3, maxHop)
    labeled, maxNum = ndi.label(np.asarray(volume) != 0, structure=structure)
    counts = np.bincount(labeled.ravel())
    countLoc = np.nonzero(counts)[0]
    sizeList = counts[countLoc]
    return labeled.astype(np.int64), list(zip(countLoc.tolist(), sizeList.tolist()))


def vesselVolumeMask(brainVolumeMask, vesselnessVolume, edtMax=10, frac1=0.8, frac2=0.7, minSize=150):
    """generateVesselVolume.py:177-199, statement by statement."""
    vesselnessVolume2 = np.array(vesselnessVolume, copy=True)
    edt = ndi.distance_transform_edt(brainVolumeMask)                                                        # :183
    minV, maxV = np.amin(vesselnessVolume), np.amax(vesselnessVolume)                                        # :187
    mask = np.logical_and(edt <= edtMax, vesselnessVolume2 <= minV + frac1 * (maxV - minV))                  # :188
    vesselnessVolume2[mask] = 0
    mask = vesselnessVolume2 <= minV + frac2 * (maxV - minV)                                                 # :190
    vesselnessVolume2[mask] = 0
    vesselnessVolume2[vesselnessVolume2 != 0] = 1                                                            # :194
    labeled, labelResult = labelVolume(vesselnessVolume2, minSize=10, maxHop=3)                              # :195
    for labelNum, labelSize in labelResult:
        if labelSize <= minSize:                                                                             # :197-199
            vesselnessVolume2[labeled == labelNum] = 0
    return vesselnessVolume2.astype(np.uint8)
