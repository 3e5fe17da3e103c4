"""MI355X-native counterparts of the voxel passes in the reference's Code/generateVesselVolume.py
(stage 1: vessel mask generation) - SURVEY.md section 8 row f2, plus the EDT that
manualCorrectionGUI.py:248 uses for vessel radii (row f4).

Same function names and return conventions as the reference where it has functions
(`labelVolume`, `maskVolume`, `loadVolume`, `saveVolume`); the body of its `main()` (:177-199) is
exposed as `vesselVolumeMask(...)`.  The work runs in HIP kernels behind include/vmask.h; no CPU path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from ._capi import product_lib, VrgError
from .nifti import loadVolume, saveVolume  # noqa: F401  (same names as the reference module exports)

_bound = None


def _lib():
    global _bound
    if _bound is None:
        dll = product_lib().dll
        p, i64 = C.c_void_p, C.c_int64
        dll.vmask_edt.argtypes = [C.c_int, p, i64, i64, i64, p]
        dll.vmask_label.argtypes = [C.c_int, p, i64, i64, i64, C.c_int, p, p, i64, C.POINTER(i64)]
        dll.vmask_vessel_mask.argtypes = [C.c_int, p, p, C.c_int, i64, i64, i64, C.c_double, C.c_double, C.c_double,
                                          i64, p, C.POINTER(i64)]
        dll.vmask_last_error.restype = C.c_char_p
        _bound = dll
    return _bound


def _check(rc):
    if rc != 0:
        raise VrgError(rc, _lib().vmask_last_error().decode())


def _on_device(a):
    """A torch tensor that lives on the GPU: the volume is handed to the C-ABI by its device pointer (include/vmask.h accepts
    host and device pointers alike) and the result comes back as a tensor on the same device - stage 1 -> VRG -> export can
    then run without a PCIe round trip per call (0.3 s each at 880x880x640 with host arrays)."""
    return hasattr(a, 'data_ptr') and bool(getattr(a, 'is_cuda', False))


def _dev_index(t):
    return t.device.index if t.device.index is not None else 0


def _u8t(t):
    import torch
    if t.dim() != 3:
        raise ValueError('expected a 3-D volume')
    return (t != 0).to(torch.uint8).contiguous()


def _u8c(a):
    a = np.asarray(a)
    if a.ndim != 3:
        raise ValueError('expected a 3-D volume')
    return np.ascontiguousarray(a != 0, dtype=np.uint8)


def distance_transform_edt(mask, device=0):
    """scipy.ndimage.distance_transform_edt(mask) with unit sampling (generateVesselVolume.py:183,
    manualCorrectionGUI.py:248): float64 distance of every non-zero voxel to the nearest zero voxel."""
    if _on_device(mask):                               # device-resident: tensor in, tensor out
        import torch
        m = _u8t(mask)
        out = torch.empty(m.shape, dtype=torch.float64, device=m.device)
        torch.cuda.synchronize(m.device)
        _check(_lib().vmask_edt(_dev_index(m), m.data_ptr(), *m.shape, out.data_ptr()))
        return out
    m = _u8c(mask)
    out = np.empty(m.shape, np.float64)
    _check(_lib().vmask_edt(device, m.ctypes.data, *m.shape, out.ctypes.data))
    return out


DISTANCE_CACHE = 'vesselVolumeDistanceTransform.npz'


def vesselDistanceTransform(vesselVolume, directory=None, device=0):
    """The vessel mask's distance transform as the reference's graph stage keeps it (manualCorrectionGUI.py:242-248,
    SURVEY.md section 8 row f4): read from ``<directory>/vesselVolumeDistanceTransform.npz`` (key ``distanceTransform``)
    when that file exists, otherwise computed on the GPU and written there.  A voxel's vessel radius is then a plain
    look-up, ``distanceTransform[tuple(coords.T)]``."""
    import os
    path = os.path.join(directory, DISTANCE_CACHE) if directory is not None else None
    if path is not None and os.path.exists(path):
        return np.load(path)['distanceTransform']
    dt = distance_transform_edt(vesselVolume, device=device)
    if path is not None:
        np.savez_compressed(path, distanceTransform=dt)
    return dt


def labelVolume(volume, minSize=1, maxHop=3, device=0):
    """
    Partition the volume into connected components and attach labels (generateVesselVolume.py:107-136).

    Returns
    -------
    labeled : ndarray
        0 = background, components 1..n numbered in raster order of their first voxel (as
        skimage.measure.label(volume, connectivity=maxHop) numbers them).
    labelResult : list
        [(label, size), ...] for every label present, background included, exactly like
        np.bincount(labeled.ravel()) filtered to non-zero counts (:131-134).  `minSize` is accepted and
        unused, as in this reference function (the copy in skeletonization.py filters by it; its only caller passes 1).

    Only binary volumes are accepted: skimage.measure.label connects voxels of EQUAL value, so a volume with several
    non-zero values would be partitioned differently from what this kernel (which looks at zero / non-zero) does -
    such input raises instead of returning different labels.
    """
    if _on_device(volume):                             # device-resident: labels stay on the GPU (int32), the sizes come to the host
        import torch
        nzv = volume[volume != 0]
        if nzv.numel() and bool((nzv != nzv.flatten()[0]).any()):
            raise ValueError('labelVolume: only binary volumes are supported (several distinct non-zero values found)')
        v = _u8t(volume)
        labeled = torch.empty(v.shape, dtype=torch.int32, device=v.device)
        n = C.c_int64()
        cap = max(1, v.numel() // 2 + 1)
        sizes = np.empty(cap, np.int64)
        torch.cuda.synchronize(v.device)
        _check(_lib().vmask_label(_dev_index(v), v.data_ptr(), *v.shape, int(maxHop), labeled.data_ptr(), sizes.ctypes.data, cap, C.byref(n)))
        ncomp = n.value
        labelResult = []
        nbg = int(v.numel() - sizes[:ncomp].sum())
        if nbg:
            labelResult.append((0, nbg))
        labelResult.extend((k + 1, int(sizes[k])) for k in range(ncomp))
        return labeled, labelResult
    a = np.asarray(volume)
    nz = a[a != 0]
    if nz.size and np.any(nz != nz.flat[0]):
        raise ValueError('labelVolume: only binary volumes are supported (several distinct non-zero values found)')
    v = _u8c(volume)
    labeled = np.empty(v.shape, np.int32)
    n = C.c_int64()
    cap = max(1, v.size // 2 + 1)
    sizes = np.empty(cap, np.int64)
    _check(_lib().vmask_label(device, v.ctypes.data, *v.shape, int(maxHop), labeled.ctypes.data, sizes.ctypes.data,
                              cap, C.byref(n)))
    ncomp = n.value
    labelResult = []
    nbg = int(v.size - sizes[:ncomp].sum())
    if nbg:
        labelResult.append((0, nbg))
    labelResult.extend((k + 1, int(sizes[k])) for k in range(ncomp))
    return labeled.astype(np.int64), labelResult


def maskVolume(volume, mask):
    """Apply the given volume mask to the given volume (generateVesselVolume.py:86-105)."""
    newVolume = np.array(volume, copy=True)
    newVolume[np.asarray(mask) == 0] = 0
    return newVolume


def vesselVolumeMask(brainVolumeMask, vesselnessVolume, edtMax=10, frac1=0.8, frac2=0.7, minSize=150, device=0):
    """The body of the reference's main() between loading and saving (generateVesselVolume.py:177-199):
    suppress weak vesselness near the brain-mask boundary (EDT <= 10 and <= min + 0.8*range), threshold at
    min + 0.7*range, binarise, drop 26-connected components of <= 150 voxels.  Returns the uint8 mask."""
    if _on_device(vesselnessVolume):                   # device-resident: the uint8 mask comes back as a tensor on the same GPU
        import torch
        ves = vesselnessVolume.contiguous()
        if ves.dtype not in (torch.float32, torch.float64):
            ves = ves.to(torch.float64)
        b = _u8t(brainVolumeMask if _on_device(brainVolumeMask) else torch.as_tensor(np.asarray(brainVolumeMask), device=ves.device))
        if tuple(b.shape) != tuple(ves.shape):
            raise ValueError('brainVolumeMask and vesselnessVolume must have the same shape')
        out = torch.empty(ves.shape, dtype=torch.uint8, device=ves.device)
        kept = C.c_int64()
        torch.cuda.synchronize(ves.device)
        _check(_lib().vmask_vessel_mask(_dev_index(ves), b.data_ptr(), ves.data_ptr(), 5 if ves.dtype == torch.float32 else 6,
                                        *ves.shape, float(edtMax), float(frac1), float(frac2), int(minSize), out.data_ptr(), C.byref(kept)))
        print('Number of voxels in segmentation: {}'.format(kept.value))      # :211
        return out
    ves = np.ascontiguousarray(vesselnessVolume)
    if ves.dtype not in (np.float32, np.float64):
        ves = ves.astype(np.float64)
    b = _u8c(brainVolumeMask)
    if b.shape != ves.shape:
        raise ValueError('brainVolumeMask and vesselnessVolume must have the same shape')
    out = np.empty(ves.shape, np.uint8)
    kept = C.c_int64()
    _check(_lib().vmask_vessel_mask(device, b.ctypes.data, ves.ctypes.data, 5 if ves.dtype == np.float32 else 6,
                                    *ves.shape, float(edtMax), float(frac1), float(frac2), int(minSize),
                                    out.ctypes.data, C.byref(kept)))
    print('Number of voxels in segmentation: {}'.format(kept.value))          # :211
    return out


def main(baseFolder=None, rawVolumeName='401 3D MRA BRAIN.nii.gz'):
    """File-level equivalent of the reference's main() (:138-228) for a folder holding the same files."""
    if baseFolder is None:
        baseFolder = os.getcwd()
    _, rawVolumeAffine = loadVolume(baseFolder, rawVolumeName)
    brainVolumeMask, _ = loadVolume(baseFolder, 'brainVolumeMask.nii.gz')
    vesselnessVolume, _ = loadVolume(baseFolder, 'vesselnessFiltered.nii.gz')
    mask = vesselVolumeMask(brainVolumeMask, vesselnessVolume)
    saveVolume(mask, rawVolumeAffine, os.path.join(baseFolder, 'vesselVolumeMask.nii.gz'), astype=np.uint8)   # :213-216
    return mask
