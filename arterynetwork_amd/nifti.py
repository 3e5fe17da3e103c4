"""Minimal NIfTI-1 single-file (.nii / .nii.gz) reader and writer - the file contract on either side
of the VRG stage.

The reference goes through nibabel (`loadVolume` / `saveVolume`, generateVesselVolume.py:15-40, :65-84,
duplicated in skeletonization.py:19-65); nibabel is not a dependency here, so the two helpers are
re-implemented on numpy with the same names, arguments, return values and printed messages:

    volume, affine = loadVolume(volumeFolderPath, volumeName)      # like nib.load(...).get_data(), .affine
    saveVolume(volume, affine, path, astype=None)                  # like nib.save(nib.Nifti1Image(volume.astype(astype), affine), path)

Volumes come back in Fortran order (x fastest), which is also the layout the HIP library streams, so a
loaded volume goes to the GPU without a host-side transpose.
"""
from __future__ import annotations

import gzip
import os
import struct

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8,
           512: np.uint16, 768: np.uint32, 1024: np.int64, 1280: np.uint64}
_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith('.gz') else open(path, mode)


def _quat_to_affine(b, c, d, qoff, pixdim):
    a2 = 1.0 - (b * b + c * c + d * d)
    a = np.sqrt(a2) if a2 > 1e-12 else 0.0
    R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
    qfac = -1.0 if pixdim[0] < 0 else 1.0
    zooms = np.array([pixdim[1], pixdim[2], pixdim[3] * qfac], dtype=np.float64)
    aff = np.eye(4)
    aff[:3, :3] = R * zooms
    aff[:3, 3] = qoff
    return aff


def read(path):
    """Return (data, affine, header_dict).  data is what nibabel's get_data() returns: the stored dtype,
    or float64 after scl_slope / scl_inter scaling when the header asks for it."""
    with _open(path, 'rb') as f:
        raw = f.read()
    if len(raw) < 348:
        raise ValueError('{}: not a NIfTI-1 file'.format(path))
    end = '<' if struct.unpack('<i', raw[:4])[0] == 348 else '>'
    if struct.unpack(end + 'i', raw[:4])[0] != 348:
        raise ValueError('{}: bad sizeof_hdr'.format(path))
    magic = raw[344:348]
    if magic[:3] != b'n+1':
        raise ValueError('{}: only single-file NIfTI-1 (magic n+1) is supported'.format(path))
    dim = struct.unpack(end + '8h', raw[40:56])
    datatype, bitpix = struct.unpack(end + '2h', raw[70:74])
    pixdim = struct.unpack(end + '8f', raw[76:108])
    vox_offset, slope, inter = struct.unpack(end + '3f', raw[108:120])
    qform_code, sform_code = struct.unpack(end + '2h', raw[252:256])
    qb, qc, qd, qx, qy, qz = struct.unpack(end + '6f', raw[256:280])
    srow = np.array(struct.unpack(end + '12f', raw[280:328]), dtype=np.float64).reshape(3, 4)
    if datatype not in _DTYPES:
        raise ValueError('{}: unsupported NIfTI datatype {}'.format(path, datatype))
    ndim = dim[0]
    shape = tuple(int(d) for d in dim[1:1 + ndim])
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(end)
    n = int(np.prod(shape))
    off = int(vox_offset) if vox_offset >= 352 else 352
    data = np.frombuffer(raw, dtype=dt, count=n, offset=off).reshape(shape, order='F')
    data = data.astype(dt.newbyteorder('='), copy=True, order='F')
    if np.isfinite(slope) and slope != 0 and not (slope == 1.0 and (inter == 0 or not np.isfinite(inter))):
        data = data.astype(np.float64) * float(slope) + (float(inter) if np.isfinite(inter) else 0.0)
    if sform_code > 0:
        affine = np.vstack((srow, [0, 0, 0, 1.0]))
    elif qform_code > 0:
        affine = _quat_to_affine(qb, qc, qd, (qx, qy, qz), pixdim)
    else:                                   # nibabel's base affine: zooms on the diagonal, x flipped, centred
        zooms = np.array([pixdim[1] or 1.0, pixdim[2] or 1.0, pixdim[3] or 1.0], dtype=np.float64)
        affine = np.diag([-zooms[0], zooms[1], zooms[2], 1.0])
        affine[:3, 3] = -affine[:3, :3] @ ((np.array(shape[:3]) - 1) / 2.0)
    hdr = dict(dim=dim, datatype=datatype, bitpix=bitpix, pixdim=pixdim, vox_offset=vox_offset, scl_slope=slope,
               scl_inter=inter, qform_code=qform_code, sform_code=sform_code, endian=end)
    return data, affine, hdr


def _affine_to_quat(aff):
    RZS = aff[:3, :3]
    zooms = np.sqrt((RZS * RZS).sum(axis=0))
    zooms[zooms == 0] = 1.0
    R = RZS / zooms
    qfac = 1.0
    if np.linalg.det(R) < 0:
        R[:, 2] *= -1
        qfac = -1.0
    # closest rotation, then the usual matrix -> quaternion (a >= 0) conversion
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    a = 0.5 * np.sqrt(max(1.0 + tr, 0.0))
    if a > 0.5 * 1e-3 ** 0.5:
        b = 0.25 * (R[2, 1] - R[1, 2]) / a
        c = 0.25 * (R[0, 2] - R[2, 0]) / a
        d = 0.25 * (R[1, 0] - R[0, 1]) / a
    else:
        xd, yd, zd = 1.0 + R[0, 0] - (R[1, 1] + R[2, 2]), 1.0 + R[1, 1] - (R[0, 0] + R[2, 2]), 1.0 + R[2, 2] - (R[0, 0] + R[1, 1])
        if xd > 1.0:
            b = 0.5 * np.sqrt(xd); c = 0.25 * (R[0, 1] + R[1, 0]) / b; d = 0.25 * (R[0, 2] + R[2, 0]) / b; a = 0.25 * (R[2, 1] - R[1, 2]) / b
        elif yd > 1.0:
            c = 0.5 * np.sqrt(yd); b = 0.25 * (R[0, 1] + R[1, 0]) / c; d = 0.25 * (R[1, 2] + R[2, 1]) / c; a = 0.25 * (R[0, 2] - R[2, 0]) / c
        else:
            d = 0.5 * np.sqrt(zd); b = 0.25 * (R[0, 2] + R[2, 0]) / d; c = 0.25 * (R[1, 2] + R[2, 1]) / d; a = 0.25 * (R[1, 0] - R[0, 1]) / d
        if a < 0:
            b, c, d = -b, -c, -d
    return (b, c, d), zooms, qfac


def write(path, data, affine):
    """Write `data` (any supported dtype, 1..7 dims) with `affine` as sform (code 2, like nibabel's default)."""
    data = np.asarray(data)
    if data.dtype == np.bool_:
        data = data.astype(np.uint8)
    if data.dtype not in _CODES:
        raise ValueError('unsupported dtype {}'.format(data.dtype))
    affine = np.asarray(affine, dtype=np.float64).reshape(4, 4)
    hdr = bytearray(352)
    struct.pack_into('<i', hdr, 0, 348)
    dim = [data.ndim] + list(data.shape) + [1] * (7 - data.ndim)
    struct.pack_into('<8h', hdr, 40, *dim)
    struct.pack_into('<2h', hdr, 70, _CODES[data.dtype], data.dtype.itemsize * 8)
    (qb, qc, qd), zooms, qfac = _affine_to_quat(affine)
    pixdim = [qfac, zooms[0], zooms[1], zooms[2], 1.0, 1.0, 1.0, 1.0]
    struct.pack_into('<8f', hdr, 76, *pixdim)
    struct.pack_into('<3f', hdr, 108, 352.0, float('nan'), float('nan'))
    struct.pack_into('<2h', hdr, 252, 0, 2)                       # qform unknown, sform aligned
    struct.pack_into('<6f', hdr, 256, qb, qc, qd, *affine[:3, 3])
    struct.pack_into('<12f', hdr, 280, *affine[:3, :].reshape(-1))
    hdr[344:348] = b'n+1\x00'
    payload = np.asfortranarray(data).astype(data.dtype.newbyteorder('<'), copy=False).tobytes(order='F')
    with _open(path, 'wb') as f:
        f.write(bytes(hdr))
        f.write(payload)


def loadVolume(volumeFolderPath, volumeName):
    """
    Load nifti files (*.nii or *.nii.gz) - same contract as generateVesselVolume.py:15-40.

    Returns
    -------
    volume : ndarray
        Volume data in the form of numpy ndarray.
    affine : ndarray
        Associated affine transformation matrix in the form of numpy ndarray.
    """
    volumeFilePath = os.path.join(volumeFolderPath, volumeName)
    volume, affine, _ = read(volumeFilePath)
    print('Volume loaded from {} with shape = {}.'.format(volumeFilePath, volume.shape))
    return volume, affine


def saveVolume(volume, affine, path, astype=None):
    """
    Save the given volume to the specified location in specified data type - same contract as
    generateVesselVolume.py:65-84 (default type uint8).
    """
    if astype is None:
        astype = np.uint8
    write(path, np.asarray(volume).astype(astype), affine)
    print('Volume saved to {} as type {}.'.format(path, astype))
