"""NIfTI-1 file contract on either side of the VRG stage (generateVesselVolume.py:15-40,:65-84)."""
import gzip
import struct

import numpy as np
import pytest

from arterynetwork_amd import nifti


def test_round_trip_gz_and_plain(tmp_path, capsys):
    rng = np.random.default_rng(0)
    vol = rng.integers(0, 4000, size=(7, 9, 5)).astype(np.int16)
    aff = np.array([[0.5, 0, 0, -10.0], [0, 0.6, 0, 4.0], [0, 0, 0.7, 30.0], [0, 0, 0, 1.0]])
    for name in ('a.nii.gz', 'b.nii'):
        nifti.saveVolume(vol, aff, str(tmp_path / name), astype=np.int16)
        v2, a2 = nifti.loadVolume(str(tmp_path), name)
        assert v2.dtype == np.int16 and np.array_equal(v2, vol)
        assert v2.flags.f_contiguous                     # x fastest, as the GPU layout wants it
        np.testing.assert_allclose(a2, aff, atol=1e-6)
    out = capsys.readouterr().out
    assert 'Volume saved to' in out and 'Volume loaded from' in out and 'with shape = (7, 9, 5)' in out


def test_default_save_type_is_uint8(tmp_path):
    seg = np.zeros((4, 5, 6), np.int64)
    seg[1:3, 2:4, 1:5] = 1
    nifti.saveVolume(seg, np.eye(4), str(tmp_path / 'vesselVolumeMask.nii.gz'))
    v, a = nifti.loadVolume(str(tmp_path), 'vesselVolumeMask.nii.gz')
    assert v.dtype == np.uint8 and np.array_equal(v, seg)
    raw = gzip.open(str(tmp_path / 'vesselVolumeMask.nii.gz')).read()
    assert struct.unpack('<i', raw[:4])[0] == 348 and raw[344:347] == b'n+1'
    assert struct.unpack('<8h', raw[40:56])[:4] == (3, 4, 5, 6)
    assert struct.unpack('<2h', raw[70:74]) == (2, 8)
    assert len(raw) == 352 + 4 * 5 * 6


def test_reads_hand_written_file_with_scaling_and_qform(tmp_path):
    """A file nobody here wrote: big-endian float32, scl_slope/inter, qform only (90 degree rotation about z)."""
    shape = (3, 4, 2)
    data = np.arange(24, dtype=np.float32).reshape(shape, order='F')
    hdr = bytearray(352)
    struct.pack_into('>i', hdr, 0, 348)
    struct.pack_into('>8h', hdr, 40, 3, 3, 4, 2, 1, 1, 1, 1)
    struct.pack_into('>2h', hdr, 70, 16, 32)
    struct.pack_into('>8f', hdr, 76, 1.0, 2.0, 3.0, 4.0, 0, 0, 0, 0)
    struct.pack_into('>3f', hdr, 108, 352.0, 2.0, 1.0)            # value = 2 * stored + 1
    struct.pack_into('>2h', hdr, 252, 1, 0)
    s = np.sqrt(0.5)
    struct.pack_into('>6f', hdr, 256, 0.0, 0.0, s, 5.0, 6.0, 7.0)
    hdr[344:348] = b'n+1\x00'
    p = tmp_path / 'q.nii'
    p.write_bytes(bytes(hdr) + data.astype('>f4').tobytes(order='F'))
    v, aff, h = nifti.read(str(p))
    assert v.dtype == np.float64 and np.allclose(v, 2 * data + 1)
    expect = np.array([[0, -3.0, 0, 5.0], [2.0, 0, 0, 6.0], [0, 0, 4.0, 7.0], [0, 0, 0, 1.0]])
    np.testing.assert_allclose(aff, expect, atol=1e-5)


def test_value_map_policy():
    from arterynetwork_amd.refine import build_value_map
    data = np.arange(27, dtype=np.float32).reshape(3, 3, 3)
    seed = np.zeros((3, 3, 3), np.uint8); seed[1, 1, 1] = 1
    brain = np.ones((3, 3, 3), np.uint8); brain[0] = 0
    vm = build_value_map(data, seed, brain, exclude_below=10)
    assert vm[1, 1, 1] == 0 and (vm[0] == 4).all() and vm[1, 0, 0] == 4 and vm[2, 2, 2] == 3
    assert set(np.unique(vm)) <= {0, 3, 4}


def test_bad_file(tmp_path):
    p = tmp_path / 'x.nii'
    p.write_bytes(b'not a nifti')
    with pytest.raises(ValueError):
        nifti.read(str(p))
