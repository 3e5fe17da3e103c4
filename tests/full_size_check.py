#!/usr/bin/env python3
"""Full-size property checks of the HIP path; run by tests/test_gpu_parity.py in their own process.

default      880x880x640 (the headline size), the torch-generated bench volume, 60 sweeps
--oracle3    the same volume, 100 sweeps, against the ORACLE at full size (all cores; ~25 GB of host memory, a few minutes):
             labels of all 495 616 000 voxels, both band list orders and densities, `segmented` order, whole trace
--config5    1024^3 with 16-bit intensity storage (BASELINE configs[4] on one GPU), 40 sweeps, and the same volume with
             fp32 storage: labels, `segmented` and the whole trace (incl. the f64 intensity sums) must be identical
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch                      # before the HIP library: see INTEGRATION.md (one ROCm runtime per process)
from arterynetwork_amd import phantoms
from arterynetwork_amd._capi import Session


def check_invariants(labels):
    seg = labels <= 1
    pad = np.pad(labels, 1, constant_values=255)
    nx, ny, nz = labels.shape
    bad = 0
    any_nonseg = np.zeros(labels.shape, bool)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                if dx == dy == dz == 0:
                    continue
                nb = pad[1 + dx:1 + dx + nx, 1 + dy:1 + dy + ny, 1 + dz:1 + dz + nz]
                bad += int(np.count_nonzero(seg & ((nb == 3) | (nb == 4))))
                any_nonseg |= (nb >= 2) & (nb <= 4)
    assert bad == 0
    assert not np.any((labels == 0) & any_nonseg)


def run_and_check(shape, I, vm, dev, sweeps, storage16):
    s = Session(shape)
    if storage16:
        s.set_option('storage16', 1)
    s.set_option('batch', 32)
    s.set_volume_ptr(I.data_ptr(), np.float32, list(I.stride()))
    s.set_labels_ptr(vm.data_ptr(), np.uint8, list(vm.stride()))
    s.init(2.25)
    r = s.run(sweeps, 10 ** 12, None)
    assert r.sweeps == sweeps and r.stop_reason == 4
    tr = s.trace()
    assert np.all(np.diff(tr['nseg']) >= 0) and tr['nseg'][-1] > 1.5 * tr['nseg'][0]
    vals, hin, hout, rin, rout = s.levels()
    assert np.array_equal(hin, rin) and np.array_equal(hout, rout)        # incremental histograms == dense recount
    assert int(hin.sum()) == tr['n_in'][-1] and int(hout.sum()) == tr['n_out'][-1]
    lab = torch.empty((shape[2], shape[1], shape[0]), dtype=torch.uint8, device=dev).permute(2, 1, 0)
    s._check(s.lib.get_labels(s._h, lab.data_ptr(), 0, (ctypes.c_int64 * 3)(*lab.stride())))
    torch.cuda.synchronize()
    seg = s.segmented()
    assert int((lab <= 1).sum()) == tr['n_in'][-1] == len(seg)
    assert int(((lab == 2) | (lab == 3)).sum()) == tr['n_out'][-1]
    assert int((lab == 1).sum()) == tr['ni'][-1] and int((lab == 2).sum()) == tr['no'][-1]
    assert abs(float(I[lab <= 1].double().sum()) - tr['sum_in'][-1]) <= 1e-9 * abs(tr['sum_in'][-1])
    lo = np.maximum(seg.min(0) - 3, 0)
    hi = np.minimum(seg.max(0) + 4, shape)
    crop = lab[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]].contiguous().cpu().numpy()
    check_invariants(crop)
    bands = (s.band(0), s.band(1))
    s.close()
    return lab, seg, tr, bands


def oracle_full_size(dev, sweeps=100):
    import time
    import parity
    from oracle import vrg_oracle as O
    shape = (880, 880, 640)
    I, vm = phantoms.bench_volume_torch(shape, dev)
    torch.cuda.synchronize()
    s = Session(shape)
    s.set_option('batch', 32)
    s.set_volume_ptr(I.data_ptr(), np.float32, list(I.stride()))
    s.set_labels_ptr(vm.data_ptr(), np.uint8, list(vm.stride()))
    s.init(2.25)
    r = s.run(sweeps, 10 ** 12, None)
    assert r.sweeps == sweeps
    t0 = time.time()
    Ih = np.ascontiguousarray(I.cpu().numpy(), dtype=np.float64)        # logical (x,y,z), C order for the oracle
    vh = np.ascontiguousarray(vm.cpu().numpy())
    o = O.Oracle(Ih, vh, 2.25, density_mode=1, omp=True)
    del Ih
    o.init()
    t1 = time.time()
    for _ in range(sweeps):
        assert o.step(10 ** 6, 10 ** 12, -1.0) == 0
    t2 = time.time()
    lab = np.empty(shape, np.uint8)
    s.labels(out=lab)
    assert np.array_equal(lab, o.labels()), 'labels differ'
    assert np.array_equal(parity.lex_of(s.segmented(), shape), o.segmented_lex()), 'segmented order differs'
    for which in (0, 1):
        co, ip, op = s.band(which)
        oi, oip, oop = o.band(which)
        assert np.array_equal(parity.lex_of(co, shape), oi), 'band list %d order differs' % which
        parity.assert_probs_close(ip, oip, 1e-9, 'innerProb list %d' % which)
        parity.assert_probs_close(op, oop, 1e-9, 'outerProb list %d' % which)
    tr, otr = s.trace(), o.trace()
    for f in ('nflip', 'nseg', 'n_in', 'n_out', 'ni', 'no'):
        assert np.array_equal(tr[f], otr[f]), f
    np.testing.assert_allclose(tr['sum_in'], otr['sum_in'], rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(tr['sum_out'], otr['sum_out'], rtol=1e-9, atol=1e-6)
    print('ORACLE3 OK: 880x880x640 x %d sweeps identical to the oracle (nseg %d -> %d, band %d); oracle init %.0f s, sweeps %.0f s'
          % (sweeps, tr['nseg'][0], tr['nseg'][-1], tr['ni'][-1] + tr['no'][-1], t1 - t0, t2 - t1))
    s.close(); o.close()


def main():
    dev = torch.device('cuda', 0)
    if '--oracle3' in sys.argv:
        oracle_full_size(dev)
        return
    if '--config5' in sys.argv:
        shape = (1024, 1024, 1024)
        I, vm = phantoms.bench_volume_torch(shape, dev, seed=5)
        torch.cuda.synchronize()
        lab16, seg16, tr16, b16 = run_and_check(shape, I, vm, dev, 40, True)
        lab32, seg32, tr32, b32 = run_and_check(shape, I, vm, dev, 40, False)
        assert bool((lab16 == lab32).all()) and np.array_equal(seg16, seg32)
        assert tr16.tobytes() == tr32.tobytes()                       # integer trace and f64 sums: bit-identical
        for a, b in zip(b16, b32):
            for x, y in zip(a, b):
                assert np.array_equal(x, y)
        print('CONFIG5 OK: 1024^3, storage16 == fp32 storage, nseg %d -> %d in 40 sweeps, band %d'
              % (tr16['nseg'][0], tr16['nseg'][-1], tr16['ni'][-1] + tr16['no'][-1]))
        return
    shape = (880, 880, 640)
    I, vm = phantoms.bench_volume_torch(shape, dev)
    torch.cuda.synchronize()
    lab, seg, tr, _ = run_and_check(shape, I, vm, dev, 60, False)
    assert tr['nseg'][-1] > 2 * tr['nseg'][0]
    print('FULL SIZE OK: nseg %d -> %d in 60 sweeps, band %d' % (tr['nseg'][0], tr['nseg'][-1], tr['ni'][-1] + tr['no'][-1]))


if __name__ == '__main__':
    main()
