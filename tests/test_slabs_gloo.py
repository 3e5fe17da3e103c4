"""N > 1 path on CPU: two processes, torch.distributed (gloo), the sequential host model as compute.

Each rank recounts its own Z-slab; the per-sweep region statistics are summed through the reduce
callback (dist.all_reduce).  Labels, band lists, densities and traces must equal the single-process
run bit for bit on every rank.  (On the GPU box the same driver uses RCCL: tests/test_gpu_parity.py.)
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT

HM = os.path.join(ROOT, 'tests', 'hostmodel', 'libvrg_hostmodel.so')


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs():
    sys.path.insert(0, ROOT)
    from arterynetwork_amd import phantoms
    data, vmap = phantoms.tube_phantom(shape=(40, 32, 21), radius=3.0, seed=4, seed_planes=3, amp_y=6.0, amp_z=3.0,
                                       levels=16, brain_mask=True)
    return data, vmap


def _run(session, data, vmap, sweeps):
    session.set_volume(data)
    session.set_labels(vmap)
    session.init(2.25)
    r = session.run(sweeps, 10 ** 9, None)
    return dict(labels=session.labels(), seg=session.segmented(), tr=session.trace(),
                b0=session.band(0), b1=session.band(1), sweeps=r.sweeps, nseg=r.nseg)


def _worker(rank, world, port, sweeps, outdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from arterynetwork_amd import slabs
    from arterynetwork_amd._capi import VrgLib
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lib = VrgLib(HM, 'vrgm_')
    data, vmap = _inputs()
    s = slabs.make_slab_session(data.shape, rank, world, lib=lib, reduce='callback')
    out = _run(s, data, vmap, sweeps)
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), labels=out['labels'], seg=out['seg'], tr=out['tr'],
             ip=np.concatenate((out['b0'][1], out['b1'][1])), op=np.concatenate((out['b0'][2], out['b1'][2])),
             idx=np.concatenate((out['b0'][0], out['b1'][0])), slab=np.asarray(s.slab))
    s.close()
    dist.barrier()
    dist.destroy_process_group()


def test_partition():
    from arterynetwork_amd.slabs import partition
    assert partition(640, 8) == [(80 * i, 80 * i + 80) for i in range(8)]
    p = partition(170, 8)
    assert p[0][0] == 0 and p[-1][1] == 170 and all(a[1] == b[0] for a, b in zip(p, p[1:]))
    assert max(b - a for a, b in p) - min(b - a for a, b in p) <= 1


@pytest.mark.parametrize('world', [2, 3, 8])
def test_slabs_equal_single_process(tmp_path, world):
    from arterynetwork_amd._capi import Session, VrgLib
    import subprocess
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'tests', 'hostmodel'), '-s', 'libvrg_hostmodel.so'])
    sweeps = 12
    data, vmap = _inputs()
    ref = _run(Session(data.shape, lib=VrgLib(HM, 'vrgm_')), data, vmap, sweeps)
    assert ref['sweeps'] == sweeps and ref['nseg'] > 100
    mp.spawn(_worker, args=(world, free_port(), sweeps, str(tmp_path)), nprocs=world, join=True)
    slabs_seen = []
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r))
        assert np.array_equal(z['labels'], ref['labels'])
        assert np.array_equal(z['seg'], ref['seg'])
        assert np.array_equal(z['idx'], np.concatenate((ref['b0'][0], ref['b1'][0])))
        assert np.array_equal(z['ip'], np.concatenate((ref['b0'][1], ref['b1'][1])))
        assert np.array_equal(z['op'], np.concatenate((ref['b0'][2], ref['b1'][2])))
        for f in ('nflip', 'nseg', 'n_in', 'n_out', 'ni', 'no'):
            assert np.array_equal(z['tr'][f], ref['tr'][f]), f
        np.testing.assert_allclose(z['tr']['sum_in'], ref['tr']['sum_in'], rtol=1e-12)
        slabs_seen.append(tuple(z['slab']))
    assert slabs_seen[0][0] == 0 and slabs_seen[-1][1] == data.shape[2]
