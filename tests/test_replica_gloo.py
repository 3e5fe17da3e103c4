"""Leader / follower replication on CPU: N processes, torch.distributed (gloo) as the log's transport (host callbacks), the
sequential host model as compute.  The leader runs the band chain and logs every sweep; the followers apply the log and count
the sweeps assigned to them.  Labels, `segmented` order, the whole trace (sums included, bit for bit) and the result must equal
the single-process run on EVERY rank.  (On the GPU box the same driver runs over hipIpc with N processes on one GPU and over
RCCL: tests/test_gpu_parity.py.)
"""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT
from test_slabs_gloo import HM, free_port, _inputs


def _run(session, data, vmap, sweeps, calls=1):
    session.set_volume(data)
    session.set_labels(vmap)
    session.init(2.25)
    r = None
    for k in range(calls):                      # (vrg_run may be called again with a larger iterMax: the log goes on)
        r = session.run(sweeps * (k + 1) // calls, 10 ** 9, None)
    return dict(labels=session.labels(), seg=session.segmented(), tr=session.trace(), sweeps=r.sweeps, nseg=r.nseg,
                res=np.array([r.stop_reason, r.iter_num, r.nseg, r.n_in, r.n_out, r.ni, r.no, r.ties], np.int64), sums=np.array([r.sum_in, r.sum_out]))


def _worker(rank, world, port, sweeps, outdir, leader_verifies, options, calls):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from arterynetwork_amd import replica
    from arterynetwork_amd._capi import VrgLib
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lib = VrgLib(HM, 'vrgm_')
    data, vmap = _inputs()
    s = replica.make_replica_session(data.shape, rank, world, lib=lib, transport='callback', leader_verifies=leader_verifies, options=options)
    out = _run(s, data, vmap, sweeps, calls)
    st = s.repl_stats()
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), labels=out['labels'], seg=out['seg'], tr=out['tr'], res=out['res'], sums=out['sums'],
             stats=np.array([st['batches'], st['records'], st['sweeps'], st['verified'], st['verifiers'], st['slot']], np.int64))
    s.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,leader_verifies,options,calls', [
    (2, True, {}, 1), (2, False, {}, 1), (3, True, {'batch': 3}, 2), (8, False, {'batch': 4}, 1),
    (3, False, {'verify_every': 4}, 1), (2, True, {'verify_every': 0, 'batch': 5}, 1), (4, True, {'fused': 0, 'batch': 2}, 1)])
def test_replicas_equal_single_process(tmp_path, world, leader_verifies, options, calls):
    from arterynetwork_amd._capi import Session, VrgLib
    import subprocess
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'tests', 'hostmodel'), '-s', 'libvrg_hostmodel.so'])
    sweeps = 14
    data, vmap = _inputs()
    ref_s = Session(data.shape, lib=VrgLib(HM, 'vrgm_'))
    for k, v in options.items():
        ref_s.set_option(k, v)
    ref = _run(ref_s, data, vmap, sweeps, calls)
    assert ref['sweeps'] == sweeps // calls + (sweeps % calls if calls > 1 else 0) or calls > 1
    assert ref['nseg'] > 100
    mp.spawn(_worker, args=(world, free_port(), sweeps, str(tmp_path), leader_verifies, options, calls), nprocs=world, join=True)
    every = options.get('verify_every', 1)
    verified_total = 0
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r))
        assert np.array_equal(z['labels'], ref['labels']), r
        assert np.array_equal(z['seg'], ref['seg']), r
        assert np.array_equal(z['res'], ref['res']), (r, z['res'], ref['res'])
        for f in ('nflip', 'nseg', 'n_in', 'n_out', 'ni', 'no', 'ties', 'near_ties'):
            assert np.array_equal(z['tr'][f], ref['tr'][f]), (r, f)
        # the sums: bit for bit where the single process has them (verify_every leaves sweeps out in both)
        for f in ('sum_in', 'sum_out'):
            a, b = z['tr'][f], ref['tr'][f]
            assert np.array_equal(np.isnan(a), np.isnan(b)), (r, f, a, b)
            assert np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)]), (r, f)
        if every != 0:
            assert np.array_equal(z['sums'], ref['sums']) or np.isnan(ref['sums']).all()
        st = z['stats']
        assert st[2] == sweeps                                       # every rank saw every sweep of the log
        assert st[4] == (world if leader_verifies else world - 1)
        if r > 0:
            verified_total += st[3]
        if r == 0:
            assert st[5] == (0 if leader_verifies else -1)
    due = sweeps if every == 1 else (0 if every == 0 else sweeps // every)
    if not leader_verifies:
        assert verified_total >= due                                 # (+1 when the run's last sweep is counted after all)


def _worker_stream(rank, world, port, outdir, options, expect_error):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from arterynetwork_amd import replica
    from arterynetwork_amd._capi import VrgLib, VrgError
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lib = VrgLib(HM, 'vrgm_')
    data, vmap = _inputs()
    s = replica.make_replica_session(data.shape, rank, world, lib=lib, transport='callback', leader_verifies=False, options=options)
    err = ''
    try:
        _run(s, data, vmap, 14)
    except VrgError as e:
        err = str(e)
    st = s.repl_stats()
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), err=np.str_(err), stats=np.array([st['batches'], st['chunks'], st['sweeps']], np.int64), labels=s.labels() if not err else np.zeros(1))
    s.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('stream', [1, 0])
def test_log_travels_in_chunks(tmp_path, stream):
    """Per-sweep streaming: with trips enqueued four at a time the log of 14 sweeps travels in more chunks than batches (the band chain
    publishes a sweep as soon as its records and header are complete); with repl_stream = 0 every batch is one chunk.  Same labels."""
    mp.spawn(_worker_stream, args=(3, free_port(), str(tmp_path), {'batch': 4, 'repl_stream': stream, 'small_flips': 4096, 'fuse_max': 128}, False), nprocs=3, join=True)
    z = [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r)) for r in range(3)]
    for r in range(3):
        assert str(z[r]['err']) == '', (r, str(z[r]['err']))
        assert z[r]['stats'][2] == 14
        assert np.array_equal(z[r]['labels'], z[0]['labels'])
        if stream:
            assert z[r]['stats'][1] > z[r]['stats'][0], (r, z[r]['stats'])          # chunks > batches
        else:
            assert z[r]['stats'][1] == z[r]['stats'][0], (r, z[r]['stats'])
    assert np.array_equal(z[0]['stats'], z[1]['stats']) and np.array_equal(z[1]['stats'], z[2]['stats'])


@pytest.mark.parametrize('fault', [2, -3])
def test_a_failing_rank_ends_the_run_for_every_rank(tmp_path, fault):
    """A replicated vrg_run is collective, and so are its failures (round-5 advice): a leader that fails on the host side while the
    followers wait for the log (fault > 0: when it opens its 2nd batch), or a follower that cannot use a chunk (fault < 0: its 3rd),
    must not leave the other ranks waiting - every rank returns an error."""
    mp.spawn(_worker_stream, args=(3, free_port(), str(tmp_path), {'batch': 4, 'repl_fault': fault, 'small_flips': 4096, 'fuse_max': 128}, True), nprocs=3, join=True)
    for r in range(3):
        z = np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r))
        assert str(z['err']) != '', 'rank %d returned without an error' % r
