#!/usr/bin/env python3
"""Full-size property checks of the HIP path; run by tests/test_gpu_parity.py in their own process.

default      880x880x640 (the headline size), the torch-generated bench volume, 60 sweeps
--oracle3    the same volume, 100 sweeps, against the ORACLE at full size (all cores; ~25 GB of host memory, a few minutes):
             labels of all 495 616 000 voxels, both band list orders and densities, `segmented` order, whole trace
--slabs W NXxNYxNZ SWEEPS
             BASELINE configs[3]'s partition at full size on ONE GPU: W ranks (processes) share GPU 0, each recounts its own
             Z-slab of the same bench volume (host-callback reduction over gloo); every rank's labels, `segmented` and
             integer trace must equal the single-process run, and the ranks' slab counts must add up to the incremental
             region sizes of every sweep
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


def oracle_full_size(dev, sweeps=100, shape=(880, 880, 640), storage16=False, seed=3, tag='ORACLE3'):
    import time
    import parity
    from oracle import vrg_oracle as O
    I, vm = phantoms.bench_volume_torch(shape, dev, seed=seed)
    torch.cuda.synchronize()
    s = Session(shape)
    s.set_option('batch', 32)
    if storage16:
        s.set_option('storage16', 1)
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
    print('%s OK: %s%s x %d sweeps identical to the oracle (nseg %d -> %d, band %d); oracle init %.0f s, sweeps %.0f s'
          % (tag, 'x'.join(map(str, shape)), ' (16-bit storage)' if storage16 else '', sweeps, tr['nseg'][0], tr['nseg'][-1], tr['ni'][-1] + tr['no'][-1], t1 - t0, t2 - t1))
    s.close(); o.close()


def label_digest(s, shape, dev):
    """sha256 of the uint8 label volume (x-fastest layout), fetched through the C-ABI into a device buffer."""
    import hashlib
    lab = torch.empty((shape[2], shape[1], shape[0]), dtype=torch.uint8, device=dev).permute(2, 1, 0)
    s._check(s.lib.get_labels(s._h, lab.data_ptr(), 0, (ctypes.c_int64 * 3)(*lab.stride())))
    torch.cuda.synchronize()
    return hashlib.sha256(lab.permute(2, 1, 0).contiguous().cpu().numpy().tobytes()).hexdigest(), lab


def slab_worker(rank, world, port, shape, sweeps, outdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from arterynetwork_amd import slabs
    dev = torch.device('cuda', 0)                          # every rank on GPU 0
    dist.init_process_group('gloo', rank=rank, world_size=world)
    I, vm = phantoms.bench_volume_torch(shape, dev)
    torch.cuda.synchronize()
    seen = []
    s = slabs.make_slab_session(shape, rank, world, device=0, reduce='callback', observer=lambda p, t: seen.append(p + t))
    s.set_option('batch', 16)
    s.set_volume_ptr(I.data_ptr(), np.float32, list(I.stride()))
    s.set_labels_ptr(vm.data_ptr(), np.uint8, list(vm.stride()))
    s.init(2.25)
    r = s.run(sweeps, 10 ** 12, None)
    assert r.sweeps == sweeps, (r.sweeps, r.stop_reason)
    digest, _ = label_digest(s, shape, dev)
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), digest=np.str_(digest), seg=s.segmented(), tr=s.trace(),
             seen=np.asarray(seen, np.float64), slab=np.asarray(s.slab), dense_bytes=np.int64(s.stats()['dense_bytes']))
    s.close()
    dist.barrier()
    dist.destroy_process_group()


def slabs_one_gpu(world, shape, sweeps):
    """Parent: the single-process run, then `world` rank processes on the same GPU; compare."""
    import subprocess
    import tempfile
    import socket
    dev = torch.device('cuda', 0)
    I, vm = phantoms.bench_volume_torch(shape, dev, tubes=int(os.environ.get('VRG_CHECK_TUBES', '1')))     # (env: several tubes = thousands of flips per sweep)
    torch.cuda.synchronize()
    s = Session(shape)
    s.set_option('batch', 16)
    s.set_volume_ptr(I.data_ptr(), np.float32, list(I.stride()))
    s.set_labels_ptr(vm.data_ptr(), np.uint8, list(vm.stride()))
    s.init(2.25)
    r = s.run(sweeps, 10 ** 12, None)
    assert r.sweeps == sweeps
    ref_digest, _ = label_digest(s, shape, dev)
    ref_seg, ref_tr, ref_bytes = s.segmented(), s.trace(), s.stats()['dense_bytes']
    s.close()
    del I, vm
    torch.cuda.empty_cache()
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    outdir = tempfile.mkdtemp(prefix='slabs_')
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--slab-worker', str(rk), str(world), str(port),
                               'x'.join(map(str, shape)), str(sweeps), outdir]) for rk in range(world)]
    rcs = [p.wait() for p in procs]
    assert all(rc == 0 for rc in rcs), rcs
    planes = []
    seen_sum = None
    nbytes = 0
    for rk in range(world):
        z = np.load(os.path.join(outdir, 'rank%d.npz' % rk))
        assert str(z['digest']) == ref_digest, 'rank %d: labels differ from the single-process run' % rk
        assert np.array_equal(z['seg'], ref_seg), 'rank %d: segmented differs' % rk
        for f in ('nflip', 'nseg', 'n_in', 'n_out', 'ni', 'no', 'ties'):
            assert np.array_equal(z['tr'][f], ref_tr[f]), (rk, f)
        np.testing.assert_allclose(z['tr']['sum_in'], ref_tr['sum_in'], rtol=1e-12)
        np.testing.assert_allclose(z['tr']['sum_out'], ref_tr['sum_out'], rtol=1e-12)
        seen = z['seen']
        assert seen.shape == (sweeps + 1, 8), seen.shape           # one reduction at init, one per sweep
        # what every rank got back is the same total, and it is the incremental region size of that sweep
        assert np.array_equal(seen[:, 4], ref_tr['n_in'].astype(np.float64)) and np.array_equal(seen[:, 5], ref_tr['n_out'].astype(np.float64))
        seen_sum = seen[:, :4].copy() if seen_sum is None else seen_sum + seen[:, :4]
        planes.append(int(z['slab'][1] - z['slab'][0]))
        nbytes += int(z['dense_bytes'])
    # the ranks' slab counts add up to the region sizes, sweep by sweep (exact: integers below 2^53)
    assert np.array_equal(seen_sum[:, 0], ref_tr['n_in'].astype(np.float64)) and np.array_equal(seen_sum[:, 1], ref_tr['n_out'].astype(np.float64))
    np.testing.assert_allclose(seen_sum[:, 2], ref_tr['sum_in'], rtol=1e-12)
    assert sum(planes) == shape[2] and max(planes) - min(planes) <= 1
    # the slabs' dense passes fetch what the single pass fetches, plus at most the two class-word units a face cuts
    assert ref_bytes <= nbytes <= ref_bytes + world * 2 * (256 + 4 * 1024 * 4)
    print('SLABS OK: %s, %d ranks on one GPU (slabs of %s planes), %d sweeps: labels / segmented / trace identical on every rank, '
          'slab counts add up every sweep (nseg %d -> %d)' % ('x'.join(map(str, shape)), world, sorted(set(planes)), sweeps, ref_tr['nseg'][0], ref_tr['nseg'][-1]))


def replica_worker(rank, world, port, shape, sweeps, outdir, transport, leader_verifies):
    """One rank of a leader / follower group, every rank on GPU 0 (own process, own HIP context)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from arterynetwork_amd import replica
    dev = torch.device('cuda', 0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    I, vm = phantoms.bench_volume_torch(shape, dev, tubes=int(os.environ.get('VRG_CHECK_TUBES', '1')))
    torch.cuda.synchronize()
    fault = int(os.environ.get('VRG_CHECK_FAULT', '0'))     # (option repl_fault: a rank fails on the host side in the middle of the run - every rank must return an error)
    s = replica.make_replica_session(shape, rank, world, device=0, transport=transport, leader_verifies=bool(leader_verifies), options={'batch': 16, 'repl_fault': fault})
    s.set_volume_ptr(I.data_ptr(), np.float32, list(I.stride()))
    s.set_labels_ptr(vm.data_ptr(), np.uint8, list(vm.stride()))
    s.init(2.25)
    if fault:
        from arterynetwork_amd._capi import VrgError
        err = ''
        try:
            s.run(sweeps, 10 ** 12, None)
        except VrgError as e:
            err = str(e)
        np.savez(os.path.join(outdir, 'rank%d.npz' % rank), err=np.str_(err))
        s.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    r = s.run(sweeps // 2, 10 ** 12, None)                  # two calls: the log goes on where the first run ended
    r2 = s.run(sweeps, 10 ** 12, None)
    assert r.sweeps + r2.sweeps == sweeps, (r.sweeps, r2.sweeps, r2.stop_reason)
    digest, _ = label_digest(s, shape, dev)
    st = s.repl_stats()
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), digest=np.str_(digest), seg=s.segmented(), tr=s.trace(), transport=np.str_(s.replica['transport']),
             stats=np.array([st['batches'], st['records'], st['sweeps'], st['verified'], st['verifiers'], st['slot']], np.int64),
             res=np.array([r2.stop_reason, r2.iter_num, r2.nseg, r2.n_in, r2.n_out, r2.ni, r2.no, r2.ties], np.int64))
    s.close()
    dist.barrier()
    dist.destroy_process_group()


def replicas_one_gpu(world, shape, sweeps, transport, leader_verifies):
    """Parent: the single-process run, then `world` rank processes of a leader / follower group on the same GPU; every rank's
    labels, `segmented` order, whole trace (intensity sums bit for bit) and result must equal the single process's."""
    import subprocess
    import tempfile
    import socket
    dev = torch.device('cuda', 0)
    I, vm = phantoms.bench_volume_torch(shape, dev, tubes=int(os.environ.get('VRG_CHECK_TUBES', '1')))     # (env: several tubes = thousands of flips per sweep)
    torch.cuda.synchronize()
    s = Session(shape)
    s.set_option('batch', 16)
    s.set_volume_ptr(I.data_ptr(), np.float32, list(I.stride()))
    s.set_labels_ptr(vm.data_ptr(), np.uint8, list(vm.stride()))
    s.init(2.25)
    r = s.run(sweeps, 10 ** 12, None)
    assert r.sweeps == sweeps
    ref_digest, _ = label_digest(s, shape, dev)
    ref_seg, ref_tr = s.segmented(), s.trace()
    ref_res = np.array([r.stop_reason, r.iter_num, r.nseg, r.n_in, r.n_out, r.ni, r.no, r.ties], np.int64)
    s.close()
    del I, vm
    torch.cuda.empty_cache()
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    outdir = tempfile.mkdtemp(prefix='replicas_')
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--replica-worker', str(rk), str(world), str(port),
                               'x'.join(map(str, shape)), str(sweeps), outdir, transport, str(int(leader_verifies))]) for rk in range(world)]
    rcs = [p.wait(timeout=600) for p in procs]
    assert all(rc == 0 for rc in rcs), rcs
    if int(os.environ.get('VRG_CHECK_FAULT', '0')):
        errs = [str(np.load(os.path.join(outdir, 'rank%d.npz' % rk))['err']) for rk in range(world)]
        assert all(errs), 'a rank returned without an error: %r' % (errs,)
        print('REPLICAS FAIL TOGETHER: %d ranks on one GPU over %s, injected fault %s: every rank returned an error, none hung' % (world, transport, os.environ['VRG_CHECK_FAULT']))
        return
    counted = 0
    for rk in range(world):
        z = np.load(os.path.join(outdir, 'rank%d.npz' % rk))
        assert str(z['transport']) == transport, (rk, str(z['transport']))
        assert str(z['digest']) == ref_digest, 'rank %d: labels differ from the single-process run' % rk
        assert np.array_equal(z['seg'], ref_seg), 'rank %d: segmented differs' % rk
        assert z['tr'].tobytes() == ref_tr.tobytes(), 'rank %d: trace differs (integer fields or the bits of the intensity sums)' % rk
        assert np.array_equal(z['res'][:7], ref_res[:7]), (rk, z['res'], ref_res)
        st = z['stats']
        assert st[2] == sweeps and st[4] == (world if leader_verifies else world - 1)
        if rk > 0:
            counted += int(st[3])
            assert st[3] >= sweeps // st[4], (rk, st)                     # its share of the sweeps
    assert counted == (sweeps - (sweeps + world - 1) // world if leader_verifies else sweeps), (counted, sweeps)   # every sweep counted exactly once (the leader's share: sweeps 1, 1 + N, ...)
    print('REPLICAS OK: %s, %d ranks on one GPU over %s (leader %s), %d sweeps: labels / segmented / trace (sums bit for bit) / result identical '
          'on every rank; the followers counted %d sweeps (nseg %d -> %d)' % ('x'.join(map(str, shape)), world, transport,
          'counts a share' if leader_verifies else 'only leads', sweeps, counted, ref_tr['nseg'][0], ref_tr['nseg'][-1]))


def main():
    if '--replica-worker' in sys.argv:
        a = sys.argv[sys.argv.index('--replica-worker') + 1:]
        replica_worker(int(a[0]), int(a[1]), int(a[2]), tuple(int(v) for v in a[3].split('x')), int(a[4]), a[5], a[6], int(a[7]))
        return
    if '--replicas' in sys.argv:
        a = sys.argv[sys.argv.index('--replicas') + 1:]
        replicas_one_gpu(int(a[0]), tuple(int(v) for v in a[1].split('x')), int(a[2]), a[3], int(a[4]) if len(a) > 4 else 0)
        return
    if '--slab-worker' in sys.argv:
        a = sys.argv[sys.argv.index('--slab-worker') + 1:]
        slab_worker(int(a[0]), int(a[1]), int(a[2]), tuple(int(v) for v in a[3].split('x')), int(a[4]), a[5])
        return
    if '--slabs' in sys.argv:
        a = sys.argv[sys.argv.index('--slabs') + 1:]
        slabs_one_gpu(int(a[0]), tuple(int(v) for v in a[1].split('x')), int(a[2]))
        return
    dev = torch.device('cuda', 0)
    if '--oracle3' in sys.argv:                            # [sweeps]: BASELINE configs[2] states 500
        a = sys.argv[sys.argv.index('--oracle3') + 1:]
        oracle_full_size(dev, int(a[0]) if a and a[0].isdigit() else 100)
        return
    if '--config5-oracle' in sys.argv:                     # configs[4]'s family at a size the oracle runs: a 1024x1024x128 volume, 16-bit storage
        a = sys.argv[sys.argv.index('--config5-oracle') + 1:]
        oracle_full_size(dev, int(a[0]) if a and a[0].isdigit() else 100, shape=(1024, 1024, 128), storage16=True, seed=5, tag='CONFIG5-ORACLE')
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
