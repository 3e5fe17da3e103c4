"""Parity of the HIP path (libvrg_hip.so through the C-ABI) with the oracle, on a real MI355X.

Bit-exact: labels after every sweep, band list orders, `segmented` order, counts, stop reason and
iteration number.  Floating point (band densities innerProb/outerProb, region intensity sums):
relative 1e-9 (north_star allows 1e-5).  Exact mathematical ties of the sign test (:87) are outside
parity - the reference decides them by np.sum's rounding - and are detected and skipped.
"""
import ctypes
import io
import contextlib

import numpy as np
import pytest

import parity
from conftest import golden_names
from test_hostmodel import random_case

pytestmark = pytest.mark.gpu

SMALL = [n for n in golden_names() if n != 'config1_tube']


@pytest.fixture(scope='module')
def lib():
    from arterynetwork_amd._capi import product_lib
    return product_lib()


@pytest.mark.parametrize('variant', [0, 1])
@pytest.mark.parametrize('name', SMALL)
def test_goldens_stepwise(lib, golden_loader, name, variant):
    g = golden_loader(name)
    data, vmap = g.inputs()
    iterMax = g.max_sweeps if g.max_sweeps >= 0 else 200
    res, k = parity.run_stepwise(lib, data, vmap, g.H, g.maxSegmentSize, iterMax, density_mode=1,
                                 check_hist=True, options={'sweep_variant': variant})
    assert res is not None and k == g.ncalls - 1
    assert res.nseg == int(g.z['nseg'][-1])
    # and directly against the reference's own recorded outputs
    z = g.z
    from arterynetwork_amd._capi import Session
    s = Session(g.shape, lib=lib)
    s.set_volume(data); s.set_labels(vmap); s.init(g.H)
    s.run(iterMax, g.maxSegmentSize, None)
    assert np.array_equal(s.labels(), z['final_labels'])
    assert np.array_equal(parity.lex_of(s.segmented(), g.shape), z['final_segmented'])
    tr = s.trace()
    for f in ('nflip', 'nseg', 'n_in', 'n_out', 'ni', 'no'):
        assert np.array_equal(tr[f], z[f]), f
    s.close()


def test_reference_kats_through_the_drop_in_function(golden_loader):
    """The reference's two self-tests (:284-314) through the mirrored Python function."""
    from arterynetwork_amd import variationalRegionGrowing
    for name, it, n in (('kat_straight_line', 16, 80), ('kat_sphere', 11, 4169)):
        volume, valueMap = golden_loader(name).inputs()
        volume = volume.astype(int)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            segmented, segmentedMap, vm = variationalRegionGrowing(volume, valueMap)
        assert vm is valueMap and vm.dtype == np.int64
        assert all(volume[tuple(segmented.T)]) and np.count_nonzero(volume) == len(segmented) == n
        assert segmentedMap.dtype == np.int64 and segmentedMap.sum() == n
        assert buf.getvalue() == str(golden_loader(name).z['stdout'])
        assert 'Finished at iteration {}\n'.format(it) in buf.getvalue()


def test_config1_against_reference_trace(lib, golden_loader):
    """BASELINE.json configs[0]: 128x128x64 continuous-valued tube, 50 sweeps, vs the reference's record."""
    from arterynetwork_amd._capi import Session
    g = golden_loader('config1_tube')
    data, vmap = g.inputs()
    z = g.z
    s = Session(g.shape, lib=lib)
    s.set_volume(data); s.set_labels(vmap); s.init(g.H)
    snap = {int(t): k for k, t in enumerate(z['snap_iters'])}
    prob = {int(t): j for j, t in enumerate(z['prob_snaps'])}
    for it in sorted(snap):
        if it:
            s.run(it, g.maxSegmentSize, None)
        _, gi, go = g.snapshot(snap[it])
        ci, ip, op = s.band(0)
        co, ip2, op2 = s.band(1)
        assert np.array_equal(parity.lex_of(ci, g.shape), gi), it
        assert np.array_equal(parity.lex_of(co, g.shape), go), it
        _, gip, gop = g.probs(prob[it])
        # (about a million distinct values: the exact densities go through the bin moments - bound 2e-8, asserted at 1e-6)
        parity.assert_probs_close(np.concatenate((ip, ip2)), gip, parity.density_rtol(data), f'innerProb at {it}')
        parity.assert_probs_close(np.concatenate((op, op2)), gop, parity.density_rtol(data), f'outerProb at {it}')
    assert np.array_equal(s.labels(), z['final_labels'])
    assert np.array_equal(parity.lex_of(s.segmented(), g.shape), z['final_segmented'])
    tr = s.trace()
    for f in ('nflip', 'nseg', 'n_in', 'n_out', 'ni', 'no'):
        assert np.array_equal(tr[f], z[f]), f
    assert tr['nseg'][-1] == 2090 and tr['ni'][-1] + tr['no'][-1] == 4086
    s.close()


def test_random_volumes(lib):
    sweeps = done = 0
    for sd in range(120):
        I, vm, H, variant, dmode = random_case(sd)
        res, k = parity.run_stepwise(lib, I, vm, H, None, 40, density_mode=dmode, check_hist=True,
                                     options={'sweep_variant': variant})
        sweeps += k
        done += res is not None
    for sd in range(5000, 5030):
        I, vm, H, variant, dmode = random_case(sd, 8, 26)
        res, k = parity.run_stepwise(lib, I, vm, H, None, 30, density_mode=1, check_hist=True,
                                     options={'sweep_variant': variant})
        sweeps += k
        done += res is not None
    assert sweeps > 500
    assert done >= 140, 'too many cases cut short by an exact tie: {} of 150 completed'.format(done)


def test_random_volumes_in_one_call(lib):
    """Same cases, but the whole run in ONE vrg_run call: sweeps are enqueued in batches without host
    synchronisation and the dense pass of sweep k overlaps the band kernels of sweep k+1."""
    sweeps = done = 0
    for sd in range(300, 400):
        I, vm, H, variant, dmode = random_case(sd)
        res, k = parity.run_batched(lib, I, vm, H, None, 40, density_mode=dmode,
                                    options={'sweep_variant': variant, 'batch': 1 + sd % 7})
        sweeps += k
        done += res is not None
    for sd in range(6000, 6020):
        I, vm, H, variant, dmode = random_case(sd, 8, 26)
        res, k = parity.run_batched(lib, I, vm, H, None, 30, density_mode=1,
                                    options={'sweep_variant': variant, 'batch': 8,
                                             'storage16': sd % 2, 'small_flips': (4096, 0, 6)[sd % 3]})
        sweeps += k
        done += res is not None
    assert sweeps > 400
    assert done >= 110, 'too many cases cut short by an exact tie: {} of 120 completed'.format(done)


def test_integer_class_goldens_and_tie_rate(lib, golden_loader):
    """The reference's own input class (:284-314: binary integer volumes).  (1) The seven integer fixtures the REAL reference
    produced (the two KATs, KAT shapes with 5 % salt noise, two touching tubes, a torus, a three-level phantom): whole run
    in one call, ties == 0, every label / list order / count equal to the reference's record.  (2) How often exact ties
    occur on this class: random binary / few-level integer volumes, stepwise against the oracle; a case is cut at its first
    tie (parity.run_stepwise) - the rate goes to gpurun_out/ties_integer.json (committed as profiles/r04_ties_integer.json)."""
    import json
    import os
    from arterynetwork_amd._capi import Session
    from conftest import ROOT
    rep = {'fixtures': {}, 'random': {}}
    for name in ('kat_straight_line', 'kat_sphere', 'int_line_salt', 'int_sphere_salt', 'int_two_tubes', 'int_torus', 'int_three_level'):
        g = golden_loader(name)
        data, vmap = g.inputs()
        z = g.z
        s = Session(g.shape, lib=lib)
        s.set_volume(data.astype(np.int16)); s.set_labels(vmap); s.init(g.H)
        r = s.run(g.max_sweeps if g.max_sweeps >= 0 else 200, g.maxSegmentSize, None)
        assert r.ties == 0, (name, r.ties)
        assert np.array_equal(s.labels(), z['final_labels']), name
        assert np.array_equal(parity.lex_of(s.segmented(), g.shape), z['final_segmented']), name
        tr = s.trace()
        for f in ('nflip', 'nseg', 'n_in', 'n_out', 'ni', 'no'):
            assert np.array_equal(tr[f], z[f]), (name, f)
        k = len(z['snap_iters']) - 1
        _, gi, go = g.snapshot(k)
        assert np.array_equal(parity.lex_of(s.band(0)[0], g.shape), gi) and np.array_equal(parity.lex_of(s.band(1)[0], g.shape), go), name
        rep['fixtures'][name] = {'sweeps': int(r.sweeps), 'ties': int(r.ties), 'near_ties': int(r.near_ties), 'levels': int(len(np.unique(data)))}
        s.close()
    for levels in (2, 3, 5):
        cut = sweeps = 0
        n = 60
        for sd in range(n):
            rng = np.random.default_rng(7000 + 100 * levels + sd)
            shape = tuple(int(v) for v in rng.integers(6, 20, size=3))
            blob = rng.random(shape) < 0.35
            I = np.where(blob, levels - 1, 0) + (rng.random(shape) < 0.1) * rng.integers(0, levels, size=shape)
            I = np.clip(I, 0, levels - 1).astype(np.float64)
            vm = np.full(shape, 3, dtype=np.int64)
            vm[(rng.random(shape) < 0.08) & blob] = 0
            vm[rng.random(shape) > 0.85] = 4
            if not (vm == 0).any():
                vm.reshape(-1)[int(np.argmax(I.reshape(-1)))] = 0
            res, k = parity.run_stepwise(lib, I, vm, 2.25, None, 30, density_mode=1, check_hist=True)
            sweeps += k
            cut += res is None
        rep['random']['{}_levels'.format(levels)] = {'volumes': n, 'cut_short_by_a_tie': int(cut), 'sweeps_compared': int(sweeps)}
    d = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(d):
        json.dump(rep, open(os.path.join(d, 'ties_integer.json'), 'w'), indent=1)
    print(json.dumps(rep))


def test_verify_every(lib, golden_loader):
    """Option verify_every through the C-ABI on the GPU (the same case the host model runs), and through the drop-in function:
    identical results with the dense pass on every sweep, every 4th, never."""
    from test_hostmodel import _verify_every_case
    from arterynetwork_amd import variationalRegionGrowing
    _verify_every_case(lib, golden_loader)
    g = golden_loader('int_torus')
    data, vmap = g.inputs()
    outs = []
    for every in (1, 4, 0):
        vm = vmap.copy()
        tr = []
        seg, segMap, vm2 = variationalRegionGrowing(data, vm, H=g.H, maxSegmentSize=g.maxSegmentSize, quiet=True, trace=tr, verify_every=every)
        outs.append((seg, segMap, vm2.copy(), [t['nseg'] for t in tr]))
        assert not np.isnan(tr[-1]['sum_in'])                      # the last sweep is always counted
    for o in outs[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(o[:3], outs[0][:3])) and o[3] == outs[0][3]
    assert np.array_equal(outs[0][2], g.z['final_labels'].reshape(g.shape))
    with pytest.raises(ValueError):
        variationalRegionGrowing(data, vmap.copy(), quiet=True, verify_every=-1)


def test_binned_exact_densities(lib, golden_loader):
    """Large level tables evaluate the exact densities (:252-255) through bin moments (vrg_items.h; proved bound 2e-8).  On the
    continuous-valued fixtures of the REAL reference, with the bins forced on (bin_above = 0) and - config 1, a million distinct
    values - by default: labels after every sweep, both list orders, `segmented`, the integer trace identical to the oracle;
    densities within 1e-6; the moments equal to ones rebuilt from a dense recount bit for bit (s.levels()); and the measured
    deviation from the level sums of the same library goes to gpurun_out/binned_deviation.json."""
    from test_hostmodel import _binned_case
    _binned_case(lib, golden_loader, 'gpu')


def test_fused_trips_with_and_without_memo(lib, golden_loader):
    """Fused trips (k_band -> k_sweep) against the oracle with the per-level memo of the corrections forced on (option
    memo_above = 0: k_memo behind every fused sweep, what large bands get) and off, stepwise and in one call; the two must
    also agree with each other and with the four-launch chain (fused = 0) in every label and list, densities to 1e-12."""
    from arterynetwork_amd._capi import Session
    for name in ('tube_q_small', 'adv_noise_q', 'int_torus', 'int_three_level', 'kat_sphere'):
        g = golden_loader(name)
        data, vmap = g.inputs()
        iterMax = g.max_sweeps if g.max_sweeps >= 0 else 200
        outs = []
        for opts in ({'memo_above': 0}, {'memo_above': 1 << 30}, {'fused': 0}):
            res, k = parity.run_stepwise(lib, data, vmap, g.H, g.maxSegmentSize, iterMax, density_mode=1, check_hist=True, options=opts)
            assert res is not None and k == g.ncalls - 1, (name, opts)
            s = Session(g.shape, lib=lib)
            for kk, v in dict(opts, batch=7).items():
                s.set_option(kk, v)
            s.set_volume(data); s.set_labels(vmap); s.init(g.H)
            s.run(iterMax, g.maxSegmentSize, None)
            st = s.stats()
            if 'fused' in opts:
                assert st['fused_trips'] == 0
            else:
                assert st['fused_trips'] > 0 and (st['memo_trips'] > 0) == (opts['memo_above'] == 0), (name, opts, st)
            outs.append((s.labels(), s.segmented(), s.band(0), s.band(1)))
            s.close()
        for o in outs[1:]:
            assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1]), name
            for w in (2, 3):
                assert np.array_equal(o[w][0], outs[0][w][0]), name
                np.testing.assert_allclose(o[w][1], outs[0][w][1], rtol=1e-12, atol=1e-15); np.testing.assert_allclose(o[w][2], outs[0][w][2], rtol=1e-12, atol=1e-15)


def test_fused_sweeps_on_large_level_tables(lib):
    from test_hostmodel import _fused_large_table_case
    _fused_large_table_case(lib)


def test_skip_rule_closure_is_race_free(lib):
    """Regression: the skip-rule fix-point is computed by every workgroup of k_relabel for itself.  With a wrong
    termination test one workgroup could stop before another one's write became visible; this case then failed in
    6 % of its runs (labels of sweep 1).  300 repetitions."""
    I, vm, H, variant, dmode = random_case(200041, 10, 40)
    for _ in range(300):
        res, k = parity.run_stepwise(lib, I, vm, H, None, 2, density_mode=1, options={'sweep_variant': 0})
        assert k == 2


def test_whole_run_is_repeatable(lib):
    """The same medium case 100 times in one call each (sweeps batched, kernels of consecutive sweeps back to back,
    dense pass trailing): every run has to reproduce the oracle - a cheap trap for rare intra-kernel races."""
    I, vm, H, variant, dmode = random_case(200041, 10, 40)
    for rep in range(100):
        res, k = parity.run_batched(lib, I, vm, H, None, 6, density_mode=1, options={'sweep_variant': 0, 'batch': 1 + rep % 4})
        assert res is not None


def test_level_table_paths(lib):
    """Level tables from a few hundred to tens of thousands of distinct intensities (one per voxel, nearly): float noise
    volumes, stepwise and in one call."""
    for shape, sd in (((12, 13, 11), 1), ((30, 31, 29), 2), ((34, 33, 32), 3)):
        rng = np.random.default_rng(sd)
        I = rng.standard_normal(shape).astype(np.float32).astype(np.float64)
        u = rng.random(shape)
        vm = np.full(shape, 3, dtype=np.int64); vm[u < 0.1] = 0; vm[u > 0.8] = 4
        assert len(np.unique(I)) > 0.99 * I.size
        res, k = parity.run_stepwise(lib, I, vm, 2.25, None, 4, density_mode=1, check_hist=True)
        assert res is not None and k >= 3
        res, k = parity.run_batched(lib, I, vm, 2.25, None, 4, density_mode=1, options={'batch': 3})
        assert res is not None


def test_medium_tube_vs_oracle(lib):
    """A 160x96x64 integer-level tube with a brain mask, 60 sweeps: the oracle (level mode) takes seconds."""
    from arterynetwork_amd import phantoms
    data, vmap = phantoms.tube_phantom(shape=(160, 96, 64), radius=3.5, seed=9, seed_planes=3, amp_y=18.0,
                                       amp_z=9.0, levels=64, brain_mask=True)
    res, k = parity.run_stepwise(lib, data, vmap, 2.25, None, 60, density_mode=1, every=10, check_hist=True)
    assert res is not None and k == 60 and res.nseg > 1000
    for opts in ({'batch': 3}, {'storage16': 1, 'small_flips': 20}, {'capacity_floor': 64, 'batch': 5}):
        res, k = parity.run_batched(lib, data, vmap, 2.25, None, 60, density_mode=1, options=opts)
        assert res is not None and k == 60


def test_run_to_run_determinism(lib):
    from arterynetwork_amd import phantoms
    from arterynetwork_amd._capi import Session
    data, vmap = phantoms.scattered_seeds()
    outs = []
    for _ in range(2):
        s = Session(data.shape, lib=lib)
        s.set_volume(data); s.set_labels(vmap); s.init(2.25)
        s.run(15, 10 ** 9, None)
        outs.append((s.labels(), s.segmented(), s.band(0), s.band(1), s.trace()))
        s.close()
    a, b = outs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for w in (2, 3):
        for x, y in zip(a[w], b[w]):
            assert np.array_equal(x, y)          # densities bit-identical run to run
    assert a[4].tobytes() == b[4].tobytes()


def test_error_paths(lib):
    from arterynetwork_amd import variationalRegionGrowing
    with pytest.raises(ValueError):
        variationalRegionGrowing(np.zeros((4, 4, 4)), np.full((4, 4, 4), 3), quiet=True)   # no seeds (:48)
    with pytest.raises(ValueError):
        variationalRegionGrowing(np.zeros((4, 4, 4)), np.full((4, 4, 4), 2), quiet=True)   # bad label
    with pytest.raises(ValueError):
        variationalRegionGrowing(np.zeros((4, 4)), np.zeros((4, 4)), quiet=True)


def check_invariants(labels):
    """Invariants of the reference's state, via 26-neighbour shifts."""
    seg = labels <= 1
    pad = np.pad(labels, 1, constant_values=255)
    nx, ny, nz = labels.shape
    bad = 0
    any_nonseg_nbr = np.zeros(labels.shape, bool)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                if dx == dy == dz == 0:
                    continue
                nb = pad[1 + dx:1 + dx + nx, 1 + dy:1 + dy + ny, 1 + dz:1 + dz + nz]
                # a segmented voxel never has a label-3 or label-4 neighbour
                bad += int(np.count_nonzero(seg & ((nb == 3) | (nb == 4))))
                any_nonseg_nbr |= (nb >= 2) & (nb <= 4)
    assert bad == 0
    # label 0 = segmented with no non-segmented neighbour
    assert not np.any((labels == 0) & any_nonseg_nbr)


def test_config2_size_properties(lib):
    """512x512x170 (BASELINE configs[1] size): size-independent properties after 40 sweeps."""
    from arterynetwork_amd import phantoms
    from arterynetwork_amd._capi import Session
    shape = (512, 512, 170)
    data, vmap = phantoms.bench_volume(shape, seed=2)
    s = Session(shape, lib=lib)
    s.set_volume(data); s.set_labels(vmap.astype(np.uint8)); s.init(2.25)
    r = s.run(40, 10 ** 9, None)
    assert r.sweeps == 40 and r.stop_reason == 4
    lab = s.labels()
    tr = s.trace()
    assert tr['n_in'][-1] == np.count_nonzero(lab <= 1) == len(s.segmented())
    assert tr['n_out'][-1] == np.count_nonzero((lab == 2) | (lab == 3))
    assert tr['ni'][-1] == np.count_nonzero(lab == 1) and tr['no'][-1] == np.count_nonzero(lab == 2)
    assert np.all(np.diff(tr['nseg']) >= 0) and tr['nseg'][-1] > tr['nseg'][0]   # the front advances along the tube
    vals, hin, hout, rin, rout = s.levels()
    assert np.array_equal(hin, rin) and np.array_equal(hout, rout)          # incremental histograms == dense recount
    assert np.isclose(tr['sum_in'][-1], float(data[lab <= 1].astype(np.float64).sum()), rtol=1e-9)
    check_invariants(lab)
    # sweeping on is idempotent once converged / deterministic: the reference variant gives the same state
    s2 = Session(shape, lib=lib)
    s2.set_option('sweep_variant', 1)
    s2.set_volume(data); s2.set_labels(vmap.astype(np.uint8)); s2.init(2.25)
    s2.run(40, 10 ** 9, None)
    assert np.array_equal(s2.labels(), lab)
    for w in (0, 1):
        (co, ip, op), (co2, ip2, op2) = s.band(w), s2.band(w)
        assert np.array_equal(co, co2)                        # same lists in the same order
        np.testing.assert_allclose(ip, ip2, rtol=1e-12)       # (the check variant sums the corrections in another order)
        np.testing.assert_allclose(op, op2, rtol=1e-12)
    s.close(); s2.close()


def test_config2_full_size_vs_oracle(lib):
    """BASELINE configs[1] at FULL size: 512x512x170, 200 sweeps in ONE vrg_run call against the oracle (level mode,
    ~1 min of CPU): final labels, both band list orders and densities, `segmented` order, the whole trace and the
    incremental class histograms."""
    from arterynetwork_amd import phantoms
    data, vmap = phantoms.bench_volume((512, 512, 170), seed=2)
    res, k = parity.run_batched(lib, data, vmap.astype(np.uint8), 2.25, None, 200, density_mode=1, options={'batch': 64})
    assert res is not None and k == 200 and res.stop_reason == 4
    assert res.nseg > 5000


def test_config5_size_storage16():
    """BASELINE configs[4] on the one GPU of the box: 1024^3 with 16-bit intensity storage.  Runs
    tests/full_size_check.py --config5 in a fresh process: properties after 40 sweeps and bit-identity of labels,
    `segmented` and trace with fp32 storage of the same volume."""
    import subprocess, sys, os
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'full_size_check.py'), '--config5'], capture_output=True, text=True)
    assert out.returncode == 0 and 'CONFIG5 OK' in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def _gpu_slab_worker(rank, world, port, sweeps, outdir):
    import os, sys
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from arterynetwork_amd import slabs, phantoms
    dist.init_process_group('gloo', rank=rank, world_size=world)
    data, vmap = phantoms.tube_phantom(shape=(96, 64, 40), radius=3.0, seed=4, seed_planes=3, amp_y=10.0, amp_z=5.0,
                                       levels=32, brain_mask=True)
    s = slabs.make_slab_session(data.shape, rank, world, device=0, reduce='callback')   # both ranks share GPU 0
    s.set_volume(data); s.set_labels(vmap); s.init(2.25)
    s.run(sweeps, 10 ** 9, None)
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), labels=s.labels(), seg=s.segmented(), tr=s.trace())
    s.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_zslabs_processes_one_gpu(lib, tmp_path, world):
    """Z-slab driver with two / three ranks (sharing the one GPU of the box; host-callback reduction over gloo).
    The slab faces cut the 1024-voxel units of the dense pass (plane = 112 x 68 voxels), so its edge masking is
    exercised; the summed recount has to reproduce the incremental region sizes every sweep."""
    import torch.multiprocessing as mp
    from test_slabs_gloo import free_port
    from arterynetwork_amd import phantoms
    from arterynetwork_amd._capi import Session
    sweeps = 25
    data, vmap = phantoms.tube_phantom(shape=(96, 64, 40), radius=3.0, seed=4, seed_planes=3, amp_y=10.0, amp_z=5.0,
                                       levels=32, brain_mask=True)
    s = Session(data.shape, lib=lib)
    s.set_volume(data); s.set_labels(vmap); s.init(2.25)
    s.run(sweeps, 10 ** 9, None)
    ref = (s.labels(), s.segmented(), s.trace())
    s.close()
    mp.spawn(_gpu_slab_worker, args=(world, free_port(), sweeps, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        z = np.load(str(tmp_path / ('rank%d.npz' % r)))
        assert np.array_equal(z['labels'], ref[0]) and np.array_equal(z['seg'], ref[1])
        for f in ('nflip', 'nseg', 'n_in', 'n_out', 'ni', 'no'):
            assert np.array_equal(z['tr'][f], ref[2][f]), f


def test_rccl_single_rank_comm(lib):
    """The RCCL path (communicator + on-stream all-reduce) with a one-rank communicator."""
    from arterynetwork_amd import phantoms
    from arterynetwork_amd._capi import Session
    data, vmap = phantoms.scattered_seeds()
    outs = []
    for use_comm in (False, True):
        s = Session(data.shape, lib=lib)
        if use_comm:
            s.comm_init(1, 0, s.comm_unique_id())
        s.set_volume(data); s.set_labels(vmap); s.init(2.25)
        s.run(12, 10 ** 9, None)
        outs.append((s.labels(), s.segmented(), s.trace()))
        s.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert outs[0][2].tobytes() == outs[1][2].tobytes()


def test_refine_nifti_in_nifti_out(tmp_path):
    """File-level drop-in: stage-1 NIfTI files in, refined uint8 mask out, equal to the oracle's segmentedMap."""
    from arterynetwork_amd import nifti, phantoms
    from arterynetwork_amd.refine import refine, build_value_map
    from oracle import vrg_oracle as O
    data, vmap = phantoms.tube_phantom(shape=(64, 48, 32), radius=3.0, seed=6, seed_planes=4, amp_y=8.0, amp_z=4.0,
                                       levels=32, brain_mask=True, dtype=np.float32)
    aff = np.array([[0.4, 0, 0, -20.0], [0, 0.4, 0, -15.0], [0, 0, 0.5, 3.0], [0, 0, 0, 1.0]])
    nifti.saveVolume(data, aff, str(tmp_path / 'brainVolume.nii.gz'), astype=np.float32)
    nifti.saveVolume(vmap == 0, aff, str(tmp_path / 'vesselVolumeMask.nii.gz'))
    nifti.saveVolume(vmap != 4, aff, str(tmp_path / 'brainVolumeMask.nii.gz'))
    seg, segMap, vm = refine(str(tmp_path), iterMax=30, quiet=True)
    out, aff2 = nifti.loadVolume(str(tmp_path), 'vesselVolumeMaskRefined.nii.gz')
    assert out.dtype == np.uint8 and np.allclose(aff2, aff, atol=1e-6)
    vm_o = build_value_map(data, vmap == 0, vmap != 4).astype(np.int64)
    assert np.array_equal(vm_o, vmap)
    seg_o, segMap_o, _ = O.variationalRegionGrowing(data.astype(np.float64), vm_o, maxSegmentSize=data.size + 1,
                                                    iterMax=30, maxTime=-1.0, density_mode=1, quiet=True)
    assert np.array_equal(out, segMap_o) and np.array_equal(seg, seg_o) and np.array_equal(vm, vm_o)
    assert out.sum() > (vmap == 0).sum()


def test_refine_scaled_nifti_goes_through_as_float64(tmp_path):
    """A NIfTI with scl_slope / scl_inter (hand-written here: int16 storage, value = 0.1 * stored + 0.03) loads as float64
    with values float32 cannot hold; refine() hands it on as it is - the C-ABI keeps such a volume as float64
    (vrg_set_volume) - and the result equals the oracle's on the same float64 array."""
    import struct
    from arterynetwork_amd import nifti, phantoms
    from arterynetwork_amd.refine import refine, build_value_map
    from oracle import vrg_oracle as O
    data, vmap = phantoms.tube_phantom(shape=(40, 36, 24), radius=3.0, seed=9, seed_planes=3, amp_y=6.0, amp_z=3.0,
                                       levels=24, brain_mask=True, dtype=np.float32)
    stored = np.round(data * 24).astype(np.int16)                     # integer levels 0..24 (and noise below / above)
    hdr = bytearray(352)
    struct.pack_into('<i', hdr, 0, 348)
    struct.pack_into('<8h', hdr, 40, 3, *stored.shape, 1, 1, 1, 1)
    struct.pack_into('<2h', hdr, 70, 4, 16)                           # DT_INT16
    struct.pack_into('<8f', hdr, 76, 1.0, 0.5, 0.5, 0.5, 0, 0, 0, 0)
    struct.pack_into('<3f', hdr, 108, 352.0, 0.1, 0.03)               # vox_offset, scl_slope, scl_inter
    struct.pack_into('<2h', hdr, 252, 0, 1)                           # sform
    struct.pack_into('<12f', hdr, 280, 0.5, 0, 0, -9.0, 0, 0.5, 0, -8.0, 0, 0, 0.5, 2.0)
    hdr[344:348] = b'n+1\x00'
    (tmp_path / 'brainVolume.nii').write_bytes(bytes(hdr) + stored.astype('<i2').tobytes(order='F'))
    aff = np.array([[0.5, 0, 0, -9.0], [0, 0.5, 0, -8.0], [0, 0, 0.5, 2.0], [0, 0, 0, 1.0]])
    nifti.saveVolume(vmap == 0, aff, str(tmp_path / 'vesselVolumeMask.nii.gz'))
    nifti.saveVolume(vmap != 4, aff, str(tmp_path / 'brainVolumeMask.nii.gz'))
    vol, _ = nifti.loadVolume(str(tmp_path), 'brainVolume.nii')
    assert vol.dtype == np.float64 and not np.array_equal(vol.astype(np.float32).astype(np.float64), vol)
    seg, segMap, vm = refine(str(tmp_path), dataName='brainVolume.nii', H=20.0, iterMax=25, quiet=True)
    out, _ = nifti.loadVolume(str(tmp_path), 'vesselVolumeMaskRefined.nii.gz')
    vm_o = build_value_map(vol, vmap == 0, vmap != 4).astype(np.int64)
    seg_o, segMap_o, _ = O.variationalRegionGrowing(np.ascontiguousarray(vol), vm_o, H=20.0, maxSegmentSize=vol.size + 1, iterMax=25,
                                                    maxTime=-1.0, density_mode=1, quiet=True)
    assert np.array_equal(out, segMap_o) and np.array_equal(seg, seg_o) and np.array_equal(vm, vm_o)
    assert out.sum() > (vmap == 0).sum()


def test_layouts_dtypes_and_stop_order(lib):
    """C / Fortran order and integer dtypes through the device repack kernels; stop-test order :91 > :97 > :101."""
    from arterynetwork_amd._capi import Session
    from arterynetwork_amd import variationalRegionGrowing, phantoms
    rng = np.random.default_rng(3)
    shape = (37, 29, 23)
    I = rng.integers(0, 6, size=shape)
    vm = np.full(shape, 3, dtype=np.int64)
    vm[rng.random(shape) < 0.15] = 0
    vm[rng.random(shape) < 0.1] = 4
    if not (vm == 0).any():
        vm[0, 0, 0] = 0
    outs = []
    for data, lab in ((I.astype(np.float64), vm), (np.asfortranarray(I.astype(np.float32)), np.asfortranarray(vm.astype(np.uint8))),
                      (I.astype(np.int16), vm.astype(np.int32)), (I.astype(np.uint16), np.asfortranarray(vm))):
        s = Session(shape, lib=lib)
        s.set_volume(data); s.set_labels(lab); s.init(2.25)
        s.run(8, 10 ** 9, None)
        out_c = s.labels()
        assert np.array_equal(out_c, s.labels(out=np.empty(shape, dtype=np.int64, order='F')))
        outs.append((out_c, s.segmented()))
        s.close()
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1])
    # in-place write-back keeps the caller's dtype and memory order
    vmf = np.asfortranarray(vm.astype(np.uint8))
    seg, segMap, back = variationalRegionGrowing(I.astype(np.int16), vmf, iterMax=8, maxSegmentSize=10 ** 9, quiet=True)
    assert back is vmf and vmf.dtype == np.uint8 and np.array_equal(vmf, outs[0][0])
    # stop order
    data, vmap = phantoms.shell_with_holes()
    s = Session(data.shape, lib=lib)
    s.set_volume(data); s.set_labels(vmap); s.init(2.25)
    before = s.labels()
    r = s.run(200, 1, 0.0)
    assert r.stop_reason == 2 and r.sweeps == 0 and np.array_equal(s.labels(), before)
    assert s.run(200, 1, None).stop_reason == 3
    assert s.run(200, 10 ** 9, None).stop_reason == 1
    assert s.run(200, 1, 0.0).stop_reason == 1
    s.close()


def test_reference_dtypes_host_arrays_cross_chunks():
    """The reference's own calling convention - int64 valueMap, float64 dataArray, C order (:44-46, :288) - on a volume large enough
    for the host-side narrowing / widening to work in several chunks (40 M voxels > 32 M per chunk): same labels, `segmented`,
    segmentedMap (int64, built on the device) as the uint8 / fp32 Fortran-order call; valueMap is updated in place."""
    from arterynetwork_amd import variationalRegionGrowing, phantoms
    shape = (352, 352, 320)
    data, vmap = phantoms.bench_volume(shape, seed=2)
    a_d, a_v = np.asfortranarray(data), np.asfortranarray(vmap.astype(np.uint8))
    b_d, b_v = np.ascontiguousarray(data, dtype=np.float64), np.ascontiguousarray(vmap, dtype=np.int64)
    sa, ma, va = variationalRegionGrowing(a_d, a_v, iterMax=20, maxSegmentSize=10 ** 12, maxTime=None, quiet=True)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        sb, mb, vb = variationalRegionGrowing(b_d, b_v, iterMax=20, maxSegmentSize=10 ** 12, maxTime=None)
    assert vb is b_v and vb.dtype == np.int64 and mb.dtype == np.int64 and ma.dtype == np.int64
    assert np.array_equal(sa, sb) and np.array_equal(va, vb) and np.array_equal(ma, mb)
    assert np.array_equal(mb, (vb <= 1).astype(np.int64)) and int(mb.sum()) == len(sb)
    assert 'Total segmented voxels: {}/{}'.format(len(sb), int(np.count_nonzero(b_d))) in buf.getvalue()
    # a float64 volume that fp32 cannot hold goes through as float64 (the narrowing notices and hands the raw array over)
    c_d = b_d[:96, :96, :64].copy() + 1e-9
    c_v = b_v[:96, :96, :64].copy(); c_v[c_v <= 1] = 0; c_v[(c_v != 0) & (c_v != 4)] = 3
    if (c_v == 0).any():
        s1 = variationalRegionGrowing(c_d, c_v.copy(), iterMax=5, maxSegmentSize=10 ** 12, maxTime=None, quiet=True)[0]
        s2 = variationalRegionGrowing(np.asfortranarray(c_d), np.asfortranarray(c_v), iterMax=5, maxSegmentSize=10 ** 12, maxTime=None, quiet=True)[0]
        assert np.array_equal(s1, s2)


def test_16bit_storage_identical(lib, golden_loader):
    """storage16: the dense pass streams 2 B level indices (values from an LDS table) - same results and sums."""
    from arterynetwork_amd import phantoms
    from arterynetwork_amd._capi import Session, VrgError
    for name in ('adv_noise_q', 'adv_scattered_q', 'tube_q_small', 'kat_sphere'):
        g = golden_loader(name)
        data, vmap = g.inputs()
        iterMax = g.max_sweeps if g.max_sweeps >= 0 else 200
        res, k = parity.run_stepwise(lib, data, vmap, g.H, g.maxSegmentSize, iterMax, density_mode=1, check_hist=True,
                                     options={'storage16': 1})
        assert res is not None and k == g.ncalls - 1
    # up to 4096 levels the dense pass keeps the level values in LDS as doubles (kernel mode 3), beyond as floats (mode 1):
    # a 255-step and a 12-bit quantisation (the latter: > 4096 distinct values)
    for levels, mode in ((255, 3), (4095, 1)):
        data, vmap = phantoms.bench_volume((256, 192, 96), seed=4, levels=levels)
        outs = []
        for st in (0, 1):
            s = Session(data.shape, lib=lib)
            s.set_option('storage16', st)
            s.set_volume(data); s.set_labels(vmap.astype(np.uint8)); s.init(2.25)
            s.run(40, 10 ** 9, None)
            outs.append((s.labels(), s.segmented(), s.trace(), s.band(0), s.band(1), s.stats()['dense_kernel'], s.nlevels()))
            s.close()
        assert (outs[1][6] > 4096) == (mode == 1) and ',%d,' % mode in outs[1][5], (outs[1][5], outs[1][6])
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        assert outs[0][2].tobytes() == outs[1][2].tobytes()          # trace incl. the f64 intensity sums: bit-identical
        for w in (3, 4):
            for x, y in zip(outs[0][w], outs[1][w]):
                assert np.array_equal(x, y)
    # continuous-valued data has too many distinct values for 16-bit storage: loud error, no silent fallback
    d2, v2 = phantoms.config1()
    s = Session(d2.shape, lib=lib)
    s.set_option('storage16', 1)
    s.set_volume(d2); s.set_labels(v2)
    with pytest.raises(VrgError):
        s.init(2.25)
    s.close()


def test_skip_excluded_is_bit_identical(lib):
    """The dense pass does not fetch runs of excluded voxels (option skip_excluded, default 1): same labels and the same
    f64 intensity sums, bit for bit, as the pass that streams every voxel - fp32, 16-bit and float64 storage; brain
    mask (long runs), scattered excluded voxels (every group mixed), nothing excluded.  The byte count the roofline
    uses follows the documented definition."""
    from arterynetwork_amd import phantoms
    from arterynetwork_amd._capi import Session
    from test_hostmodel import dense_bytes_by_definition
    rng = np.random.default_rng(21)
    cases = []
    d, v = phantoms.bench_volume((256, 192, 96), seed=4)
    cases += [(d, v, 0, 32, 40), (d, v, 1, 64, 40)]
    shape = (83, 61, 47)
    u = rng.random(shape)
    vm = np.full(shape, 3, dtype=np.int64); vm[u < 0.03] = 0; vm[u > 0.55] = 4
    cases.append((rng.integers(0, 7, size=shape).astype(np.float64), vm, 0, 32, 12))
    cases.append((rng.integers(0, 7, size=shape).astype(np.float64), vm, 1, 64, 12))
    cases.append((np.round(rng.standard_normal(shape), 1) + 1e-9 * rng.standard_normal(shape), vm, 0, 16, 6))   # float64 storage
    vm2 = np.full(shape, 3, dtype=np.int64); vm2[u < 0.03] = 0
    cases.append((rng.integers(0, 7, size=shape).astype(np.float64), vm2, 0, 32, 12))
    for I, vmap, s16, line, sweeps in cases:
        outs = []
        for skip in (0, 1):
            s = Session(I.shape, lib=lib)
            s.set_option('storage16', s16); s.set_option('skip_excluded', skip)
            s.set_volume(I); s.set_labels(vmap.astype(np.uint8)); s.init(2.25)
            r = s.run(sweeps, 10 ** 9, None)
            outs.append((s.labels(), s.trace(), r.sweeps, s.stats()['dense_bytes']))
            s.close()
        assert outs[0][2] == outs[1][2] > 0
        assert np.array_equal(outs[0][0], outs[1][0])
        assert outs[0][1].tobytes() == outs[1][1].tobytes()            # trace incl. sum_in / sum_out: bit-identical
        nx, ny, nz = I.shape
        streamed = ((nx + 2 + 15) // 16 * 16) * (ny + 4) * nz * (128 // line) + ((nx + 2 + 15) // 16 * 16) * (ny + 4) * nz // 4
        assert outs[0][3] == streamed                                  # skip_excluded = 0: every voxel of the slab
        assert outs[1][3] == dense_bytes_by_definition(outs[1][0], line) <= streamed + 512


def test_integer_valued_volume_level_map(lib):
    """Integer intensities within a span of 65 536 (what scanners deliver): a voxel's level comes from the direct map
    VrgCtx::lev_map instead of a search through the level table - here with more levels (> 2048) than k_mark_relabel keeps
    in LDS, so the relabel kernel itself goes through the map.  Against the oracle step by step; and the same volume divided
    by its quantisation (fractional values: no map) with H scaled accordingly takes the same decisions."""
    from arterynetwork_amd import phantoms
    from arterynetwork_amd._capi import Session
    levels = 4095
    frac, vmap = phantoms.bench_volume((112, 96, 80), seed=11, levels=levels)
    ints = np.round(frac.astype(np.float64) * levels).astype(np.float32)
    assert np.array_equal(ints, np.floor(ints)) and len(np.unique(ints)) > 2048
    H = 2.25 / float(levels) ** 2
    res, k = parity.run_stepwise(lib, ints, vmap, H, None, 25, density_mode=1, check_hist=True)
    assert res is not None and k == 25
    outs = []
    for data, h in ((ints, H), (ints.astype(np.float64) / levels, 2.25)):
        s = Session(data.shape, lib=lib)
        s.set_volume(data); s.set_labels(vmap.astype(np.uint8)); s.init(h)
        r = s.run(25, 10 ** 9, None)
        tr = s.trace()
        outs.append((s.labels(), s.segmented(), r.sweeps, tr['nflip'].copy(), tr['nseg'].copy(), r.ties))
        s.close()
    assert outs[0][2] == outs[1][2] == 25 and outs[0][5] == outs[1][5] == 0
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][3], outs[1][3]) and np.array_equal(outs[0][4], outs[1][4])


def test_dense_pipe_is_bit_identical(lib):
    """The two-trips-deep dense kernel (option dense_pipe, default 1; fp32 storage with skip_excluded) makes the same
    additions in the same order as k_recount_bits: same trace, bit for bit - and vrg_get_stats names the kernel that ran."""
    from arterynetwork_amd import phantoms
    from arterynetwork_amd._capi import Session
    d, v = phantoms.bench_volume((256, 192, 96), seed=4)
    outs = []
    for pipe in (0, 1):
        s = Session(d.shape, lib=lib)
        s.set_option('dense_pipe', pipe)
        s.set_volume(d.astype(np.float32)); s.set_labels(v.astype(np.uint8)); s.init(2.25)
        r = s.run(40, 10 ** 9, None)
        st = s.stats()
        outs.append((s.labels(), s.trace(), r.sweeps, st['dense_bytes'], st['dense_kernel']))
        s.close()
    assert outs[0][2] == outs[1][2] > 0 and np.array_equal(outs[0][0], outs[1][0])
    assert outs[0][1].tobytes() == outs[1][1].tobytes() and outs[0][3] == outs[1][3]
    assert outs[0][4].startswith('k_recount_bits<3,') and outs[1][4].startswith('k_recount_pipe<3,')


def test_dense_pass_event_sampling(lib):
    """Option events = n: the dense pass of every n-th sweep of a batch is timed (vrg_result.sweep_kernel_ms over
    sweep_launches launches); 0 times nothing.  Results do not depend on it."""
    from arterynetwork_amd import phantoms
    from arterynetwork_amd._capi import Session
    d, v = phantoms.bench_volume((96, 80, 64), seed=6)
    traces = []
    for every, batch in ((0, 8), (1, 8), (4, 8), (3, 5)):
        s = Session(d.shape, lib=lib)
        s.set_option('events', every); s.set_option('batch', batch)
        s.set_volume(d); s.set_labels(v.astype(np.uint8)); s.init(2.25)
        r = s.run(24, 10 ** 9, None)
        assert r.sweeps == 24
        if every == 0:
            assert r.sweep_launches == 0 and r.sweep_kernel_ms == 0
        else:
            # batches of `batch` trips (the last one carries the trip that finds the stop flag): every n-th trip of each
            want = sum(len(range(0, min(batch, 24 - b0), every)) for b0 in range(0, 24, batch))
            assert abs(r.sweep_launches - want) <= 1 and r.sweep_kernel_ms > 0
            assert 0.001 < r.sweep_kernel_ms / r.sweep_launches < 5.0
        traces.append(s.trace().tobytes())
        s.close()
    assert all(t == traces[0] for t in traces)


def test_host_driven_and_one_workgroup_sweeps_identical(lib, golden_loader):
    """update() as the three batched kernels (k_order / k_mark_relabel / k_close) and as host-driven device-wide kernels
    (sweeps with more flips than "small_flips"; rocPRIM sorts) must give the same state; so must arrays that start tiny
    and grow on demand."""
    from arterynetwork_amd._capi import Session
    g = golden_loader('adv_scattered')
    data, vmap = g.inputs()
    for opts in ({'small_flips': 0}, {'small_flips': 25, 'capacity_floor': 32}, {'small_flips': 4096, 'capacity_floor': 32}):
        res, k = parity.run_stepwise(lib, data, vmap, g.H, g.maxSegmentSize, 200, density_mode=1, check_hist=True, options=opts)
        assert res is not None and k == g.ncalls - 1
    # whole runs against the reference's recorded results: config 1 (continuous-valued; fused sweeps switched off and at most 10
    # flips for the four-launch chain, so that its sweeps are host-driven) and a quantised tube that starts with 16-entry
    # arrays and never leaves the batched path
    for name, small, floor in (('config1_tube', 10, 1 << 16), ('tube_q_small', 4096, 16)):
        g2 = golden_loader(name)
        d2, v2 = g2.inputs()
        iterMax = g2.max_sweeps if g2.max_sweeps >= 0 else 200
        s = Session(g2.shape, lib=lib)
        s.set_option('small_flips', small); s.set_option('batch', 16); s.set_option('capacity_floor', floor)
        s.set_option('fused', 0 if small == 10 else 1)
        s.set_volume(d2); s.set_labels(v2); s.init(g2.H)
        s.run(iterMax, g2.maxSegmentSize, None)
        assert np.array_equal(s.labels(), g2.z['final_labels'])
        assert np.array_equal(parity.lex_of(s.segmented(), g2.shape), g2.z['final_segmented'])
        st = s.stats()
        assert (st['host_driven_trips'] > 0) if small == 10 else (st['host_driven_trips'] == 0 and st['pool_capacity'] > 16)
        s.close()


def test_many_flip_sweeps_vs_oracle(lib):
    """Several disjoint tubes growing at once (SURVEY.md 8(d), config 5's shape): 1 600 flips per sweep (16 tubes) and 4 800
    (48 tubes).  The four-launch chain orders more than 512 flips chip-wide (k_rank_wide / k_list_wide / k_prepass_wide /
    k_fix_wide) and relabels four flips at a time per workgroup; above "small_flips" the trips are host-driven (rocPRIM sorts).
    Every variant must reproduce the oracle's labels, band lists (order included) and trace, twelve sweeps in one call."""
    import torch
    from arterynetwork_amd import phantoms
    from arterynetwork_amd._capi import Session
    from oracle import vrg_oracle as O
    # (levels=1300 -> 2035 distinct values: the level table and the per-level counts of k_mark_relabel's workgroups at the upper end of what they keep in LDS)
    for shape, tubes, fmin, levels in (((160, 160, 64), 16, 1500, 255), ((256, 192, 96), 48, 4097, 255), ((160, 160, 64), 16, 1000, 1300)):
        I, vm = phantoms.bench_volume_torch(shape, torch.device('cpu'), tubes=tubes, levels=levels)
        d = np.asfortranarray(I.numpy().astype(np.float64)); v = np.asfortranarray(vm.numpy())
        o = O.Oracle(d, v, 2.25, 1); o.init()
        k = 0
        while o.step(12, 10 ** 9, -1.0) == 0:
            k += 1
        otr = o.trace()
        assert k == 12 and int(otr['nflip'][1:].min()) >= fmin
        for opts, host_driven in (({}, False), ({'small_flips': 4096}, tubes == 48), ({'batch': 5, 'fused': 0}, False), ({'small_flips': 600}, True)):
            s = Session(shape, lib=lib)
            for kk, vv in opts.items():
                s.set_option(kk, vv)
            s.set_volume(d.astype(np.float32)); s.set_labels(v); s.init(2.25)
            r = s.run(12, 10 ** 9, None)
            assert r.sweeps == k and r.ties == 0, (tubes, opts)
            parity.compare_state(s, o, shape, parity.density_rtol(d), 'many flips, %d tubes, %r' % (tubes, opts))
            tr = s.trace()
            for f in ('nflip', 'nseg', 'n_in', 'n_out', 'ni', 'no'):
                assert np.array_equal(tr[f], otr[f]), (f, tubes, opts)
            assert (s.stats()['host_driven_trips'] > 0) == host_driven, (tubes, opts, s.stats())
            s.close()
        o.close()


def test_many_flips_beside_excluded_voxels_vs_oracle(lib):
    """Sweeps of 1 600 flips in a volume whose brain mask is SMALLER than the vessels' lattice (phantoms.bench_volume_torch brain_scale): some
    tubes grow through excluded voxels - the 4 -> 3 inclusion of :166-168 / :177-179 in every sweep - others inside the mask.  The compact
    relabel kernel (a flip per half-wave, 1-ring only) must hand exactly the flips with an excluded voxel in their 5x5x5 cube to the general
    kernel: some, not all (Session.stats()['slow_flips']); labels, band lists and trace equal the oracle's - as they do with the general
    kernel alone (mark_compact = 0)."""
    import torch
    from arterynetwork_amd import phantoms
    from arterynetwork_amd._capi import Session
    from oracle import vrg_oracle as O
    shape = (160, 160, 64)
    I, vm = phantoms.bench_volume_torch(shape, torch.device('cpu'), tubes=16, brain_scale=0.3)
    d = np.asfortranarray(I.numpy().astype(np.float64)); v = np.asfortranarray(vm.numpy())
    assert 0.3 < float((v == 4).mean()) < 0.95
    o = O.Oracle(d, v, 2.25, 1); o.init()
    k = 0
    while o.step(12, 10 ** 9, -1.0) == 0:
        k += 1
    otr = o.trace()
    assert k == 12 and int(otr['nflip'][1:].min()) > 1000
    total = int(otr['nflip'][1:].sum())
    for opts in ({}, {'mark_compact': 0}, {'fused': 0, 'batch': 5}):
        s = Session(shape, lib=lib)
        for kk, vv in opts.items():
            s.set_option(kk, vv)
        s.set_volume(d.astype(np.float32)); s.set_labels(v); s.init(2.25)
        r = s.run(12, 10 ** 9, None)
        assert r.sweeps == k and r.ties == 0, opts
        parity.compare_state(s, o, shape, parity.density_rtol(d), 'flips beside excluded voxels, %r' % (opts,))
        tr = s.trace()
        for f in ('nflip', 'nseg', 'n_in', 'n_out', 'ni', 'no'):
            assert np.array_equal(tr[f], otr[f]), (f, opts)
        st = s.stats()
        assert st['host_driven_trips'] == 0
        if opts.get('mark_compact', 1):
            assert 0 < st['slow_flips'] < total, (st['slow_flips'], total)     # some flips took the general kernel, some the compact one
        else:
            assert st['slow_flips'] == 0
        s.close()
    o.close()


@pytest.mark.parametrize('kind', ['bins', 'float64', 'storage16'])
def test_many_flips_other_storages_vs_oracle(lib, kind):
    """The compact relabel kernel on the other ways a volume is kept: a level table of thousands of values (12-bit quantisation: exact densities through
    bins, a per-voxel level index instead of a search), float64 intensities that fp32 cannot hold, 16-bit level storage - 1 600 flips per sweep each,
    eight sweeps against the oracle (labels, band lists with their densities, trace)."""
    import torch
    from arterynetwork_amd import phantoms
    from arterynetwork_amd._capi import Session
    from oracle import vrg_oracle as O
    shape = (160, 160, 64)
    I, vm = phantoms.bench_volume_torch(shape, torch.device('cpu'), tubes=16, levels=4095 if kind == 'bins' else 255)
    d = np.asfortranarray(I.numpy().astype(np.float64)); v = np.asfortranarray(vm.numpy())
    if kind == 'float64':
        d = np.asfortranarray(d * (1.0 + 2.0 ** -30))          # values float32 cannot hold: the device keeps float64
    o = O.Oracle(d, v, 2.25, 1); o.init()
    k = 0
    while o.step(8, 10 ** 9, -1.0) == 0:
        k += 1
    otr = o.trace()
    assert k == 8 and int(otr['nflip'][1:].min()) > 1000
    s = Session(shape, lib=lib)
    if kind == 'storage16':
        s.set_option('storage16', 1)
    s.set_volume(d if kind == 'float64' else d.astype(np.float32)); s.set_labels(v); s.init(2.25)
    if kind == 'bins':
        assert s.stats()['density_bins'] > 0
    r = s.run(8, 10 ** 9, None)
    assert r.sweeps == k and r.ties == 0
    parity.compare_state(s, o, shape, parity.density_rtol(d), 'many flips, %s' % kind)
    tr = s.trace()
    for f in ('nflip', 'nseg', 'n_in', 'n_out', 'ni', 'no'):
        assert np.array_equal(tr[f], otr[f]), (f, kind)
    assert s.stats()['host_driven_trips'] == 0
    s.close(); o.close()


def test_refine_like_mask_to_convergence_vs_oracle(lib):
    """What the pipeline uses this stage for (the reference's README.md:69-71, :209: VRG smooths an existing vessel mask): the seed is a
    PERTURBED mask of all vessels at once (phantoms.bench_volume_torch seed_mode='noisy-mask': a random half of the mask's surface taken
    off + 2 % salt, inside the brain mask and out of it), run to CONVERGENCE (:91) - thousands of flips in the first sweeps, then a
    decaying count, so the run passes through every kind of trip.  Labels, both band lists (order and densities), `segmented` order and
    the whole trace must equal the oracle's; so must the same run with the trips forced host-driven / never fused."""
    import torch
    from arterynetwork_amd import phantoms
    shape = (256, 192, 96)
    I, vm = phantoms.bench_volume_torch(shape, torch.device('cpu'), tubes=16, seed_mode='noisy-mask')
    d = np.asfortranarray(I.numpy().astype(np.float64)); v = np.asfortranarray(vm.numpy())
    res, k = parity.run_batched(lib, d, v, iterMax=300)
    assert res is not None, 'the oracle met an exact tie at sweep %d: the workload does not pin parity' % k
    assert res.stop_reason == 1 and k >= 3, (res.stop_reason, k)
    for opts in ({'small_flips': 600}, {'fused': 0, 'batch': 3}):
        r2, k2 = parity.run_batched(lib, d, v, iterMax=300, options=opts)
        assert r2 is not None and k2 == k and r2.stop_reason == 1, (opts, k2, k)


def test_two_live_sessions_are_independent(lib, golden_loader):
    """Two handles in one process (own streams, events and state each): their sweeps interleaved call by call, both
    must reproduce the oracle."""
    from arterynetwork_amd._capi import Session
    from oracle import vrg_oracle as O
    cases = []
    for name in ('adv_shell', 'tube_q_small'):
        g = golden_loader(name)
        data, vmap = g.inputs()
        s = Session(g.shape, lib=lib)
        s.set_volume(data); s.set_labels(vmap); s.init(g.H)
        o = O.Oracle(data, vmap, g.H, 1); o.init()
        cases.append((g, s, o))
    for k in range(1, 12):
        for g, s, o in cases:
            rc = o.step(200, g.maxSegmentSize, -1.0)
            r = s.run(k, g.maxSegmentSize, None)
            if rc == 0:
                assert r.sweeps == 1
            parity.compare_state(s, o, g.shape, parity.density_rtol(data), '%s sweep %d' % (g.name, k))
    for g, s, o in cases:
        s.close(); o.close()


def test_handle_reuse(lib):
    """One handle used again and again: new labels on the same volume, a new volume (other level table), 16-bit storage
    switched on and off, a run continued with a larger iterMax - each time the result of a fresh handle."""
    from arterynetwork_amd._capi import Session
    from arterynetwork_amd import phantoms
    d1, v1 = phantoms.tube_phantom(shape=(40, 36, 24), radius=3.0, seed=5, seed_planes=3, amp_y=7.0, amp_z=4.0, levels=16, brain_mask=True)
    d2, v2 = phantoms.noise_volume((40, 36, 24), seed=9, p_seed=0.1, p_excl=0.2, levels=6)
    v1b = v1.copy(); v1b[20:23, 16:20, 10:14] = 0

    def fresh(data, vmap, n, st16=0):
        s = Session(data.shape, lib=lib)
        s.set_option('storage16', st16)
        s.set_volume(data); s.set_labels(vmap); s.init(2.25)
        s.run(n, 10 ** 9, None)
        out = (s.labels(), s.segmented(), s.trace(), s.band(0), s.band(1))
        s.close()
        return out

    def same(a, b):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        for f in ('nflip', 'nseg', 'n_in', 'n_out', 'ni', 'no'):
            assert np.array_equal(a[2][f], b[2][f]), f
        for w in (3, 4):
            assert np.array_equal(a[w][0], b[w][0])
            np.testing.assert_allclose(a[w][1], b[w][1], rtol=1e-12); np.testing.assert_allclose(a[w][2], b[w][2], rtol=1e-12)

    s = Session(d1.shape, lib=lib)
    steps = ((d1, v1, 12, 0), (None, v1b, 9, 0), (d2, v2, 6, 0), (d1, v1, 12, 1), (None, v1b, 9, 0))
    cur = None
    for data, vmap, n, st16 in steps:
        if data is not None:
            s.set_volume(data); cur = data
        s.set_option('storage16', st16)
        s.set_labels(vmap); s.init(2.25)
        s.run(n // 2, 10 ** 9, None)
        s.run(n, 10 ** 9, None)                          # continued with a larger iterMax
        same((s.labels(), s.segmented(), s.trace(), s.band(0), s.band(1)), fresh(cur, vmap, n, st16))
    s.close()


def test_float64_volume(lib):
    """Values fp32 cannot hold (the reference computes in float64 on whatever it is given): float64 storage, the dense
    pass streams 8 B per voxel; same results as the oracle."""
    rng = np.random.default_rng(5)
    done = 0
    for sd, shape in enumerate(((9, 11, 8), (33, 20, 17), (40, 37, 29))):
        I = rng.standard_normal(shape) * (1.0 + sd) + 1e-9 * rng.standard_normal(shape)
        assert np.any(I.astype(np.float32).astype(np.float64) != I)
        u = rng.random(shape)
        vm = np.full(shape, 3, dtype=np.int64); vm[u < 0.1] = 0; vm[u > 0.8] = 4
        res, k = parity.run_stepwise(lib, I, vm, 2.25, None, 6, density_mode=1, check_hist=True)
        done += res is not None
        res, k = parity.run_batched(lib, I, vm, 2.25, None, 6, density_mode=1, options={'batch': 4})
    assert done >= 2


def test_config3_full_size_properties():
    """880x880x640 (the headline size), same torch-generated volume as bench.py: size-independent properties after
    60 sweeps.  Runs tests/full_size_check.py in a fresh process that imports torch BEFORE the HIP library (torch
    bundles its own ROCm runtime; whichever is loaded first serves both - see INTEGRATION.md)."""
    import subprocess, sys, os
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'full_size_check.py')], capture_output=True, text=True)
    assert out.returncode == 0 and 'FULL SIZE OK' in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_config3_full_size_vs_oracle():
    """BASELINE configs[2] at FULL size against the oracle: 880x880x640, the bench volume, 100 sweeps - labels of all
    495 616 000 voxels, both band list orders and densities, `segmented` order, whole trace.  The oracle (all-cores build)
    needs ~25 GB of host memory and about a minute; skipped on a host with less than 64 GB available."""
    import subprocess, sys, os
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    if bench.host_memory_available_gb() < 64:
        pytest.skip('needs 64 GB of host memory')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'full_size_check.py'), '--oracle3'], capture_output=True, text=True)
    assert out.returncode == 0 and 'ORACLE3 OK' in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_config5_family_vs_oracle():
    """BASELINE configs[4]'s family against the ORACLE at a size it runs: a 1024x1024x128 volume (seed 5, as the 1024^3 test), 16-bit
    intensity storage, 60 sweeps - labels of all 134 217 728 voxels, both band list orders and densities, `segmented` order, trace.
    (The 1024^3 volume itself is compared with the fp32-storage run of the same library: test_config5_size_storage16.)"""
    import subprocess, sys, os
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    if bench.host_memory_available_gb() < 24:
        pytest.skip('needs 24 GB of host memory')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'full_size_check.py'), '--config5-oracle', '60'], capture_output=True, text=True)
    assert out.returncode == 0 and 'CONFIG5-ORACLE OK' in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_drop_in_messages_and_trace(golden_loader, capsys):
    """Every exit path of the reference prints its own message (:94-104, :118-120); the wrapper prints the same."""
    from arterynetwork_amd import variationalRegionGrowing
    from oracle import vrg_oracle as O
    # size stop: the reference's recorded stdout
    g = golden_loader('border_size_stop')
    data, vmap = g.inputs()
    vm = vmap.copy()
    seg, segMap, out = variationalRegionGrowing(data, vm, maxSegmentSize=g.maxSegmentSize)
    assert capsys.readouterr().out == str(g.z['stdout'])
    assert np.array_equal(out, g.z['final_labels']) and len(seg) == int(g.z['nseg'][-1])
    # iteration cap: same text as the oracle's restatement of :118-120, and the trace hook
    g = golden_loader('adv_shell')
    data, vmap = g.inputs()
    tr = []
    vm1, vm2 = vmap.copy(), vmap.copy()
    seg, _, _ = variationalRegionGrowing(data, vm1, maxSegmentSize=10 ** 9, iterMax=2, trace=tr)
    mine = capsys.readouterr().out
    O.variationalRegionGrowing(data, vm2, maxSegmentSize=10 ** 9, iterMax=2, maxTime=-1.0)
    theirs = capsys.readouterr().out
    assert mine == theirs and 'Max iteration reached! Finished at iteration 3' in mine
    assert np.array_equal(vm1, vm2)
    assert [t['nflip'] for t in tr] == [int(v) for v in g.z['nflip'][:3]] and tr[2]['nseg'] == int(g.z['nseg'][2])


@pytest.mark.parametrize('shape,sweeps', [('880x880x640', 40), ('512x512x170', 40)])
def test_config4_partition_8_ranks_one_gpu(shape, sweeps):
    """BASELINE configs[3]'s partition at FULL size on the one GPU of the box: 8 rank processes share GPU 0, each recounts
    its own Z-slab of the bench volume (880x880x640: 80 planes each, plane = 896 x 884 voxels against 1024-voxel units;
    512x512x170: uneven slabs of 22 / 21 planes), host-callback reduction over gloo.  Every rank's labels, `segmented`
    and integer trace equal the single-process run; the ranks' slab counts add up to the incremental region sizes of
    every sweep.  (tests/full_size_check.py --slabs, own processes.)"""
    import subprocess, sys, os
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'full_size_check.py'), '--slabs', '8', shape, str(sweeps)],
                         capture_output=True, text=True)
    assert out.returncode == 0 and 'SLABS OK' in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.parametrize('world,shape,sweeps,transport,leader_verifies,tubes', [
    (8, '880x880x640', 40, 'ipc', 0, 1), (3, '512x512x170', 30, 'ipc', 1, 1), (4, '512x512x170', 30, 'callback', 0, 1),
    (3, '256x192x96', 24, 'ipc', 0, 48)])            # (48 tubes: 4 800 flips per sweep - the four-launch chain's log, tens of thousands of records per sweep)
def test_replicas_n_ranks_one_gpu(world, shape, sweeps, transport, leader_verifies, tubes):
    """Leader / follower replication (BASELINE configs[3] re-partitioned) at FULL size on the one GPU of the
    box: `world` rank processes share GPU 0; the leader runs the band chain and logs every sweep, the followers map its log
    buffers (hipIpc - the transport ranks of one node use between GPUs) or receive them through gloo callbacks, apply them and
    count their share of the sweeps over the whole volume.  Every rank's labels, `segmented` order, whole trace - the intensity
    sums bit for bit - and result equal the single-process run.  (tests/full_size_check.py --replicas, own processes.)"""
    import subprocess, sys, os
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'full_size_check.py'), '--replicas', str(world), shape, str(sweeps), transport, str(leader_verifies)],
                         capture_output=True, text=True, env=dict(os.environ, VRG_CHECK_TUBES=str(tubes)))
    assert out.returncode == 0 and 'REPLICAS OK' in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.parametrize('transport,fault', [('ipc', 2), ('ipc', -3), ('callback', 2), ('callback', -2)])
def test_replicas_fail_together_one_gpu(transport, fault):
    """A replicated vrg_run is collective, and so are its failures: three rank processes on GPU 0; the leader fails on the host side when it opens
    its 2nd batch (fault > 0) or a follower cannot use a chunk (fault < 0) - over hipIpc (polling followers, a control block) and over callbacks
    (chunks).  Every rank's vrg_run must return an error; none may wait for the others for ever (the parent gives up after ten minutes)."""
    import subprocess, sys, os
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'full_size_check.py'), '--replicas', '3', '256x192x96', '60', transport, '0'],
                         capture_output=True, text=True, env=dict(os.environ, VRG_CHECK_FAULT=str(fault)), timeout=900)
    assert out.returncode == 0 and 'REPLICAS FAIL TOGETHER' in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.parametrize('transport', ['ipc', 'callback'])
def test_bench_n_rank_body_on_one_gpu(transport):
    """bench.py --gpus 3 as the driver starts it (torch.distributed.run, one process per rank), its three ranks sharing GPU 0
    (VRG_BENCH_SHARE_GPU: the box has one GPU; RCCL refuses two ranks on a device, so the log goes over hipIpc / callbacks): the
    N-rank body runs to its JSON line, valid, every sweep counted."""
    import subprocess, sys, os, json
    from conftest import ROOT
    env = dict(os.environ, VRG_BENCH_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '3', '--shape', '256x256x128', '--steps', '40', '--warmup', '8',
                          '--transport', transport, '--no-cpu-baseline', '--repl-batch', '8'], capture_output=True, text=True, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert out.returncode == 0 and lines, out.stdout[-2000:] + out.stderr[-3000:]
    d = json.loads(lines[-1])
    assert d['valid'] and d['n_gpus'] == 3 and d['steps'] == 40 and d['config']['transport'] == transport
    assert sum(r['sweeps_counted'] or 0 for r in d['config']['ranks']) + 16 >= 40      # (the leader counts a third of the 48 sweeps itself)
    assert d['roofline']['frac'] and d['roofline']['kernel_ms_avg'] > 0


def _replica_run(s, data, vmap, sweeps):
    s.set_volume(data); s.set_labels(vmap); s.init(2.25)
    r = s.run(sweeps, 10 ** 9, None)
    return s.labels(), s.segmented(), s.trace(), r


def test_replica_rccl_single_rank(lib):
    """The RCCL transport of the change log with a one-rank communicator: broadcasts and the closing all-reduce run through
    ncclBroadcast / ncclAllReduce on the transport stream; leader alone that counts everything, and one that counts nothing
    (its last sweep is still counted when the run ends).  Results identical to the plain handle."""
    from arterynetwork_amd import phantoms, replica
    from arterynetwork_amd._capi import Session
    data, vmap = phantoms.tube_phantom(shape=(96, 64, 40), radius=3.0, seed=4, seed_planes=3, amp_y=10.0, amp_z=5.0, levels=32, brain_mask=True)
    ref = _replica_run(Session(data.shape, lib=lib), data, vmap, 25)
    for lv in (True, False):
        s = replica.make_replica_session(data.shape, 0, 1, lib=lib, transport='rccl', leader_verifies=lv, options={'batch': 4})
        assert s.replica['transport'] == 'rccl'
        out = _replica_run(s, data, vmap, 25)
        st = s.repl_stats()
        s.close()
        assert np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1])
        for f in ('nflip', 'nseg', 'n_in', 'n_out', 'ni', 'no', 'ties'):
            assert np.array_equal(out[2][f], ref[2][f]), f
        if lv:
            assert out[2].tobytes() == ref[2].tobytes()
        else:                                                           # nobody summed the intensities but for the last sweep
            assert np.isnan(out[2]['sum_in'][1:-1]).all() and out[2]['sum_in'][-1] == ref[2]['sum_in'][-1]
        assert st['sweeps'] == 25 and st['transport'] == 'rccl'


def _rccl_replica_worker(rank, world, port, sweeps, outdir, leader_verifies):
    import os, sys
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from arterynetwork_amd import replica, phantoms
    torch.cuda.set_device(rank)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    data, vmap = phantoms.tube_phantom(shape=(96, 64, 40), radius=3.0, seed=4, seed_planes=3, amp_y=10.0, amp_z=5.0, levels=32, brain_mask=True)
    s = replica.make_replica_session(data.shape, rank, world, device=rank, transport='rccl', leader_verifies=leader_verifies, options={'batch': 4})
    out = _replica_run(s, data, vmap, sweeps)
    st = s.repl_stats()
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), labels=out[0], seg=out[1], tr=out[2], transport=np.str_(s.replica['transport']),
             stats=np.array([st['batches'], st['chunks'], st['sweeps'], st['verified']], np.int64))
    s.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,leader_verifies', [(2, True), (2, False), (4, True), (4, False), (8, True), (8, False)])
def test_replica_rccl_world_n(lib, tmp_path, world, leader_verifies):
    """RCCL with MORE THAN ONE rank: `world` processes, one GPU each, the change log streamed sweep by sweep over ncclBroadcast (xGMI
    between the GPUs), the leader counting a share (leader_verifies) or only leading - the 8-rank configuration.  Every rank's labels,
    `segmented` order and trace equal the single-process run (the sums bit for bit where somebody counted the sweep).
    Skipped where fewer than `world` GPUs are visible (the driver's multi-GPU node runs it)."""
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip('needs {} GPUs'.format(world))
    import torch.multiprocessing as mp
    from test_slabs_gloo import free_port
    from arterynetwork_amd import phantoms
    from arterynetwork_amd._capi import Session
    data, vmap = phantoms.tube_phantom(shape=(96, 64, 40), radius=3.0, seed=4, seed_planes=3, amp_y=10.0, amp_z=5.0, levels=32, brain_mask=True)
    ref = _replica_run(Session(data.shape, lib=lib), data, vmap, 25)
    mp.spawn(_rccl_replica_worker, args=(world, free_port(), 25, str(tmp_path), leader_verifies), nprocs=world, join=True)
    counted = 0
    for r in range(world):
        z = np.load(str(tmp_path / ('rank%d.npz' % r)))
        assert str(z['transport']) == 'rccl'
        assert np.array_equal(z['labels'], ref[0]) and np.array_equal(z['seg'], ref[1]) and z['tr'].tobytes() == ref[2].tobytes()
        assert z['stats'][2] == 25 and z['stats'][1] >= z['stats'][0]      # every sweep of the log seen; at least a chunk per batch
        counted += int(z['stats'][3])
    assert counted >= (25 if not leader_verifies else 25 - 25 // world - 1)  # (the followers' shares; a verifying leader counts its own)


def test_reports_ties(lib):
    """An integer volume with proportional class histograms (constant intensity): every sign test (:87) is an exact tie,
    which the reference decides by np.sum's rounding.  The library counts them (vrg_result.ties, trace field `ties`) and
    the drop-in function warns - 'bit-exact labels' holds unless ties > 0."""
    from arterynetwork_amd import variationalRegionGrowing
    from arterynetwork_amd._capi import Session
    from test_hostmodel import tie_volume
    I, vm = tie_volume()
    s = Session(I.shape, lib=lib)
    s.set_volume(I); s.set_labels(vm); s.init(2.25)
    r = s.run(3, 10 ** 9, None)
    tr = s.trace()
    band0 = int(tr['ni'][0] + tr['no'][0])
    assert r.ties >= band0 > 0 and int(tr['ties'][1]) == band0
    s.close()
    with pytest.warns(RuntimeWarning, match='exact ties'):
        variationalRegionGrowing(I, vm.copy(), iterMax=3, maxSegmentSize=10 ** 9, quiet=True)
    import warnings
    from arterynetwork_amd import phantoms
    d, v = phantoms.scattered_seeds()
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        variationalRegionGrowing(d, v.copy(), iterMax=5, maxSegmentSize=10 ** 9, quiet=True)     # no tie: no warning


def test_handed_back_trip_with_callback_reduce(lib):
    """A trip handed back because of its flip count (VBAIL_FLIPS) in the middle of a long batch, with a slab reduction
    callback and no event timing: the launches of the rest of the batch must drain before the stop word is cleared
    (else a leftover gate would take the next sweep's request).  Result = the oracle's."""
    from arterynetwork_amd._capi import Session
    from oracle import vrg_oracle as O
    I, vm, H, _, _ = random_case(200041, 10, 40)
    for small in (1, 3):
        o = O.Oracle(I, vm, H, 1); o.init()
        k = 0
        while o.step(25, 10 ** 9, -1.0) == 0:
            k += 1
        s = Session(I.shape, lib=lib)
        calls = []
        s.set_reduce_callback(lambda v: (calls.append(list(v)), v)[1])
        for kk, vv in (('small_flips', small), ('events', 0), ('batch', 64)):
            s.set_option(kk, vv)
        s.set_volume(I); s.set_labels(vm); s.init(H)
        r = s.run(25, 10 ** 9, None)
        assert r.sweeps == k and r.ties == 0
        assert s.stats()['bail_flips'] + s.stats()['bail_fuse'] >= 1
        parity.compare_state(s, o, I.shape, parity.density_rtol(I), 'handed-back trips, small_flips %d' % small)
        tr, otr = s.trace(), o.trace()
        for f in ('nflip', 'nseg', 'n_in', 'n_out', 'ni', 'no'):
            assert np.array_equal(tr[f], otr[f]), f
        assert len(calls) == k + 1 and [c[0] for c in calls] == [float(x) for x in tr['n_in']]
        s.close(); o.close()
