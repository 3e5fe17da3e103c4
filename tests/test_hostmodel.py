"""CPU checks of the HIP pipeline's logic through the sequential host model (test infrastructure).

tests/hostmodel/libvrg_hostmodel.so runs the same item functions as the HIP kernels
(arterynetwork_amd/csrc/vrg_items.h) one item at a time.  Here it is compared with the oracle after
every sweep: labels, band list orders, `segmented` order, counts and stop reason exactly; densities
to 1e-9.  The GPU-side twin of this file is tests/test_gpu_parity.py.
"""
import os
import subprocess

import numpy as np
import pytest

import parity
from conftest import golden_names, ROOT
from arterynetwork_amd._capi import VrgLib

HM_DIR = os.path.join(ROOT, 'tests', 'hostmodel')


@pytest.fixture(scope='module')
def hm():
    subprocess.check_call(['make', '-C', HM_DIR, '-s', 'libvrg_hostmodel.so'])
    return VrgLib(os.path.join(HM_DIR, 'libvrg_hostmodel.so'), 'vrgm_')


SMALL = [n for n in golden_names() if n != 'config1_tube']


@pytest.mark.parametrize('variant', [0, 1])
@pytest.mark.parametrize('name', SMALL)
def test_hostmodel_matches_oracle_on_goldens(hm, golden_loader, name, variant):
    g = golden_loader(name)
    data, vmap = g.inputs()
    iterMax = g.max_sweeps if g.max_sweeps >= 0 else 200
    res, k = parity.run_stepwise(hm, data, vmap, g.H, g.maxSegmentSize, iterMax, density_mode=1,
                                 check_hist=True, options={'sweep_variant': variant})
    assert res is not None
    assert k == g.ncalls - 1
    assert res.nseg == int(g.z['nseg'][-1])


def random_case(sd, lo=1, hi=13):
    rng = np.random.default_rng(sd)
    shape = tuple(int(x) for x in rng.integers(lo, hi, size=3))
    p_seed = rng.choice([0.02, 0.1, 0.3, 0.6])
    p_excl = rng.choice([0.0, 0.2, 0.5])
    kind = rng.integers(0, 3)
    if kind == 0:
        I = rng.standard_normal(shape).astype(np.float32).astype(np.float64)
    elif kind == 1:
        I = rng.integers(0, 3, size=shape).astype(np.float64)
    else:
        I = (np.round(rng.standard_normal(shape) * 3) / 3).astype(np.float32).astype(np.float64)
    u = rng.random(shape)
    vm = np.full(shape, 3, dtype=np.int64)
    vm[u < p_seed] = 0
    vm[u > 1 - p_excl] = 4
    if not (vm == 0).any():
        vm.reshape(-1)[0] = 0
    H = float(rng.choice([0.3, 1.0, 2.25, 6.0]))
    return I, vm, H, int(rng.integers(0, 2)), int(rng.integers(0, 2))


def test_hostmodel_random_volumes(hm):
    """300 random tiny volumes (degenerate axes, excluded voxels, ties in integer data)."""
    sweeps = done = 0
    for sd in range(300):
        I, vm, H, variant, dmode = random_case(sd)
        res, k = parity.run_stepwise(hm, I, vm, H, None, 40, density_mode=dmode, check_hist=True,
                                     options={'sweep_variant': variant})
        sweeps += k
        done += res is not None
    assert sweeps > 1000
    assert done >= 280, 'too many cases cut short by an exact tie: {} of 300 completed'.format(done)


def test_hostmodel_fused_sweeps(hm, golden_loader):
    """The fused sweep (k_sweep's phase functions, run workgroup by workgroup and thread by thread by the host model) with the
    limit the device uses (128 flips per sweep; the model's default of 5 keeps the hand-back to the four-launch chain busy
    instead): every golden stepwise and in one call, random volumes, and a count of the trips that really ran fused."""
    from arterynetwork_amd._capi import Session
    fused = 0
    for name in SMALL:
        g = golden_loader(name)
        data, vmap = g.inputs()
        iterMax = g.max_sweeps if g.max_sweeps >= 0 else 200
        res, k = parity.run_stepwise(hm, data, vmap, g.H, g.maxSegmentSize, iterMax, density_mode=1, check_hist=True, options={'fuse_max': 128})
        assert res is not None and k == g.ncalls - 1, name
        res, k = parity.run_batched(hm, data, vmap, g.H, g.maxSegmentSize, iterMax, density_mode=1, options={'fuse_max': 128, 'small_flips': 64, 'batch': 5})
        assert res is not None, name
        s = Session(g.shape, lib=hm)
        s.set_option('fuse_max', 128)
        s.set_volume(data); s.set_labels(vmap); s.init(g.H)
        s.run(iterMax, g.maxSegmentSize, None)
        fused += s.stats()['fused_trips']
        off = Session(g.shape, lib=hm)
        off.set_option('fused', 0)
        off.set_volume(data); off.set_labels(vmap); off.init(g.H)
        off.run(iterMax, g.maxSegmentSize, None)
        assert off.stats()['fused_trips'] == 0
        assert np.array_equal(s.labels(), off.labels()) and np.array_equal(s.segmented(), off.segmented()), name
        s.close(); off.close()
    assert fused > 200, fused
    sweeps = 0
    for sd in range(700, 800):
        I, vm, H, variant, dmode = random_case(sd)
        res, k = parity.run_stepwise(hm, I, vm, H, None, 40, density_mode=dmode, check_hist=True, options={'sweep_variant': variant, 'fuse_max': 128})
        sweeps += k
    assert sweeps > 300


def _verify_every_case(lib, golden_loader):
    """Option verify_every (the dense pass on every n-th sweep only / never): labels, lists, densities and the integer trace are
    the same for every value; the trace's intensity sums are there exactly for the sweeps that were counted (NaN elsewhere) -
    and for the run's LAST sweep whatever the value (it is counted when the run ends); a handle keeps working when the value
    changes between runs."""
    from arterynetwork_amd._capi import Session
    g = golden_loader('tube_q_small')
    data, vmap = g.inputs()
    ref = None
    for every in (1, 3, 0):
        for fused in (1, 0):
            s = Session(g.shape, lib=lib)
            s.set_option('fused', fused); s.set_option('verify_every', every); s.set_option('batch', 5)
            s.set_volume(data); s.set_labels(vmap); s.init(g.H)
            r1 = s.run(17, g.maxSegmentSize, None)
            s.set_option('verify_every', 1 if every == 0 else every)      # (any time: the next run counts again)
            r2 = s.run(25, g.maxSegmentSize, None)
            assert r1.sweeps == 17 and r2.sweeps == 8
            tr = s.trace()
            out = (s.labels(), s.segmented(), s.band(0), s.band(1), [tr[f].copy() for f in ('nflip', 'nseg', 'n_in', 'n_out', 'ni', 'no')])
            counted = ~np.isnan(tr['sum_in'])
            want = np.zeros(len(tr), bool)
            want[0] = True
            for k in range(1, 26):
                e = every if k <= 17 else (1 if every == 0 else every)
                want[k] = (e == 1) or (e > 1 and k % e == 0) or k in (17, 25)
            assert np.array_equal(counted, want), (every, fused, counted.nonzero()[0], want.nonzero()[0])
            if ref is None:
                ref = (out, tr['sum_in'].copy(), tr['sum_out'].copy())
            else:
                assert np.array_equal(out[0], ref[0][0]) and np.array_equal(out[1], ref[0][1]), (every, fused)
                for a, b in zip(out[2] + out[3], ref[0][2] + ref[0][3]):
                    assert np.array_equal(a, b) if a.dtype.kind == 'i' else np.allclose(a, b, rtol=1e-9, atol=1e-12), (every, fused)
                for a, b in zip(out[4], ref[0][4]):
                    assert np.array_equal(a, b)
                np.testing.assert_allclose(tr['sum_in'][counted], ref[1][counted], rtol=1e-12)
                np.testing.assert_allclose(tr['sum_out'][counted], ref[2][counted], rtol=1e-12)
            s.close()


def test_hostmodel_verify_every(hm, golden_loader):
    _verify_every_case(hm, golden_loader)


def _binned_case(lib, golden_loader, tag):
    import json
    from arterynetwork_amd._capi import Session
    rep = {}
    for name in ('adv_noise0', 'adv_noise2', 'adv_scattered', 'adv_shell', 'adv_noise_q', 'tube_q_small'):
        g = golden_loader(name)
        data, vmap = g.inputs()
        iterMax = g.max_sweeps if g.max_sweeps >= 0 else 200
        res, k = parity.run_stepwise(lib, data, vmap, g.H, g.maxSegmentSize, iterMax, density_mode=1, check_hist=True, options={'bin_above': 0})
        assert res is not None and k == g.ncalls - 1, name
        res, k = parity.run_batched(lib, data, vmap, g.H, g.maxSegmentSize, iterMax, density_mode=1, options={'bin_above': 0, 'batch': 6})
        assert res is not None, name
        dev = 0.0
        outs = []
        for above in (0, 1 << 30):                            # bins / sums over the levels, same library
            s = Session(g.shape, lib=lib)
            s.set_option('bin_above', above)
            s.set_volume(data); s.set_labels(vmap); s.init(g.H)
            assert (s.stats()['density_bins'] > 0) == (above == 0)
            s.run(iterMax, g.maxSegmentSize, None)
            outs.append((s.labels(), s.band(0), s.band(1)))
            s.close()
        assert np.array_equal(outs[0][0], outs[1][0]), name
        for w in (1, 2):
            assert np.array_equal(outs[0][w][0], outs[1][w][0]), name
            for q in (1, 2):
                a, b = outs[0][w][q], outs[1][w][q]
                scale = max(1e-300, float(np.max(np.abs(b)))) if len(b) else 1.0
                if len(b):
                    dev = max(dev, float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3 * scale))))
        assert dev <= 1e-7, (name, dev)                       # (bound 2e-8 on the exact sums; entries that went through corrections since carry it on)
        rep[name] = {'levels': int(len(np.unique(data))), 'max_relative_deviation_bins_vs_level_sums': dev}
    d = os.path.join(ROOT, 'gpurun_out')
    if tag == 'gpu' and os.path.isdir(d):
        json.dump(rep, open(os.path.join(d, 'binned_deviation.json'), 'w'), indent=1)
    print(json.dumps(rep))


def test_hostmodel_binned_exact_densities(hm, golden_loader):
    _binned_case(hm, golden_loader, 'cpu')


def _fused_large_table_case(lib, extra=None):
    """Fused sweeps on LARGE level tables (continuous-valued and 12-bit tubes: 46 059 / 3 109 distinct values - bins for the
    exact densities, every voxel's level index kept, the touched levels listed by their first toucher and sorted by the
    workgroup that closes the sweep): stepwise and in one call against the oracle, and the trips really ran fused."""
    from arterynetwork_amd import phantoms
    from arterynetwork_amd._capi import Session
    for lv in (None, 4095):
        data, vmap = phantoms.tube_phantom(shape=(48, 40, 24), radius=2.5, seed=5, seed_planes=3, amp_y=8.0, amp_z=4.0, levels=lv, brain_mask=True)
        opts = dict(extra or {})
        res, k = parity.run_stepwise(lib, data, vmap, 2.25, None, 30, density_mode=1, check_hist=True, options=opts)
        assert res is not None and k == 30
        res, k = parity.run_batched(lib, data, vmap, 2.25, None, 30, density_mode=1, options=dict(opts, small_flips=4096, batch=7))
        assert res is not None
        s = Session(data.shape, lib=lib)
        for kk, v in dict(opts, small_flips=4096).items():
            s.set_option(kk, v)
        s.set_volume(data); s.set_labels(vmap); s.init(2.25)
        s.run(30, 10 ** 9, None)
        st = s.stats()
        assert st['fused_trips'] >= 30 and st['bail_fuse'] == 0 and st['density_bins'] > 0, st
        s.close()


def test_hostmodel_fused_sweeps_on_large_level_tables(hm):
    _fused_large_table_case(hm, {'fuse_max': 128})


def test_hostmodel_arrays_grow_on_demand(hm):
    """Pool and marked-voxel arrays start tiny (capacity_floor 16) and grow when a trip is handed back (VBAIL_MARKS /
    VBAIL_POOL) or when init counts more band voxels than fit; small_flips 0/3/10^6 runs every sweep host-driven /
    mixed / as one "workgroup".  Results stay those of the oracle."""
    from arterynetwork_amd._capi import Session
    grown = 0
    for sd, small in ((3, 0), (17, 3), (41, 10 ** 6), (77, 3), (123, 0)):
        I, vm, H, variant, dmode = random_case(sd, 5, 13)
        res, k = parity.run_stepwise(hm, I, vm, H, None, 25, density_mode=1, check_hist=True,
                                     options={'capacity_floor': 16, 'small_flips': small})
        assert res is not None
    data, vmap = __import__('arterynetwork_amd.phantoms', fromlist=['x']).scattered_seeds()
    s = Session(data.shape, lib=hm)
    s.set_option('capacity_floor', 16); s.set_option('small_flips', 5)
    s.set_volume(data); s.set_labels(vmap); s.init(2.25)
    s.run(15, 10 ** 9, None)
    st = s.stats()
    # (a trip that is handed back settles everything its flip count implies in one round trip: whichever reason came first - more flips than a fused
    # or a four-launch trip takes - also switched the host-driven trips on and let the arrays grow)
    assert st['grow_marks'] > 0 and st['pool_capacity'] > 16 and st['bail_flips'] + st['bail_fuse'] > 0 and st['host_driven_trips'] > 0
    s.close()


def test_hostmodel_handle_reuse(hm):
    """One handle used again and again: new labels on the same volume, a new volume (other level table), 16-bit storage
    switched on and off, a run continued with a larger iterMax - each time the result of a fresh handle."""
    from arterynetwork_amd._capi import Session
    from arterynetwork_amd import phantoms
    d1, v1 = phantoms.tube_phantom(shape=(40, 36, 24), radius=3.0, seed=5, seed_planes=3, amp_y=7.0, amp_z=4.0, levels=16, brain_mask=True)
    d2, v2 = phantoms.noise_volume((40, 36, 24), seed=9, p_seed=0.1, p_excl=0.2, levels=6)
    v1b = v1.copy(); v1b[20:23, 16:20, 10:14] = 0

    def fresh(data, vmap, n, st16=0):
        s = Session(data.shape, lib=hm)
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

    s = Session(d1.shape, lib=hm)
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


def test_hostmodel_rejects_bad_inputs(hm):
    from arterynetwork_amd._capi import Session, VrgError
    s = Session((4, 5, 6), lib=hm)
    with pytest.raises(VrgError):            # init before inputs
        s.init(2.25)
    s.set_volume(np.zeros((4, 5, 6)))
    with pytest.raises(VrgError):            # label outside {0,3,4}
        s.set_labels(np.full((4, 5, 6), 1))
    s.set_labels(np.full((4, 5, 6), 3))
    with pytest.raises(VrgError) as e:       # no seed: the reference raises at :48
        s.init(2.25)
    assert e.value.code == -5
    s.set_volume(np.full((4, 5, 6), 0.1))    # 0.1 is not an fp32 value: kept as float64 (no VRG_E_INEXACT any more)
    s.close()


def test_hostmodel_float64_volume(hm):
    """Volumes with values fp32 cannot hold (the reference computes in float64 on whatever it is given, and its
    pipeline writes float64 NIfTI): stored as float64, same results as the oracle."""
    rng = np.random.default_rng(11)
    done = 0
    for sd in range(12):
        shape = tuple(int(x) for x in rng.integers(3, 12, size=3))
        I = rng.standard_normal(shape) * (1.0 + sd) + 1e-9 * rng.standard_normal(shape)
        assert np.any(I.astype(np.float32).astype(np.float64) != I)
        u = rng.random(shape)
        vm = np.full(shape, 3, dtype=np.int64); vm[u < 0.15] = 0; vm[u > 0.8] = 4
        if not (vm == 0).any():
            vm.reshape(-1)[0] = 0
        res, k = parity.run_stepwise(hm, I, vm, 2.25, None, 25, density_mode=sd % 2, check_hist=True)
        done += res is not None
    assert done >= 10


def test_hostmodel_layouts_and_dtypes(hm):
    """C order / Fortran order / integer dtypes give the same result (strides are part of the ABI)."""
    from arterynetwork_amd._capi import Session
    rng = np.random.default_rng(3)
    shape = (7, 9, 5)
    I = rng.integers(0, 4, size=shape)
    vm = np.full(shape, 3, dtype=np.int64)
    vm[rng.random(shape) < 0.2] = 0
    outs = []
    for data, lab in ((I.astype(np.float64), vm), (np.asfortranarray(I.astype(np.float32)), np.asfortranarray(vm.astype(np.uint8))),
                      (I.astype(np.int16), vm.astype(np.int32))):
        s = Session(shape, lib=hm)
        s.set_volume(data)
        s.set_labels(lab)
        s.init(2.25)
        s.run(10, 10 ** 9, None)
        out_c = s.labels()
        out_f = s.labels(out=np.empty(shape, dtype=np.int64, order='F'))
        assert np.array_equal(out_c, out_f)
        outs.append((out_c, s.segmented()))
        s.close()
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1])


def test_hostmodel_time_cap_order(hm):
    """:91 no flips beats :97 time beats :101 size, all before update() is applied."""
    from arterynetwork_amd._capi import Session
    from arterynetwork_amd import phantoms
    data, vmap = phantoms.shell_with_holes()
    s = Session(data.shape, lib=hm)
    s.set_volume(data); s.set_labels(vmap); s.init(2.25)
    before = s.labels()
    r = s.run(200, 1, 0.0)                       # time cap already reached and size cap reached: time wins
    assert r.stop_reason == 2 and r.iter_num == 1 and r.sweeps == 0
    assert np.array_equal(s.labels(), before)
    r = s.run(200, 1, None)                      # no time cap: size stop
    assert r.stop_reason == 3 and r.sweeps == 0
    r = s.run(200, 10 ** 9, None)                # runs to convergence; afterwards "no flips" beats an expired timer
    assert r.stop_reason == 1
    r = s.run(200, 1, 0.0)
    assert r.stop_reason == 1 and r.sweeps == 0
    s.close()


@pytest.mark.parametrize('name', ['adv_noise_q', 'adv_scattered_q', 'tube_q_small', 'kat_sphere'])
def test_hostmodel_16bit_storage(hm, golden_loader, name):
    """storage16 (level index per voxel instead of the fp32 intensity) gives the same result, bit for bit."""
    g = golden_loader(name)
    data, vmap = g.inputs()
    iterMax = g.max_sweeps if g.max_sweeps >= 0 else 200
    res, k = parity.run_stepwise(hm, data, vmap, g.H, g.maxSegmentSize, iterMax, density_mode=1, check_hist=True,
                                 options={'storage16': 1})
    assert res is not None and k == g.ncalls - 1


def dense_bytes_by_definition(labels, line_voxels=32):
    """The bytes one dense pass has to fetch for a label volume [x][y][z], from the documented layout (DESIGN.md
    section 3/4): padded x-fastest volume; 256 B of class words (+ 4 B of list entry for a whole unit of the slab) per
    1024-voxel unit that holds an included (label != 4) voxel (the units the slab's faces cut: always) + 128 B for every
    aligned run of `line_voxels` voxels (one cache line of intensities) that holds an included voxel."""
    nx, ny, nz = labels.shape
    PX, PY = (nx + 2 + 15) // 16 * 16, ny + 4
    pad = np.zeros(((nz + 4) * PY * PX + 2048,), dtype=bool)
    vol = pad[:(nz + 4) * PY * PX].reshape(nz + 4, PY, PX)
    vol[2:nz + 2, 2:ny + 2, :nx] = np.transpose(labels != 4, (2, 1, 0))
    lo, hi = 2 * PY * PX, (nz + 2) * PY * PX
    f_lo, f_hi = (lo + 1023) >> 10, hi >> 10
    f_hi = max(f_hi, f_lo)
    listed = pad[:(len(pad) // 1024) * 1024].reshape(-1, 1024).any(axis=1)
    nbytes = 0
    for u in range(lo >> 10, ((hi - 1) >> 10) + 1):
        whole = f_lo <= u < f_hi
        if whole and not listed[u]:
            continue
        nbytes += 260 if whole else 256
    first = (lo // line_voxels) * line_voxels
    last = -(-hi // line_voxels) * line_voxels
    lines = pad[first:last].reshape(-1, line_voxels).any(axis=1).sum()
    return nbytes + 128 * int(lines)


def test_hostmodel_dense_bytes_counter(hm):
    """vrg_get_stats out[8] (what bench.py's roofline divides by the kernel time) counts class words + intensity lines
    with an included voxel: checked against the definition on volumes with a brain mask, scattered excluded voxels,
    none, and for the three storage widths (16-bit: 64 voxels per line, float64: 16)."""
    from arterynetwork_amd import phantoms
    from arterynetwork_amd._capi import Session
    cases = []
    d, v = phantoms.bench_volume((70, 45, 33), seed=4)
    cases.append((d, v, 0, 32)); cases.append((d, v, 1, 64))
    rng = np.random.default_rng(11)
    shape = (37, 22, 19)
    u = rng.random(shape)
    vm = np.full(shape, 3, dtype=np.int64); vm[u < 0.05] = 0; vm[u > 0.6] = 4
    cases.append((rng.integers(0, 5, size=shape).astype(np.float64), vm, 0, 32))
    cases.append((rng.standard_normal(shape) + 1e-9 * rng.standard_normal(shape), vm, 0, 16))      # float64 storage
    vm2 = np.full(shape, 3, dtype=np.int64); vm2[u < 0.05] = 0
    cases.append((rng.integers(0, 5, size=shape).astype(np.float64), vm2, 0, 32))                   # nothing excluded
    for I, vmap, s16, line in cases:
        s = Session(I.shape, lib=hm)
        s.set_option('storage16', s16)
        s.set_volume(I); s.set_labels(vmap.astype(np.uint8)); s.init(2.25)
        s.run(5, 10 ** 9, None)
        got = s.stats()['dense_bytes']
        lab = s.labels()
        s.close()
        assert got == dense_bytes_by_definition(lab, line), (I.shape, s16, line)


def tie_volume():
    """An integer volume whose class histograms are proportional (here: constant intensity): every sign test (:87) is an
    exact mathematical tie, inner/innerSize == outer/outerSize == A."""
    I = np.full((12, 11, 10), 3, dtype=np.int64)
    vm = np.full(I.shape, 3, dtype=np.int64)
    vm[4:7, 4:7, 3:6] = 0
    return I, vm


def test_hostmodel_reports_ties(hm):
    """Exact ties are counted where they are decided (vrg_result.ties, trace field `ties`), so a caller knows when
    'bit-exact labels' does not apply; a volume without ties reports none."""
    from arterynetwork_amd._capi import Session
    I, vm = tie_volume()
    s = Session(I.shape, lib=hm)
    s.set_volume(I); s.set_labels(vm); s.init(2.25)
    r = s.run(3, 10 ** 9, None)
    tr = s.trace()
    band0 = int(tr['ni'][0] + tr['no'][0])
    assert r.ties >= band0 > 0 and int(tr['ties'][1]) == band0       # every entry of the initial band was a tie
    assert int(tr['ties'].sum()) <= r.ties
    s.close()
    g_I, g_vm, H, _, _ = random_case(7)
    s = Session(g_I.shape, lib=hm)
    s.set_volume(g_I); s.set_labels(g_vm); s.init(H)
    r = s.run(5, 10 ** 9, None)
    assert r.ties == 0 and int(s.trace()['ties'].sum()) == 0
    s.close()


ASAN_SCRIPT = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, 'tests'))
import numpy as np
import parity
from conftest import Golden
from test_hostmodel import random_case
from arterynetwork_amd._capi import VrgLib
lib = VrgLib(os.path.join({root!r}, 'tests', 'hostmodel', 'libvrg_hostmodel_asan.so'), 'vrgm_')
for name in ('adv_noise_q', 'tube_q_small', 'border_size_stop'):
    g = Golden(name)
    data, vmap = g.inputs()
    res, k = parity.run_stepwise(lib, data, vmap, g.H, g.maxSegmentSize, g.max_sweeps if g.max_sweeps >= 0 else 200, density_mode=1, check_hist=True,
                                 options={{'capacity_floor': 16}})
    assert res is not None
for sd in range(40):
    I, vm, H, variant, dmode = random_case(sd)
    parity.run_stepwise(lib, I, vm, H, None, 12, density_mode=dmode, options={{'sweep_variant': variant, 'capacity_floor': 16, 'storage16': sd % 2 if len(np.unique(I)) < 1000 else 0}})
print('ASAN RUN OK')
'''


def test_hostmodel_under_address_and_ub_sanitizers():
    """The engine (vrg_engine.cpp: handle management, arrays that grow on demand, hand-back protocol) and the item
    functions under -fsanitize=address,undefined, through the sequential test model that shares them with the product:
    goldens and random volumes with 16-entry initial arrays.  Any report fails the test."""
    import sys
    subprocess.check_call(['make', '-C', HM_DIR, '-s', 'libvrg_hostmodel_asan.so'])
    libasan = subprocess.check_output(['gcc', '-print-file-name=libasan.so'], text=True).strip()
    libubsan = subprocess.check_output(['gcc', '-print-file-name=libubsan.so'], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=libasan + ' ' + libubsan, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:exitcode=23',
               UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    out = subprocess.run([sys.executable, '-c', ASAN_SCRIPT.format(root=ROOT)], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0 and 'ASAN RUN OK' in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
    assert 'AddressSanitizer' not in out.stderr and 'runtime error' not in out.stderr, out.stderr[-3000:]
