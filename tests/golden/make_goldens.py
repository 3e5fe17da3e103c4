#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (read-only, /root/reference) here.

Runs only in the build container (the reference never ships): it imports
/root/reference/Code/variationalRegionGrowing.py with its two unused imports
(nibabel, nrrd; variationalRegionGrowing.py:3-4) stubbed, neutralises the 120 s
wall-clock cap (:38,:97 look ``timeit`` up as a module global), lifts maxSegmentSize,
and wraps ``update`` (:124) to record the state after every call and to stop after a
chosen number of incremental sweeps (the reference hard-codes iterMax=200 at :56).

Outputs (data only - inputs by recipe or value, expected outputs by value):
    tests/golden/<case>.npz
Usage:  python tests/golden/make_goldens.py [case ...]
"""
from __future__ import annotations

import hashlib
import io
import contextlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from arterynetwork_amd import phantoms as P  # noqa: E402

REF = '/root/reference/Code'


def load_reference():
    sys.modules.setdefault('nibabel', types.ModuleType('nibabel'))
    sys.modules.setdefault('nrrd', types.ModuleType('nrrd'))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import variationalRegionGrowing as V
    V.timeit = types.SimpleNamespace(default_timer=lambda: 0.0)   # no 120 s cap
    return V


PROB_ITERS_FULL = (0, 1, 2, 3, 5, 8, 13, 21, 34)


class StopSweeps(Exception):
    pass


def lex(coords, shape):
    c = np.asarray(coords, dtype=np.int64).reshape(-1, 3)
    return (c[:, 0] * shape[1] + c[:, 1]) * shape[2] + c[:, 2]


def run_reference(V, data, vmap, H=2.25, maxSegmentSize=None, max_sweeps=None, full=True,
                  prob_iters=()):
    """Run the reference; return dict of recorded outputs.

    full=True stores labels / lists / probs after every update call; otherwise only the
    per-call trace plus lists+probs at the calls listed in prob_iters (0 = init)."""
    shape = data.shape
    vmap = vmap.copy()
    if maxSegmentSize is None:
        maxSegmentSize = data.size + 1
    rec = dict(nflip=[], nseg=[], n_in=[], n_out=[], ni=[], no=[],
               labels=[], inner=[], outer=[], ip=[], op=[], snap_iters=[])
    orig_update = V.update
    state = {}

    def wrapped(dataArray, segmented, segmentedMap, valueMap, H, flipedPoints=None, *a, **k):
        ncall = len(rec['nflip'])          # 0 = init, t = t-th incremental sweep
        if max_sweeps is not None and ncall > max_sweeps:
            raise StopSweeps()
        out = orig_update(dataArray, segmented, segmentedMap, valueMap, H, flipedPoints, *a, **k)
        seg, segMap, vMap, innerBnd, outerBnd, innerProb, outerProb = out
        state.update(seg=seg, segMap=segMap, vMap=vMap)
        rec['nflip'].append(0 if flipedPoints is None else len(flipedPoints))
        rec['nseg'].append(len(seg))
        rec['n_in'].append(int(np.count_nonzero((vMap == 0) | (vMap == 1))))
        rec['n_out'].append(int(np.count_nonzero((vMap == 2) | (vMap == 3))))
        rec['ni'].append(len(innerBnd))
        rec['no'].append(len(outerBnd))
        if full or ncall in prob_iters:
            ib = np.asarray(innerBnd, dtype=np.int64).reshape(-1, 3)
            ob = np.asarray(outerBnd, dtype=np.int64).reshape(-1, 3)
            ab = np.concatenate((ib, ob))
            rec['snap_iters'].append(ncall)
            rec['inner'].append(lex(ib, shape))
            rec['outer'].append(lex(ob, shape))
            rec['ip'].append(innerProb[tuple(ab.T)].astype(np.float64))
            rec['op'].append(outerProb[tuple(ab.T)].astype(np.float64))
            if full:
                rec['labels'].append(vMap.astype(np.uint8).reshape(-1).copy())
        return out

    V.update = wrapped
    buf = io.StringIO()
    capped = False
    try:
        with contextlib.redirect_stdout(buf), np.errstate(all='ignore'):
            try:
                V.variationalRegionGrowing(data, vmap, H=H, maxSegmentSize=maxSegmentSize)
            except StopSweeps:
                capped = True
    finally:
        V.update = orig_update
    seg = np.asarray(state['seg'], dtype=np.int64).reshape(-1, 3)
    out = dict(
        shape=np.asarray(shape, dtype=np.int64), H=np.float64(H),
        maxSegmentSize=np.int64(maxSegmentSize),
        max_sweeps=np.int64(-1 if max_sweeps is None else max_sweeps),
        capped=np.bool_(capped), stdout=np.str_(buf.getvalue()),
        nflip=np.asarray(rec['nflip'], np.int64), nseg=np.asarray(rec['nseg'], np.int64),
        n_in=np.asarray(rec['n_in'], np.int64), n_out=np.asarray(rec['n_out'], np.int64),
        ni=np.asarray(rec['ni'], np.int64), no=np.asarray(rec['no'], np.int64),
        final_labels=state['vMap'].astype(np.uint8), final_segmap=state['segMap'].astype(np.uint8),
        final_segmented=lex(seg, shape).astype(np.int64),
        snap_iters=np.asarray(rec['snap_iters'], np.int64),
    )

    def ragged(lst, dt):
        off = np.zeros(len(lst) + 1, np.int64)
        for i, a in enumerate(lst):
            off[i + 1] = off[i] + len(a)
        cat = np.concatenate(lst).astype(dt) if lst else np.zeros(0, dt)
        return cat, off
    out['inner_cat'], out['inner_off'] = ragged(rec['inner'], np.int32)
    out['outer_cat'], out['outer_off'] = ragged(rec['outer'], np.int32)
    # probabilities: only at a subset of the snapshots (they are incompressible float64)
    snaps = list(rec['snap_iters'])
    keep = [i for i, t in enumerate(snaps)
            if (not full) or t in PROB_ITERS_FULL or t == snaps[-1] or t in prob_iters]
    out['prob_snaps'] = np.asarray([snaps[i] for i in keep], np.int64)   # update-call index of each prob block
    ipc, boff = ragged([rec['ip'][i] for i in keep], np.float64)
    opc, _ = ragged([rec['op'][i] for i in keep], np.float64)
    out['ip_cat'], out['op_cat'], out['prob_off'] = ipc, opc, boff
    if full:
        out['labels_t'] = np.stack(rec['labels']) if rec['labels'] else np.zeros((0, data.size), np.uint8)
    return out


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ---------------------------------------------------------------- case table
def cases():
    c = {}
    # (i) the reference's own two KATs (variationalRegionGrowing.py:284-314), default kwargs
    c['kat_straight_line'] = dict(gen=P.straight_line, kw={}, run=dict(maxSegmentSize=5000), full=True, store_inputs=True)
    c['kat_sphere'] = dict(gen=P.sphere, kw={}, run=dict(maxSegmentSize=5000), full=False,
                           prob_iters=(0, 1, 5, 10), store_inputs=True)
    # (ii) BASELINE.json configs[0]
    c['config1_tube'] = dict(gen=P.config1, kw={}, run=dict(max_sweeps=50), full=False,
                             prob_iters=(0, 1, 10, 50), store_inputs=False)
    # (iii) adversarial small volumes: quirks (ghost voxels, skipped flips, omitted density terms)
    c['adv_scattered'] = dict(gen=P.scattered_seeds, kw={}, run=dict(max_sweeps=40), full=True, store_inputs=True)
    c['adv_shell'] = dict(gen=P.shell_with_holes, kw={}, run=dict(max_sweeps=40), full=True, store_inputs=True)
    for s in range(4):
        c[f'adv_noise{s}'] = dict(gen=P.noise_volume, kw=dict(shape=(16, 18, 20), seed=100 + s),
                                  run=dict(max_sweeps=30), full=True, store_inputs=True)
    # integer-level (quantised) variants: the table/histogram density path must agree too
    c['adv_noise_q'] = dict(gen=P.noise_volume, kw=dict(shape=(16, 18, 20), seed=200, levels=4),
                            run=dict(max_sweeps=30), full=True, store_inputs=True)
    c['adv_scattered_q'] = dict(gen=P.scattered_seeds, kw=dict(seed=8), run=dict(max_sweeps=40, H=1.0),
                                full=True, store_inputs=True, quant=8)
    c['tube_q_small'] = dict(gen=P.tube_phantom,
                             kw=dict(shape=(48, 40, 24), radius=3.0, seed=5, seed_planes=3, amp_y=8.0,
                                     amp_z=4.0, levels=16, brain_mask=True),
                             run=dict(max_sweeps=40), full=True, store_inputs=True)
    # float32 dataArray: under numpy 2 the reference then evaluates every Gaussian term and every np.sum in float32
    # (:149-155, :236-255).  The product computes in float64 whatever the input dtype, so these
    # fixtures MEASURE the divergence (tests/test_float32_input.py): same recipes as config1_tube / tube_q_small /
    # adv_noise_q, passed as np.float32.
    c['config1_tube_f32'] = dict(gen=P.config1, kw={}, run=dict(max_sweeps=50), full=False,
                                 prob_iters=(0, 1, 2, 3, 5, 10, 15, 20, 25, 30, 40, 50), store_inputs=False, cast=np.float32)
    c['tube_q_small_f32'] = dict(gen=P.tube_phantom,
                                 kw=dict(shape=(48, 40, 24), radius=3.0, seed=5, seed_planes=3, amp_y=8.0,
                                         amp_z=4.0, levels=16, brain_mask=True),
                                 run=dict(max_sweeps=40), full=True, store_inputs=True, cast=np.float32)
    c['adv_noise_q_f32'] = dict(gen=P.noise_volume, kw=dict(shape=(16, 18, 20), seed=200, levels=4),
                                run=dict(max_sweeps=30), full=True, store_inputs=True, cast=np.float32)
    c['adv_scattered_f32'] = dict(gen=P.scattered_seeds, kw={}, run=dict(max_sweeps=40), full=True, store_inputs=True,
                                  cast=np.float32)
    # the reference's own input class (:284-314 are binary integer volumes): binary / few-level integer fixtures, default H.
    # Exact ties of the sign test (:87) are at home here (proportional class histograms): tests/test_gpu_parity.py::
    # test_integer_class_goldens compares up to the first reported tie and records how often they occur.
    c['int_line_salt'] = dict(gen=P.straight_line_salt, kw={}, run=dict(max_sweeps=40), full=True, store_inputs=True)
    c['int_sphere_salt'] = dict(gen=P.sphere_salt, kw={}, run=dict(max_sweeps=40), full=True, store_inputs=True)
    c['int_two_tubes'] = dict(gen=P.two_touching_tubes, kw={}, run=dict(max_sweeps=60), full=True, store_inputs=True)
    c['int_torus'] = dict(gen=P.torus, kw={}, run=dict(max_sweeps=60), full=True, store_inputs=True)
    c['int_three_level'] = dict(gen=P.three_level, kw={}, run=dict(max_sweeps=40), full=True, store_inputs=True)
    # size stop + border seeds: maxSegmentSize reached, seeds on the volume faces
    c['border_size_stop'] = dict(gen=P.noise_volume, kw=dict(shape=(10, 9, 8), seed=300, p_seed=0.5, p_excl=0.1),
                                 run=dict(maxSegmentSize=300), full=True, store_inputs=True)
    return c


def main(argv):
    V = load_reference()
    table = cases()
    names = argv or list(table)
    for name in names:
        spec = table[name]
        data, vmap = spec['gen'](**spec['kw'])
        if spec.get('quant'):
            q = spec['quant']
            data = np.round(data * q) / q
        if spec.get('cast') is not None:
            assert np.array_equal(data.astype(spec['cast']).astype(np.float64), np.asarray(data, np.float64))
            data = data.astype(spec['cast'])      # the reference now takes its float32 path
        run = dict(spec['run'])
        out = run_reference(V, data, vmap, full=spec['full'], prob_iters=spec.get('prob_iters', ()), **run)
        out['input_dtype'] = np.str_(np.asarray(data).dtype.name)
        out['recipe'] = np.str_(f"{spec['gen'].__name__}({spec['kw']}) quant={spec.get('quant')}")
        out['data_sha256'] = np.str_(sha(np.asarray(data, np.float64)))
        out['labels0_sha256'] = np.str_(sha(np.asarray(vmap, np.uint8)))
        if spec['store_inputs']:
            d = np.asarray(data)
            out['data'] = d.astype(np.float32) if np.array_equal(d.astype(np.float32), d) else d.astype(np.float64)
            out['labels0'] = np.asarray(vmap, np.uint8)
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, **out)
        print(f"{name}: shape={tuple(out['shape'])} calls={len(out['nflip'])} capped={bool(out['capped'])} "
              f"nseg={out['nseg'][-1]} ni={out['ni'][-1]} no={out['no'][-1]} "
              f"-> {os.path.getsize(path) / 1024:.0f} KiB  | {str(out['stdout']).strip().splitlines()[:1]}")


if __name__ == '__main__':
    main(sys.argv[1:])
