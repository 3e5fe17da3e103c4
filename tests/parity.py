"""Step-by-step comparison of a C-ABI implementation (HIP library, or the test host model) with the oracle.

Integer state (labels, band list orders, `segmented` order, counts, stop reason / iteration) must be
identical; band densities and region intensity sums within `rtol` (summation order differs).
"""
import numpy as np

from arterynetwork_amd._capi import Session
from oracle import vrg_oracle as O


def lex_of(coords, shape):
    c = np.asarray(coords, np.int64).reshape(-1, 3)
    return (c[:, 0] * shape[1] + c[:, 1]) * shape[2] + c[:, 2]


def assert_probs_close(a, b, rtol, what):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, what
    if a.size == 0:
        return
    scale = max(1.0, float(np.max(np.abs(b))))
    np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol * 1e-2 * scale, err_msg=what)


def density_rtol(data, options=None):
    """Tolerance of the band densities against the oracle: 1e-9 where the implementation sums over the distinct values like the
    oracle's level mode does (different association order only); 1e-6 where it evaluates the exact densities through bin
    moments (level tables above option bin_above, 2048 by default: truncation bound 2e-8, vrg_items.h) - north_star allows 1e-5."""
    above = (options or {}).get('bin_above', 2048)
    return 1e-6 if len(np.unique(data)) > above else 1e-9


def compare_state(s, o, shape, rtol, tag, check_lists=True):
    assert np.array_equal(s.labels(), o.labels()), f'{tag}: labels differ'
    if check_lists:
        for which in (0, 1):
            co, ip, op = s.band(which)
            oi, oip, oop = o.band(which)
            assert np.array_equal(lex_of(co, shape), oi), f'{tag}: band list {which} order differs'
            assert_probs_close(ip, oip, rtol, f'{tag}: innerProb list {which}')
            assert_probs_close(op, oop, rtol, f'{tag}: outerProb list {which}')


def decision_margin(o):
    """Smallest relative margin |inner/innerSize - outer/outerSize| of the pending sign tests (:87).
    A margin at rounding level means the reference's own outcome is decided by summation order."""
    nin, nout = o.sizes()
    m = np.inf
    for which in (0, 1):
        _, ip, op = o.band(which)
        if len(ip) == 0:
            continue
        if nin == 0 or nout == 0:      # a region vanished: x / 0 with x a rounding residue of the incremental corrections -
            return 0.0                 # its SIGN (+-1e-16 -> +-inf) decides in the reference: summation-order dependent
        a, b = ip / nin, op / nout
        rel = np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-300)
        m = min(m, float(rel.min()))
    return m


def run_stepwise(lib, data, vmap, H=2.25, maxSegmentSize=None, iterMax=200, density_mode=1, rtol=None,
                 every=1, options=None, check_hist=False, device=0, tie_tol=1e-11):
    """Advance implementation and oracle one sweep at a time; compare after each. Returns (session result, sweeps)."""
    shape = data.shape
    if rtol is None:
        rtol = density_rtol(data, options)
    if maxSegmentSize is None:
        maxSegmentSize = data.size + 1
    o = O.Oracle(data, vmap, H, density_mode)
    o.init()
    s = Session(shape, device=device, lib=lib)
    for k, v in (options or {}).items():
        s.set_option(k, v)
    s.set_volume(data)
    s.set_labels(vmap)
    s.init(H)
    compare_state(s, o, shape, rtol, 'init')
    k = 0
    res = None
    while True:
        # An exact tie of the sign test (:87) is not a parity question: the reference decides it by np.sum's rounding
        # (DESIGN.md section 2).  The implementation REPORTS such decisions (vrg_result.ties); the case is cut short when
        # it does - or when the oracle's own margin says so - and a tie the oracle sees clearly must have been counted.
        margin = decision_margin(o)
        rc = o.step(iterMax, maxSegmentSize, -1.0)
        res = s.run(min(k + 1, iterMax), maxSegmentSize, None)
        if res.ties > 0 or margin < tie_tol:
            assert res.ties > 0 or margin > 1e-13 or k >= iterMax, f'oracle margin {margin} at sweep {k} but no tie reported'
            s.close()
            o.close()
            return None, k
        if rc != 0:
            assert res.stop_reason == rc, f'stop reason {res.stop_reason} != oracle {rc} at sweep {k}'
            assert res.iter_num == o.iterNum
            break
        k += 1
        assert res.sweeps == 1 and res.iter_num == k + 1, (res.sweeps, res.iter_num, k)
        if k % every == 0:
            compare_state(s, o, shape, rtol, f'sweep {k}')
    compare_state(s, o, shape, rtol, 'final')
    assert np.array_equal(lex_of(s.segmented(), shape), o.segmented_lex()), 'segmented order differs'
    tr, otr = s.trace(), o.trace()
    assert len(tr) == len(otr)
    for f in ('nflip', 'nseg', 'n_in', 'n_out', 'ni', 'no'):
        assert np.array_equal(tr[f], otr[f]), f
    np.testing.assert_allclose(tr['sum_in'], otr['sum_in'], rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(tr['sum_out'], otr['sum_out'], rtol=1e-9, atol=1e-6)
    if check_hist:
        vals, hin, hout, rin, rout = s.levels()
        assert np.array_equal(hin, rin) and np.array_equal(hout, rout), 'incremental class histograms drifted'
    s.close()
    o.close()
    return res, k


def run_batched(lib, data, vmap, H=2.25, maxSegmentSize=None, iterMax=200, density_mode=1, rtol=None,
                options=None, device=0, tie_tol=1e-11):
    """Oracle sweep by sweep to its stop; the implementation in ONE vrg_run call (sweeps enqueued in batches, the
    dense pass trailing the band kernels); compare final state, list orders and the whole trace.
    Returns (result, sweeps) or (None, k) when the oracle met an exact tie."""
    shape = data.shape
    if rtol is None:
        rtol = density_rtol(data, options)
    if maxSegmentSize is None:
        maxSegmentSize = data.size + 1
    o = O.Oracle(data, vmap, H, density_mode)
    o.init()
    k = 0
    while True:
        if decision_margin(o) < tie_tol:
            o.close()
            return None, k
        rc = o.step(iterMax, maxSegmentSize, -1.0)
        if rc != 0:
            break
        k += 1
    s = Session(shape, device=device, lib=lib)
    for kk, v in (options or {}).items():
        s.set_option(kk, v)
    s.set_volume(data)
    s.set_labels(vmap)
    s.init(H)
    res = s.run(iterMax, maxSegmentSize, None)
    assert res.ties == 0, 'the oracle saw no tie in this run, the implementation counted {}'.format(res.ties)
    assert int(s.trace()['ties'].sum()) == 0
    assert res.stop_reason == rc and res.iter_num == o.iterNum and res.sweeps == k, (res.stop_reason, rc, res.sweeps, k)
    compare_state(s, o, shape, rtol, 'final (batched)')
    assert np.array_equal(lex_of(s.segmented(), shape), o.segmented_lex()), 'segmented order differs'
    tr, otr = s.trace(), o.trace()
    assert len(tr) == len(otr)
    for f in ('nflip', 'nseg', 'n_in', 'n_out', 'ni', 'no'):
        assert np.array_equal(tr[f], otr[f]), f
    np.testing.assert_allclose(tr['sum_in'], otr['sum_in'], rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(tr['sum_out'], otr['sum_out'], rtol=1e-9, atol=1e-6)
    vals, hin, hout, rin, rout = s.levels()
    assert np.array_equal(hin, rin) and np.array_equal(hout, rout), 'incremental class histograms drifted'
    s.close()
    o.close()
    return res, k
