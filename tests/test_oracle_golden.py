"""Pins the CPU oracle (oracle/vrg_oracle.c) to the real reference.

Every golden in tests/golden/ was produced by tests/golden/make_goldens.py running
/root/reference/Code/variationalRegionGrowing.py (its own two self-tests :284-314, BASELINE
config 1, and adversarial volumes).  Integer outputs (labels, list orders, counts, iteration
numbers) must match exactly; densities to 1e-11 relative (summation order differs from
np.sum's pairwise order).
"""
import numpy as np
import pytest

from conftest import golden_names
from oracle import vrg_oracle as O

FAST = [n for n in golden_names() if n not in ('config1_tube',)]


def run_case(g, density_mode):
    data, vmap = g.inputs()
    o = O.Oracle(data, vmap, g.H, density_mode)
    o.init()
    iterMax = g.max_sweeps if g.max_sweeps >= 0 else 200
    z = g.z
    snap_pos = {int(t): k for k, t in enumerate(z['snap_iters'])}
    prob_pos = {int(t): j for j, t in enumerate(z['prob_snaps'])}
    full = 'labels_t' in z.files
    call = 0
    reason = 0
    while True:
        # state after update call `call`
        if full:
            assert np.array_equal(o.labels().reshape(-1), z['labels_t'][call]), f'labels differ after call {call}'
        if call in snap_pos:
            t, gi, go = g.snapshot(snap_pos[call])
            ii, ip, op = o.band(0)
            oi, ip2, op2 = o.band(1)
            assert np.array_equal(ii, gi), f'inner list order differs after call {call}'
            assert np.array_equal(oi, go), f'outer list order differs after call {call}'
            if call in prob_pos:
                _, gip, gop = g.probs(prob_pos[call])
                # incremental updates cancel (:243-247), so tiny entries carry absolute error
                scale = max(1.0, float(np.max(np.abs(gip))), float(np.max(np.abs(gop))))
                np.testing.assert_allclose(np.concatenate((ip, ip2)), gip, rtol=1e-11, atol=1e-13 * scale)
                np.testing.assert_allclose(np.concatenate((op, op2)), gop, rtol=1e-11, atol=1e-13 * scale)
        reason = o.step(iterMax, g.maxSegmentSize, -1.0)
        if reason != 0:
            break
        call += 1
    tr = o.trace()
    assert len(tr) == g.ncalls
    for f in ('nflip', 'nseg', 'n_in', 'n_out', 'ni', 'no'):
        assert np.array_equal(tr[f], z[f]), f
    assert np.array_equal(o.labels(), z['final_labels'])
    assert np.array_equal(o.segmap(), z['final_segmap'])
    assert np.array_equal(o.segmented_lex(), z['final_segmented'])
    out = str(z['stdout'])
    if bool(z['capped']):
        assert reason == 4
    else:
        first = out.strip().splitlines()[0]
        msgs = O.finish_messages(reason, o.iterNum, len(o.segmented_lex()), int(np.count_nonzero(data)))
        assert msgs[0] == first
        assert msgs[-1] == out.strip().splitlines()[-1]
    o.close()


@pytest.mark.parametrize('name', FAST)
def test_oracle_matches_reference_bruteforce(name, golden_loader):
    run_case(golden_loader(name), density_mode=0)


@pytest.mark.parametrize('name', FAST)
def test_oracle_matches_reference_levels(name, golden_loader):
    run_case(golden_loader(name), density_mode=1)


def test_oracle_config1(golden_loader):
    """BASELINE.json configs[0]: 128x128x64 tube, 50 sweeps (survey: nseg 149 -> 2090, band 4086)."""
    g = golden_loader('config1_tube')
    run_case(g, density_mode=0)
    assert int(g.z['nseg'][0]) == 149 and int(g.z['nseg'][-1]) == 2090
    assert int(g.z['ni'][-1] + g.z['no'][-1]) == 4086


def test_reference_kats_known_answers(golden_loader):
    """The reference's only known answers (:284-314): iteration 16 / 80 voxels, iteration 11 / 4169."""
    for name, it, n in (('kat_straight_line', 16, 80), ('kat_sphere', 11, 4169)):
        data, vmap = golden_loader(name).inputs()
        vm = vmap.copy()
        seg, segMap, vm2 = O.variationalRegionGrowing(data, vm, quiet=True)
        assert vm2 is vm
        assert len(seg) == n == int(np.count_nonzero(data))
        assert all(data[tuple(seg.T)])
        o = O.Oracle(data, vmap)
        o.init()
        assert o.run(200, 5000, -1.0) == 1 and o.iterNum == it
        o.close()


def test_empty_seed_set_raises():
    data = np.zeros((4, 4, 4))
    with pytest.raises(ValueError):
        O.variationalRegionGrowing(data, np.full(data.shape, 3), quiet=True)
