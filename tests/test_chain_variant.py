"""The persistent band kernel (k_chain) - an EXPERIMENTAL build of the product sources (tools/build_variants.sh,
-DVRG_CHAIN; not what the package loads: measured slower than the four launches) - must still give
the reference's results: the band chain of a whole batch of sweeps inside ONE launch pinned to one elected XCD, its
phases separated by in-kernel barriers, every band-side load past L1."""
import os
import subprocess

import numpy as np
import pytest

import parity
from conftest import ROOT

pytestmark = pytest.mark.gpu
LIB = os.path.join(ROOT, 'arterynetwork_amd', 'csrc', 'libvrg_hip_chain.so')


@pytest.fixture(scope='module')
def chain_lib():
    from arterynetwork_amd._capi import VrgLib
    if not os.path.exists(LIB):
        subprocess.check_call(['bash', os.path.join(ROOT, 'tools', 'build_variants.sh')])
    return VrgLib(LIB, 'vrg_')


def test_product_build_has_no_chain_kernel():
    from arterynetwork_amd._capi import Session, VrgError
    s = Session((8, 8, 8))
    with pytest.raises(VrgError):
        s.set_option('chain_kernel', 1)
    s.set_option('chain_kernel', 0)
    s.close()


@pytest.mark.parametrize('name', ['tube_q_small', 'adv_scattered_q', 'adv_noise_q', 'adv_shell', 'kat_sphere'])
def test_chain_kernel_goldens(chain_lib, golden_loader, name):
    g = golden_loader(name)
    data, vmap = g.inputs()
    iterMax = g.max_sweeps if g.max_sweeps >= 0 else 200
    res, k = parity.run_stepwise(chain_lib, data, vmap, g.H, g.maxSegmentSize, iterMax, density_mode=1, check_hist=True,
                                 options={'chain_kernel': 1})
    assert res is not None and k == g.ncalls - 1
    res, k = parity.run_batched(chain_lib, data, vmap, g.H, g.maxSegmentSize, iterMax, density_mode=1,
                                options={'chain_kernel': 1, 'batch': 7})
    assert res is not None and k == g.ncalls - 1


def test_chain_kernel_medium_tube_and_handback(chain_lib):
    """60 sweeps in one call (batches of 16 trips = one launch each), arrays that start at 64 entries (trips handed back
    for more room end a launch early), and a flip limit that sends some trips to the host-driven path."""
    from arterynetwork_amd import phantoms
    data, vmap = phantoms.tube_phantom(shape=(160, 96, 64), radius=3.5, seed=9, seed_planes=3, amp_y=18.0,
                                       amp_z=9.0, levels=64, brain_mask=True)
    for opts in ({'batch': 16}, {'capacity_floor': 64, 'batch': 5}, {'small_flips': 20, 'batch': 8}):
        opts = dict(opts, chain_kernel=1)
        res, k = parity.run_batched(chain_lib, data, vmap, 2.25, None, 60, density_mode=1, options=opts)
        assert res is not None and k == 60
