"""The C-ABI library builds, loads and exports every symbol include/vrg.h declares (no GPU needed),
and the product path fails loudly - never falls back to a CPU path - when no GPU is visible."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope='module')
def product():
    from arterynetwork_amd import build
    return build.build()


def header_symbols():
    import glob
    syms = set()
    for h in glob.glob(os.path.join(ROOT, 'include', '*.h')):
        txt = re.sub(r'/\*.*?\*/', '', open(h).read(), flags=re.S)        # declarations only, not comments
        syms |= set(re.findall(r'\b((?:vrg|vmask)_[a-z_0-9]+)\s*\(', txt))
    return sorted(syms)


def test_library_exports_header_symbols(product):
    dll = ctypes.CDLL(product)
    syms = header_symbols()
    assert len(syms) >= 20 and 'vmask_edt' in syms and 'vrg_comm_init' in syms
    for name in syms:
        assert hasattr(dll, name), name


def test_hostmodel_is_not_in_the_package():
    pkg = os.path.join(ROOT, 'arterynetwork_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.h', '.hip', '.cpp')):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt, f
                assert 'libvrg_hostmodel' not in txt or f == '_capi.py' and False, f


def test_no_gpu_means_error_not_fallback(product):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU visible')
    from arterynetwork_amd import variationalRegionGrowing
    from arterynetwork_amd._capi import VrgError
    data = np.zeros((4, 4, 4))
    vm = np.full((4, 4, 4), 3)
    vm[1, 1, 1] = 0
    with pytest.raises(VrgError) as e:
        variationalRegionGrowing(data, vm, quiet=True)
    assert e.value.code == -2
