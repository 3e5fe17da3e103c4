import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden_names(float32=False):
    """Golden cases; those whose dataArray went into the reference as float32 (suffix _f32: the reference then computes
    in float32, tests/test_float32_input.py) are listed separately from the float64 / integer ones."""
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, '*.npz')))
    return [n for n in names if n.endswith('_f32') == bool(float32)]


class Golden:
    """One tests/golden/<case>.npz (produced by tests/golden/make_goldens.py from the real reference)."""

    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
        self.shape = tuple(int(s) for s in self.z['shape'])
        self.H = float(self.z['H'])
        self.maxSegmentSize = int(self.z['maxSegmentSize'])
        self.max_sweeps = int(self.z['max_sweeps'])
        self.ncalls = len(self.z['nflip'])          # init + incremental update calls

    def inputs(self, as_input_dtype=False):
        """(dataArray, valueMap); float64 data unless `as_input_dtype` asks for the dtype the reference was given."""
        dt = np.dtype(str(self.z['input_dtype'])) if (as_input_dtype and 'input_dtype' in self.z.files) else np.float64
        if 'data' in self.z.files:
            return self.z['data'].astype(dt).reshape(self.shape), self.z['labels0'].astype(np.int64).reshape(self.shape)
        from arterynetwork_amd import phantoms as P
        assert self.name in ('config1_tube', 'config1_tube_f32')
        data, vmap = P.config1()
        data = data.astype(dt)
        import hashlib
        assert hashlib.sha256(np.ascontiguousarray(data, np.float64).tobytes()).hexdigest() == str(self.z['data_sha256'])
        return data, vmap

    def snapshot(self, k):
        """lists (lex idx) after the k-th stored snapshot; returns (call_index, inner, outer)."""
        z = self.z
        io, oo = z['inner_off'], z['outer_off']
        return int(z['snap_iters'][k]), z['inner_cat'][io[k]:io[k + 1]].astype(np.int64), \
            z['outer_cat'][oo[k]:oo[k + 1]].astype(np.int64)

    def probs(self, j):
        """(call_index, ip, op) of the j-th stored probability block (band order = inner ++ outer)."""
        z = self.z
        po = z['prob_off']
        return int(z['prob_snaps'][j]), z['ip_cat'][po[j]:po[j + 1]], z['op_cat'][po[j]:po[j + 1]]


@pytest.fixture(scope='session')
def golden_loader():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return load
