"""ctypes binding for the CPU oracle (oracle/vrg_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg - never by the product package.

`variationalRegionGrowing` below mirrors the reference's signature and return tuple
(variationalRegionGrowing.py:10-37) on top of the C restatement so parity tests can call
oracle and product the same way.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'libvrg_oracle.so')
LIB_OMP = os.path.join(HERE, 'libvrg_oracle_omp.so')     # same source built with -fopenmp (dense loops on all cores)

STOP_NAMES = {1: 'converged', 2: 'time', 3: 'size', 4: 'itermax'}


class Trace(C.Structure):
    _fields_ = [('nflip', C.c_int64), ('nseg', C.c_int64), ('n_in', C.c_int64), ('n_out', C.c_int64),
                ('ni', C.c_int64), ('no', C.c_int64), ('sum_in', C.c_double), ('sum_out', C.c_double)]


def build(force=False, omp=False):
    src = os.path.join(HERE, 'vrg_oracle.c')
    out = LIB_OMP if omp else LIB
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', HERE, '-s', os.path.basename(out)])
    return out


_libs = {}


def lib(omp=False):
    _lib = _libs.get(bool(omp))
    if _lib is None:
        L = C.CDLL(build(omp=omp))
        L.vrgo_create.restype = C.c_void_p
        L.vrgo_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_int]
        L.vrgo_destroy.argtypes = [C.c_void_p]
        L.vrgo_init.argtypes = [C.c_void_p]
        L.vrgo_step.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_double]
        L.vrgo_run.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_double]
        for f in ('iter_num', 'nseg', 'ninner', 'nouter', 'inner_size', 'outer_size', 'nlevels', 'ntrace'):
            getattr(L, 'vrgo_' + f).restype = C.c_int64
            getattr(L, 'vrgo_' + f).argtypes = [C.c_void_p]
        L.vrgo_get_labels.argtypes = [C.c_void_p, C.c_void_p]
        L.vrgo_get_segmap.argtypes = [C.c_void_p, C.c_void_p]
        L.vrgo_get_segmented.argtypes = [C.c_void_p, C.c_void_p]
        L.vrgo_get_band.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vrgo_get_trace.argtypes = [C.c_void_p, C.c_void_p]
        L.vrgo_set_threads.argtypes = [C.c_int]
        L.vrgo_set_threads.restype = C.c_int
        _lib = _libs[bool(omp)] = L
    return _lib


class Oracle:
    """Stateful handle: init() then step()/run(); arrays are exchanged in the reference's
    [x][y][z] C order ("lex" indices)."""

    def __init__(self, dataArray, valueMap, H=2.25, density_mode=0, omp=False):
        self.shape = tuple(int(s) for s in dataArray.shape)
        data = np.ascontiguousarray(dataArray, dtype=np.float64)
        lab = np.ascontiguousarray(valueMap, dtype=np.uint8)
        self._L = lib(omp)
        self._h = self._L.vrgo_create(*self.shape, data.ctypes.data, lab.ctypes.data, float(H), int(density_mode))
        if not self._h:
            raise MemoryError('vrgo_create failed')
        self.V = int(np.prod(self.shape))

    def close(self):
        if self._h:
            self._L.vrgo_destroy(self._h)
            self._h = None

    __del__ = close

    def set_threads(self, n):
        """Threads of the dense loops (all-cores build, omp=True); returns the number in effect."""
        return self._L.vrgo_set_threads(int(n))

    def init(self):
        rc = self._L.vrgo_init(self._h)
        if rc != 0:
            raise ValueError('oracle init failed (empty seed set?)')

    def step(self, iterMax=200, maxSegmentSize=5000, maxSeconds=-1.0):
        return self._L.vrgo_step(self._h, iterMax, maxSegmentSize, maxSeconds)

    def run(self, iterMax=200, maxSegmentSize=5000, maxSeconds=-1.0):
        return self._L.vrgo_run(self._h, iterMax, maxSegmentSize, maxSeconds)

    @property
    def iterNum(self):
        return self._L.vrgo_iter_num(self._h)

    @property
    def nlevels(self):
        return self._L.vrgo_nlevels(self._h)

    def labels(self):
        out = np.empty(self.shape, np.uint8)
        self._L.vrgo_get_labels(self._h, out.ctypes.data)
        return out

    def segmap(self):
        out = np.empty(self.shape, np.uint8)
        self._L.vrgo_get_segmap(self._h, out.ctypes.data)
        return out

    def segmented_lex(self):
        n = self._L.vrgo_nseg(self._h)
        out = np.empty(n, np.int64)
        self._L.vrgo_get_segmented(self._h, out.ctypes.data)
        return out

    def band(self, which):
        n = (self._L.vrgo_nouter if which else self._L.vrgo_ninner)(self._h)
        idx = np.empty(n, np.int64)
        ip = np.empty(n, np.float64)
        op = np.empty(n, np.float64)
        self._L.vrgo_get_band(self._h, which, idx.ctypes.data, ip.ctypes.data, op.ctypes.data)
        return idx, ip, op

    def sizes(self):
        return self._L.vrgo_inner_size(self._h), self._L.vrgo_outer_size(self._h)

    def trace(self):
        n = self._L.vrgo_ntrace(self._h)
        arr = (Trace * n)()
        self._L.vrgo_get_trace(self._h, arr)
        return np.array([(t.nflip, t.nseg, t.n_in, t.n_out, t.ni, t.no, t.sum_in, t.sum_out) for t in arr],
                        dtype=[('nflip', 'i8'), ('nseg', 'i8'), ('n_in', 'i8'), ('n_out', 'i8'),
                               ('ni', 'i8'), ('no', 'i8'), ('sum_in', 'f8'), ('sum_out', 'f8')])


def unlex(idx, shape):
    idx = np.asarray(idx, np.int64)
    return np.stack(np.unravel_index(idx, shape), axis=1).astype(np.int64).reshape(-1, 3)


def finish_messages(reason, iterNum, nseg, nonzero, segmented=None):
    """The strings the reference prints on each exit path (:94-95, :98-99, :102-103, :118-120)."""
    tail = 'Total segmented voxels: {}/{}'.format(nseg, nonzero)
    if reason == 1:
        return ['Finished at iteration {}'.format(iterNum), tail]
    if reason == 2:
        return ['Finished at iteration {} (Max time reached)'.format(iterNum), tail]
    if reason == 3:
        return ['Finished at iteration {} (Max segment size reached)'.format(iterNum), tail]
    return ['Segmented points are: \n {}'.format(segmented),
            'Max iteration reached! Finished at iteration {}'.format(iterNum), tail]


def variationalRegionGrowing(dataArray, valueMap, H=2.25, maxSegmentSize=5000, *, iterMax=200,
                             maxTime=120.0, density_mode=0, quiet=False):
    """Oracle with the reference's call signature / return tuple (variationalRegionGrowing.py:10)."""
    o = Oracle(dataArray, valueMap, H, density_mode)
    try:
        o.init()
        reason = o.run(iterMax, maxSegmentSize, maxTime)
        seg = unlex(o.segmented_lex(), o.shape)
        segMap = o.segmap().astype(np.int64)
        valueMap[...] = o.labels()                     # the reference mutates valueMap in place
        if not quiet:
            for line in finish_messages(reason, o.iterNum, len(seg), int(np.count_nonzero(dataArray)), seg):
                print(line)
        return seg, segMap, valueMap
    finally:
        o.close()
