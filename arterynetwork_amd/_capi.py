"""ctypes binding of the C-ABI declared in include/vrg.h.

`VrgLib(path, prefix)` binds one shared library; the product always binds
arterynetwork_amd/csrc/libvrg_hip.so (prefix ``vrg_``).  There is no CPU fallback: if that library
is missing or no MI355X is visible, `VrgLib` / `Session` raise.
(The test suite binds its sequential host model through the same class with prefix ``vrgm_``.)
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB = os.environ.get('VRG_HIP_LIB') or os.path.join(HERE, 'csrc', 'libvrg_hip.so')   # env: A/B another build

DTYPE_CODES = {np.dtype(np.uint8): 0, np.dtype(np.int16): 1, np.dtype(np.uint16): 2, np.dtype(np.int32): 3,
               np.dtype(np.int64): 4, np.dtype(np.float32): 5, np.dtype(np.float64): 6}

STOP_NAMES = {0: 'running', 1: 'converged', 2: 'time', 3: 'size', 4: 'itermax'}

ERRORS = {-1: 'VRG_E_ARG', -2: 'VRG_E_NOGPU', -3: 'VRG_E_MEM', -4: 'VRG_E_STATE', -5: 'VRG_E_EMPTY',
          -6: 'VRG_E_INEXACT', -7: 'VRG_E_CAPACITY', -8: 'VRG_E_INTERNAL'}


class Result(C.Structure):
    _fields_ = [('stop_reason', C.c_int32), ('iter_num', C.c_int32), ('sweeps', C.c_int64),
                ('nseg', C.c_int64), ('n_in', C.c_int64), ('n_out', C.c_int64),
                ('ni', C.c_int64), ('no', C.c_int64), ('sum_in', C.c_double), ('sum_out', C.c_double),
                ('seconds', C.c_double), ('sweep_kernel_ms', C.c_double), ('sweep_launches', C.c_int64),
                ('chain_kernel_ms', C.c_double), ('chain_launches', C.c_int64),
                ('ties', C.c_int64), ('near_ties', C.c_int64)]


TRACE_DTYPE = np.dtype([('nflip', 'i8'), ('nseg', 'i8'), ('n_in', 'i8'), ('n_out', 'i8'),
                        ('ni', 'i8'), ('no', 'i8'), ('sum_in', 'f8'), ('sum_out', 'f8'), ('ties', 'i8'), ('near_ties', 'i8')])


REDUCE_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_void_p)
BCAST_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)
ALLSUM_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int64, C.c_void_p)


class VrgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('{} ({}): {}'.format(ERRORS.get(code, 'error'), code, msg))
        self.code = code


class VrgLib:
    def __init__(self, path=PRODUCT_LIB, prefix='vrg_'):
        if not os.path.exists(path):
            raise OSError('{} not found - build it with `python -c "import __graft_entry__ as g; g.build()"` '
                          '(hipcc --offload-arch=gfx950); this package has no CPU fallback'.format(path))
        self.path = path
        self.dll = C.CDLL(path)
        p = C.c_void_p
        i64p = C.POINTER(C.c_int64)

        def fn(name, argtypes, restype=C.c_int):
            f = getattr(self.dll, prefix + name)
            f.argtypes = argtypes
            f.restype = restype
            return f
        self.create = fn('create', [C.c_int64, C.c_int64, C.c_int64, C.c_int, C.POINTER(p)])
        self.destroy = fn('destroy', [p], None)
        self.last_error = fn('last_error', [p], C.c_char_p)
        self.set_option = fn('set_option', [p, C.c_char_p, C.c_int64])
        self.set_volume = fn('set_volume', [p, p, C.c_int, i64p])
        self.set_labels = fn('set_labels', [p, p, C.c_int, i64p])
        self.init = fn('init', [p, C.c_double])
        self.run = fn('run', [p, C.c_int64, C.c_int64, C.c_double, C.POINTER(Result)])
        self.get_labels = fn('get_labels', [p, p, C.c_int, i64p])
        self.get_segmented = fn('get_segmented', [p, p, C.c_int64, i64p])
        self.get_segmented_map = fn('get_segmented_map', [p, p, C.c_int, i64p])
        self.get_band = fn('get_band', [p, C.c_int, p, p, p, C.c_int64, i64p])
        self.get_trace = fn('get_trace', [p, p, C.c_int64, i64p])
        self.get_levels = fn('get_levels', [p, p, p, p, p, p, C.c_int64, i64p])
        self.get_stats = fn('get_stats', [p, i64p, C.c_int64])
        self.debug_stamps = fn('debug_stamps', [p, p])
        self.debug_stamps_wide = fn('debug_stamps_wide', [p, p, C.c_int64])
        self.set_slab = fn('set_slab', [p, C.c_int64, C.c_int64])
        self.comm_unique_id = fn('comm_unique_id', [p])
        self.comm_init = fn('comm_init', [p, C.c_int, C.c_int, p])
        self.set_reduce_callback = fn('set_reduce_callback', [p, REDUCE_FN, p])
        self.repl_init = fn('repl_init', [p, C.c_int, C.c_int, C.c_int])
        self.repl_set_callbacks = fn('repl_set_callbacks', [p, BCAST_FN, ALLSUM_FN, p])
        self.repl_use_rccl = fn('repl_use_rccl', [p])
        self.repl_ipc_export = fn('repl_ipc_export', [p, p, C.c_int64, i64p])
        self.repl_ipc_import = fn('repl_ipc_import', [p, p, C.c_int64])
        self.repl_stats = fn('repl_stats', [p, i64p, C.c_int64])


_product = None


def product_lib():
    global _product
    if _product is None:
        _product = VrgLib()
    return _product


def _as_supported(a):
    """Return an array with a dtype the C-ABI understands and C- or F-contiguous memory."""
    a = np.asarray(a)
    if a.dtype not in DTYPE_CODES:
        if a.dtype == np.bool_ or a.dtype == np.int8:
            a = a.astype(np.int16)
        elif a.dtype in (np.dtype(np.uint32), np.dtype(np.uint64)):
            a = a.astype(np.int64)
        elif a.dtype == np.float16:
            a = a.astype(np.float32)
        else:
            a = a.astype(np.float64)
    if not (a.flags.c_contiguous or a.flags.f_contiguous):
        a = np.ascontiguousarray(a)
    return a


def _strides(a):
    return (C.c_int64 * 3)(*[s // a.itemsize for s in a.strides])


class Session:
    """One volume on one GPU: thin object wrapper over the C-ABI handle."""

    def __init__(self, shape, device=0, lib=None):
        self.lib = lib or product_lib()
        if len(shape) != 3:
            raise ValueError('variational region growing expects a 3-D volume')
        self.shape = tuple(int(s) for s in shape)
        self._h = C.c_void_p()
        rc = self.lib.create(*self.shape, int(device), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise VrgError(rc, 'vrg_create failed (is a HIP device visible? the product has no CPU path)')

    def close(self):
        if getattr(self, '_h', None):
            self.lib.destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise VrgError(rc, self.lib.last_error(self._h).decode())

    def set_option(self, name, value):
        self._check(self.lib.set_option(self._h, name.encode(), int(value)))

    def set_volume(self, a):
        a = _as_supported(a)
        if a.shape != self.shape:
            raise ValueError('dataArray shape {} != {}'.format(a.shape, self.shape))
        self._check(self.lib.set_volume(self._h, a.ctypes.data, DTYPE_CODES[a.dtype], _strides(a)))

    def set_labels(self, a):
        a = _as_supported(a)
        if a.shape != self.shape:
            raise ValueError('valueMap shape {} != dataArray shape {}'.format(a.shape, self.shape))
        self._check(self.lib.set_labels(self._h, a.ctypes.data, DTYPE_CODES[a.dtype], _strides(a)))

    def set_volume_ptr(self, ptr, dtype, strides_elems):
        """Device-resident input (e.g. torch tensor .data_ptr()); strides in elements, order (x,y,z)."""
        self._check(self.lib.set_volume(self._h, C.c_void_p(ptr), DTYPE_CODES[np.dtype(dtype)],
                                        (C.c_int64 * 3)(*strides_elems)))

    def set_labels_ptr(self, ptr, dtype, strides_elems):
        self._check(self.lib.set_labels(self._h, C.c_void_p(ptr), DTYPE_CODES[np.dtype(dtype)],
                                        (C.c_int64 * 3)(*strides_elems)))

    # ---- multi-GPU ------------------------------------------------------------------------------
    def set_slab(self, z0, z1):
        self._check(self.lib.set_slab(self._h, int(z0), int(z1)))

    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        self._check(self.lib.comm_unique_id(buf))
        return buf.raw

    def comm_init(self, nranks, rank, id128):
        buf = C.create_string_buffer(bytes(id128), 128)
        self._check(self.lib.comm_init(self._h, int(nranks), int(rank), buf))

    def set_reduce_callback(self, fn):
        """fn(list_of_4_floats) -> list_of_4_floats (global totals); kept alive by the session."""
        def thunk(ptr, _user):
            out = fn([ptr[0], ptr[1], ptr[2], ptr[3]])
            for i in range(4):
                ptr[i] = float(out[i])
        self._reduce_thunk = REDUCE_FN(thunk)
        self._check(self.lib.set_reduce_callback(self._h, self._reduce_thunk, None))

    # ---- multi-GPU, leader / follower replication (include/vrg.h) -----------------------------------
    def repl_init(self, nranks, rank, leader_verifies=True):
        self._check(self.lib.repl_init(self._h, int(nranks), int(rank), 1 if leader_verifies else 0))

    def repl_set_callbacks(self, bcast, allsum):
        """bcast(bytearray-like memoryview, root) fills / sends the buffer in place; allsum(list of floats) -> summed list."""
        def b_thunk(ptr, nbytes, root, _user):
            bcast((C.c_uint8 * nbytes).from_address(ptr), int(root))

        def s_thunk(ptr, n, _user):
            out = allsum([ptr[i] for i in range(n)])
            for i in range(n):
                ptr[i] = float(out[i])
        self._repl_thunks = (BCAST_FN(b_thunk), ALLSUM_FN(s_thunk))
        self._check(self.lib.repl_set_callbacks(self._h, self._repl_thunks[0], self._repl_thunks[1], None))

    def repl_use_rccl(self):
        self._check(self.lib.repl_use_rccl(self._h))

    def repl_ipc_export(self):
        buf = C.create_string_buffer(256)
        n = C.c_int64()
        self._check(self.lib.repl_ipc_export(self._h, buf, 256, C.byref(n)))
        return buf.raw[:n.value]

    def repl_ipc_import(self, blob):
        buf = C.create_string_buffer(bytes(blob), len(blob))
        self._check(self.lib.repl_ipc_import(self._h, buf, len(blob)))

    def repl_stats(self):
        a = (C.c_int64 * 9)()
        self._check(self.lib.repl_stats(self._h, a, 9))
        return {'batches': a[0], 'records': a[1], 'sweeps': a[2], 'verified': a[3], 'last_verified': a[4],
                'transport': {0: 'none', 1: 'callback', 2: 'rccl', 3: 'ipc'}[a[5]], 'verifiers': a[6], 'slot': a[7], 'chunks': a[8]}

    def init(self, H=2.25):
        self._check(self.lib.init(self._h, float(H)))

    def run(self, iterMax=200, maxSegmentSize=5000, maxTime=120.0):
        r = Result()
        self._check(self.lib.run(self._h, int(iterMax), int(maxSegmentSize),
                                 -1.0 if maxTime is None else float(maxTime), C.byref(r)))
        return r

    def labels(self, out=None, dtype=np.uint8):
        """Labels 0..4; written into `out` (any supported dtype / C or F layout) when given."""
        if out is None:
            out = np.empty(self.shape, dtype=dtype)
        if out.dtype in DTYPE_CODES and (out.flags.c_contiguous or out.flags.f_contiguous):
            self._check(self.lib.get_labels(self._h, out.ctypes.data, DTYPE_CODES[out.dtype], _strides(out)))
        else:
            tmp = np.empty(self.shape, dtype=np.uint8)
            self._check(self.lib.get_labels(self._h, tmp.ctypes.data, 0, _strides(tmp)))
            out[...] = tmp
        return out

    def segmented_map(self, out=None, dtype=np.int64):
        """segmentedMap of the reference (:31-32): 1 where the label is 0 or 1; written into `out` when given."""
        if out is None:
            out = np.empty(self.shape, dtype=dtype)
        self._check(self.lib.get_segmented_map(self._h, out.ctypes.data, DTYPE_CODES[out.dtype], _strides(out)))
        return out

    def segmented(self):
        n = C.c_int64()
        self._check(self.lib.get_segmented(self._h, None, 0, C.byref(n)))
        out = np.empty((n.value, 3), np.int64)
        if n.value:
            self._check(self.lib.get_segmented(self._h, out.ctypes.data, n.value, C.byref(n)))
        return out

    def band(self, which):
        n = C.c_int64()
        self._check(self.lib.get_band(self._h, which, None, None, None, 0, C.byref(n)))
        co = np.empty((n.value, 3), np.int64)
        ip = np.empty(n.value, np.float64)
        op = np.empty(n.value, np.float64)
        if n.value:
            self._check(self.lib.get_band(self._h, which, co.ctypes.data, ip.ctypes.data, op.ctypes.data,
                                          n.value, C.byref(n)))
        return co, ip, op

    def trace(self):
        n = C.c_int64()
        self._check(self.lib.get_trace(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, TRACE_DTYPE)
        self._check(self.lib.get_trace(self._h, out.ctypes.data, n.value, C.byref(n)))
        return out

    def stats(self):
        """Diagnostics: how often a trip was handed back to the host and why, array capacities."""
        a = (C.c_int64 * 23)()
        self._check(self.lib.get_stats(self._h, a, 23))
        return {'bail_flips': a[1], 'grow_marks': a[2], 'grow_pool': a[3], 'host_driven_trips': a[4],
                'fused_trips': a[15], 'bail_fuse': a[16], 'density_bins': a[17], 'memo_trips': a[18],
                'data_nonzero': a[19], 'slow_flips': a[22], 'bin_bytes': a[20], 'level_index_bytes': a[21], 'pool_capacity': a[5], 'mark_capacity': a[6], 'pool_slots': a[7], 'dense_bytes': a[8],
                'dense_kernel': ('k_recount_pipe<3,{}>'.format('true' if a[9] else 'false') if a[14] else
                                 'k_recount_bits<{},{},{},{}>'.format(2 if a[10] == 2 else 3, 'true' if a[9] else 'false', a[10], 'true' if a[12] else 'false')),
                'dense_nt_loads': bool(a[9]), 'dense_workgroups': a[11], 'dense_listed_units': a[13]}

    def nlevels(self):
        """Number of distinct intensity values of the volume (after init)."""
        n = C.c_int64()
        self._check(self.lib.get_levels(self._h, None, None, None, None, None, 0, C.byref(n)))
        return n.value

    def chain_timing(self, H=2.25, sweeps=100):
        """Measurement aid for bench.py: the band chain of a sweep timed ALONE (option dense_off: the dense recount
        is not launched), `sweeps` more sweeps from the current state.  The session has to be re-initialised
        afterwards.  Returns {} when nothing could be measured (run stopped)."""
        import time
        try:
            self.set_option('events', 0)
            self.set_option('dense_off', 1)
        except VrgError:
            return {}
        it0 = len(self.trace()) - 1
        t0 = time.perf_counter()
        r = self.run(it0 + sweeps, 10 ** 15, None)
        dt = time.perf_counter() - t0
        if r.sweeps <= 0:
            return {}
        return {'band_chain_ms': round(dt / r.sweeps * 1e3, 4), 'band_chain_sweeps': int(r.sweeps),
                'band_chain_note': 'stream A alone (dense recount not launched), wall time per sweep incl. host enqueue'}

    def levels(self, recount=True):
        n = C.c_int64()
        self._check(self.lib.get_levels(self._h, None, None, None, None, None, 0, C.byref(n)))
        L = n.value
        vals = np.empty(L, np.float64)
        hin = np.empty(L, np.int32)
        hout = np.empty(L, np.int32)
        rin = np.empty(L, np.int32)
        rout = np.empty(L, np.int32)
        self._check(self.lib.get_levels(self._h, vals.ctypes.data, hin.ctypes.data, hout.ctypes.data,
                                        rin.ctypes.data if recount else None,
                                        rout.ctypes.data if recount else None, L, C.byref(n)))
        return vals, hin, hout, (rin if recount else None), (rout if recount else None)
