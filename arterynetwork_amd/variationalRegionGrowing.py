"""MI355X-native drop-in for the reference's Code/variationalRegionGrowing.py.

Same function name, positional/keyword signature, return tuple, dtypes, in-place mutation of
``valueMap`` and printed messages as the reference (variationalRegionGrowing.py:10-121); the work
runs in hand-written HIP kernels behind the C-ABI of include/vrg.h (ctypes, see _capi.py).
There is no CPU path: without libvrg_hip.so and a visible MI355X the call raises.

Extra knobs are keyword-only and default to the reference's hard-coded constants
(iterMax=200 at :56, 120 s wall-clock cap at :97).
"""
from __future__ import annotations

import warnings

import numpy as np

from ._capi import Session, VrgError, STOP_NAMES

A = (2 * np.pi) ** (-0.5)   # :7 (kept for callers that import it)


def _finish_messages(reason, iterNum, nseg, nonzero, segmented):
    """The strings printed on each exit path (:94-95, :98-99, :102-103, :118-120)."""
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
                             maxTime=120.0, device=0, trace=None, quiet=False, options=None, verify_every=1):
    """
    Variational region growing (https://ieeexplore.ieee.org/document/7096420) on an MI355X.

    Parameters
    ----------
    dataArray : ndarray
        The data volume to which the algorithm is applied (3-D, any numeric dtype).  Kept on the device as
        float32 when every value is exactly representable in float32 (integer-valued and float32 data are),
        as float64 otherwise; the arithmetic is float64 either way.
    valueMap : ndarray
        Initial settings, same shape: 0: inside (seed), 3: outside, 4: excluded.
        Mutated in place and returned, like the reference does.
    H : float
        Kernel parameter; larger H leads to smaller segmentation.
    maxSegmentSize : int
        Stop once this many voxels are segmented (checked before a sweep is applied, :101).
    iterMax, maxTime : keyword-only
        The reference's hard-coded 200 iterations (:56) and 120 s (:97); ``maxTime=None`` disables.
    device : int
        HIP device ordinal.
    verify_every : int, keyword-only
        The reference recounts innerSize / outerSize densely after every iteration (:113-116).  Here the decisions read sizes
        kept by increments as labels change, and the dense pass over every voxel only CHECKS them (and supplies the trace's
        intensity sums): 1 (default) runs it after every sweep like the reference; n > 1 after every n-th sweep; 0 never -
        the last sweep of the call is then checked when the call ends, so a run never returns unchecked.  Results are
        identical for every value; at 880x880x640 a sweep costs 0.19 ms with the pass and about 0.03 ms without.
    trace : list, optional
        If given, receives one dict per update() call (0 = init): nflip, nseg, n_in, n_out, ni, no,
        sum_in, sum_out.

    Returns
    -------
    segmented : ndarray (N, 3) int64
        Coordinates of the segmented voxels in the reference's list order.
    segmentedMap : ndarray int64
        1 for segmented voxels, 0 for background.
    valueMap : ndarray
        The same object that was passed in; 0: inside, 1: innerbnd, 2: outerbnd, 3: outside, 4: excluded.
    """
    dataArray = np.asarray(dataArray)
    if not isinstance(valueMap, np.ndarray):
        raise TypeError('valueMap must be a numpy array (it is updated in place)')
    if dataArray.ndim != 3 or valueMap.shape != dataArray.shape:
        raise ValueError('dataArray and valueMap must be 3-D arrays of the same shape')
    with Session(dataArray.shape, device=device) as s:
        for k, v in (options or {}).items():
            s.set_option(k, v)
        if verify_every != 1:
            if int(verify_every) != verify_every or verify_every < 0:
                raise ValueError('verify_every must be 0 (never), 1 (every sweep) or n > 1 (every n-th sweep)')
            s.set_option('verify_every', int(verify_every))
        try:
            s.set_volume(dataArray)
            s.set_labels(valueMap)
            s.init(H)
        except VrgError as e:
            if e.code in (-1, -5, -6):       # argument problems surface as ValueError like numpy's would
                raise ValueError(str(e)) from None
            raise
        r = s.run(iterMax, maxSegmentSize, maxTime)
        if r.ties:
            warnings.warn('{} sign test(s) (variationalRegionGrowing.py:87) were exact ties (relative margin < 1e-11, or an empty '
                          'region): the reference decides those by the rounding of np.sum, so the labels may differ from its '
                          'at the voxels concerned'.format(r.ties), RuntimeWarning, stacklevel=2)
        if r.near_ties and dataArray.dtype == np.float32:
            warnings.warn('{} sign test(s) had a relative margin below 2e-5: for float32 dataArray the reference computes in '
                          'float32 and may decide those differently (this library computes in float64); an indicator, not a '
                          'certificate'.format(r.near_ties),
                          RuntimeWarning, stacklevel=2)
        # (volumes with more than 2048 distinct values evaluate exact densities through intensity bins, proved relative error 2e-8: every
        #  band entry carries that bound, and a sign test it could turn is counted in r.ties - the warning above covers it)
        segmented = s.segmented()
        s.labels(out=valueMap)               # in place, caller's dtype
        segmentedMap = s.segmented_map(np.empty(dataArray.shape, np.int64, order='F' if (valueMap.flags.f_contiguous and not valueMap.flags.c_contiguous) else 'C'))
        nonzero = s.stats()['data_nonzero']  # np.count_nonzero(dataArray), counted on the device when the volume went in
        if trace is not None:
            tr = s.trace()
            trace.extend({k: tr[k][i].item() for k in tr.dtype.names} for i in range(len(tr)))
    if not quiet:
        for line in _finish_messages(r.stop_reason, r.iter_num, segmented.shape[0],
                                     int(nonzero), segmented):
            print(line)
    return segmented, segmentedMap, valueMap
