#!/usr/bin/env python3
"""Development tool: where one sweep of the band chain spends its time - in-kernel stamps of the diagnostic build
(csrc/libvrg_hip_stamps.so = the product sources compiled with -DVRG_STAMPS; in the product build no stamp executes).

usage: VRG_HIP_LIB=arterynetwork_amd/csrc/libvrg_hip_stamps.so python tools/chain_stamps.py SHAPE [dense_off] [samples]
(the library: tools/build_stamps.sh)
Prints, averaged over `samples` sweeps (each read after a run of 3 more sweeps), the stamps of the LAST sweep relative to
k_band's entry (us).  dense_off = 1: the band chain alone (no recount beside it)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from arterynetwork_amd import phantoms
from arterynetwork_amd._capi import Session

shape = tuple(int(v) for v in sys.argv[1].split('x'))
dense_off = int(sys.argv[2]) if len(sys.argv) > 2 else 0
samples = int(sys.argv[3]) if len(sys.argv) > 3 else 60
dev = torch.device('cuda', 0)
I, vm = phantoms.bench_volume_torch(shape, dev, levels=int(os.environ.get('VRG_STAMP_LEVELS', '255')), tubes=int(os.environ.get('VRG_STAMP_TUBES', '1')))   # (env: another quantisation; several tubes = many flips per sweep)
torch.cuda.synchronize()
LEADER = os.environ.get('VRG_LEADER', '0') != '0'        # the leader of a leader / follower group that only leads (band chain + change log, no dense pass)
if LEADER:
    from arterynetwork_amd import replica
    s = replica.make_replica_session(shape, 0, 1, transport='rccl', leader_verifies=False)
else:
    s = Session(shape)
s.set_option('batch', 64)
s.set_volume_ptr(I.data_ptr(), np.float32, list(I.stride()))
s.set_labels_ptr(vm.data_ptr(), np.uint8, list(vm.stride()))
s.init(2.25)
done = s.run(40, 10 ** 15, None).sweeps
NAMES = {0: 'k_band entry (wg 0)', 1: 'k_band state loaded', 2: 'k_band pool wg 0 done', 3: 'k_band first exact wg entry', 4: 'k_band first exact wg done',
         8: 'k_order entry', 9: 'k_order stop tests done', 10: 'k_order flips sorted', 11: 'k_order L/P bits + stamps written', 12: 'k_order prepass done',
         16: 'k_mark_relabel entry (wg 0)', 17: 'k_mark state loaded', 18: 'k_mark label byte + preload back', 19: 'k_mark mark atomics back',
         23: 'k_mark place in the marked list known', 20: 'k_mark stencil done', 21: 'k_mark events committed', 22: 'k_mark last workgroup done',
         24: 'k_close entry (wg 0)', 25: 'k_close state loaded', 26: 'k_close dense wait over', 27: 'k_close apply done (wg 0)',
         32: 'k_close first memo wg entry', 33: 'k_close memo wg levels sorted', 34: 'k_close memo wg done',
         28: 'k_close last ticket taken', 29: 'k_close finalize done'}
FUSED = os.environ.get('VRG_FUSED', '1') != '0'
s.set_option('fused', 1 if FUSED else 0)
if FUSED:      # update() as one launch (k_sweep)
    NAMES = {0: 'k_band entry (wg 0)', 1: 'k_band state loaded', 2: 'k_band pool wg 0 done', 3: 'k_band first exact wg entry', 4: 'k_band first exact wg done',
             7: 'k_band touched levels staged (wg 0)', 40: 'k_band slots decided (wg 0, before the drain)',
             8: 'k_sweep entry (wg 0)', 9: 'k_sweep state + flip records back, gate passed', 10: 'k_sweep flips ranked', 18: 'k_sweep tile + fields + flip rows back, skip rule done',
             19: 'k_sweep tile annotated', 11: 'k_sweep listed neighbours found', 20: 'k_sweep stencils done, stores drained', 21: 'k_sweep events committed, drained (wg 0)',
             28: 'k_sweep last ticket taken', 29: 'k_sweep sweep closed'}
acc = {k: [] for k in NAMES}
period = []
buf = (C.c_uint64 * 64)()
for _ in range(samples):
    if dense_off or LEADER:                                # (the handle has to be initialised again after a dense_off run)
        if not LEADER: s.set_option('dense_off', 0)
        s.set_labels_ptr(vm.data_ptr(), np.uint8, list(vm.stride())); s.init(2.25)
        if not LEADER: s.set_option('dense_off', 1)
        done = 0
    done += s.run(done + 48, 10 ** 15, None).sweeps        # whole batches: the last sweep's stamps are of a chain in steady state
    s._check(s.lib.debug_stamps(s._h, buf))
    t0 = buf[0]
    if not t0:
        raise SystemExit('no stamps: not the -DVRG_STAMPS build (set VRG_HIP_LIB)')
    if buf[6] and (dense_off or LEADER):      # (beside a dense pass the stamp of the sweep before is read stale more often than not)
        period.append((t0 - buf[6]) * 0.01)
    for k in NAMES:
        if buf[k] >= t0:
            acc[k].append((buf[k] - t0) * 0.01)
print('%s, dense pass %s: stamps of the last sweep of a 16-sweep batch, us after k_band entry (mean of %d, min..max)' % ('x'.join(map(str, shape)), 'OFF' if dense_off else 'beside', samples))
for k in sorted(NAMES, key=lambda k: (np.mean(acc[k]) if acc[k] else 1e9)):
    v = acc[k]
    if v:
        print('  %7.2f  (%6.2f .. %6.2f)  %s' % (np.mean(v), np.min(v), np.max(v), NAMES[k]))
if period:
    # (median: beside a dense pass the sweep before is now and then the last one of the run() call before - a host round trip away)
    print('  %7.2f  (%6.2f .. %6.2f)  k_band entry of the sweep before to this one = the step (MEDIAN, min..max)' % (np.median(period), np.min(period), np.max(period)))
s.close()
