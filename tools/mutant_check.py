#!/usr/bin/env python3
"""Do the race campaigns have teeth?  (development tool, GPU box)   usage: tools/mutant_check.py [reps]
Builds the product sources with -DVRG_MUTANT (band_deferred_done: the FIRST deferred workgroup to arrive raises the dense
pass's request instead of the last - labels and class bits may not be in place when the pass reads them), once quiet and once
with the random delays of -DVRG_CHAOS, and repeats one whole run of a 96x80x64 tube volume on each: every repetition must be
caught (the recount's cross-check against the incremental sizes raises VRG_E_INTERNAL, or the result differs from the
product build's)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'arterynetwork_amd', 'csrc')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40

CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from arterynetwork_amd import phantoms
from arterynetwork_amd._capi import Session
d, v = phantoms.bench_volume((96, 80, 64), seed=5)
import hashlib, os
caught = clean = 0
REF = os.environ.get('VRG_MUTANT_REF')                  # (the product build's result: written by the first child, read by the others)
ref = open(REF).read().strip() if REF and os.path.exists(REF) else None
for r in range(%d):
    s = Session(d.shape)
    s.set_volume(d.astype(np.float32)); s.set_labels(v.astype(np.uint8)); s.init(2.25)
    try:
        res = s.run(60, 10 ** 9, None)
        lab = s.labels().copy(); tr = s.trace()
        key = hashlib.sha256(lab.tobytes() + tr['n_in'].tobytes() + tr['n_out'].tobytes()).hexdigest()
        if ref is None:
            ref = key
            if REF: open(REF, 'w').write(key)
        if key != ref: caught += 1
        else: clean += 1
    except Exception as e:
        caught += 1
    finally:
        try: s.close()
        except Exception: pass
print('RESULT caught', caught, 'clean', clean)
'''

def build(name, flags):
    out = os.path.join(CSRC, name)
    srcs = [os.path.join(CSRC, f) for f in ('vrg_device.hip', 'vrg_items.h', 'vrg_types.h', 'vrg_engine.cpp')]
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(f) for f in srcs):
        return out                                      # (built where the sources were edited: the library travels with the tree)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared'] + flags + ['-o', out,
                           'vrg_device.hip', 'vrg_engine.cpp', 'vmask_device.hip', '-L/opt/rocm/lib', '-lrccl'], cwd=CSRC, stderr=subprocess.DEVNULL)
    return out

if len(sys.argv) > 2 and sys.argv[2] == 'build-only':
    build('libvrg_hip_mutant.so', ['-DVRG_MUTANT']); build('libvrg_hip_mutant_chaos.so', ['-DVRG_MUTANT', '-DVRG_CHAOS'])
    raise SystemExit(0)
for label, name, flags in (('product build', None, None), ('mutant, quiet', 'libvrg_hip_mutant.so', ['-DVRG_MUTANT']),
                           ('mutant + random delays', 'libvrg_hip_mutant_chaos.so', ['-DVRG_MUTANT', '-DVRG_CHAOS'])):
    env = dict(os.environ, VRG_MUTANT_REF='/tmp/vrg_mutant_ref.txt')
    if not name and os.path.exists(env['VRG_MUTANT_REF']):
        os.remove(env['VRG_MUTANT_REF'])
    if name:
        env['VRG_HIP_LIB'] = build(name, flags)
    p = subprocess.run([sys.executable, '-c', CHILD % (ROOT, reps)], env=env, capture_output=True, text=True, timeout=1500)
    line = [l for l in p.stdout.splitlines() if l.startswith('RESULT')]
    print('%-24s %d runs: %s' % (label, reps, line[0][7:] if line else 'no result (rc %d) %s' % (p.returncode, p.stderr[-300:])), flush=True)
