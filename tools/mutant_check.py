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
caught = clean = other = 0
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
        # only the library's own cross-check (VRG_E_INTERNAL, code -8) counts as a catch; anything else - out of memory, a launch
        # failure, an ABI mismatch with a stale library - is reported for what it is
        if getattr(e, 'code', None) == -8: caught += 1
        else: other += 1; print('OTHER', type(e).__name__, str(e)[:200])
    finally:
        try: s.close()
        except Exception: pass
print('RESULT caught', caught, 'clean', clean, 'other errors', other)
'''

def build(name, flags):
    out = os.path.join(CSRC, name)
    srcs = [os.path.join(CSRC, f) for f in ('vrg_device.hip', 'vmask_device.hip', 'vrg_items.h', 'vrg_types.h', 'vrg_backend.h', 'vrg_repl.h', 'vrg_engine.cpp')]
    srcs += [os.path.join(ROOT, 'include', f) for f in ('vrg.h', 'vmask.h')]
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(f) for f in srcs):
        return out                                      # (built where the sources were edited: the library travels with the tree)
    # (to a temporary name, moved into place on success: a failed compile never leaves a stale library with an older VrgCtx layout behind)
    p = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared'] + flags + ['-o', out + '.tmp',
                        'vrg_device.hip', 'vrg_engine.cpp', 'vmask_device.hip', '-L/opt/rocm/lib', '-lrccl'], cwd=CSRC, capture_output=True, text=True)
    if p.returncode != 0:
        for f in (out, out + '.tmp'):
            if os.path.exists(f): os.remove(f)
        raise SystemExit('mutant_check: %s did not compile:\n%s' % (name, p.stderr[-3000:]))
    os.replace(out + '.tmp', out)
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
