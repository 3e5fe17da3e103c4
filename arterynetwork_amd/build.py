"""Build libvrg_hip.so (hipcc, gfx950 only) in-tree so that it travels with the repo snapshot."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libvrg_hip.so')
SOURCES = ['vrg_device.hip', 'vrg_engine.cpp', 'vmask_device.hip']
HEADERS = ['vrg_types.h', 'vrg_items.h', 'vrg_backend.h', 'vrg_repl.h', os.path.join('..', '..', 'include', 'vrg.h'),
           os.path.join('..', '..', 'include', 'vmask.h')]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-o', LIB] + SOURCES + ['-L/opt/rocm/lib', '-lrccl']
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
