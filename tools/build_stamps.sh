#!/bin/bash
# Diagnostic build of the product sources with in-kernel time stamps (-DVRG_STAMPS) for tools/chain_stamps.py:
#   arterynetwork_amd/csrc/libvrg_hip_stamps.so   (never loaded by the package; VRG_HIP_LIB selects it)
# A failed compile removes the stale library instead of leaving one with an older VrgCtx layout behind.
set -euo pipefail
cd "$(dirname "$0")/../arterynetwork_amd/csrc"
out=libvrg_hip_stamps.so
if ! ${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DVRG_STAMPS -o "$out.tmp" \
        vrg_device.hip vrg_engine.cpp vmask_device.hip -L/opt/rocm/lib -lrccl; then
    rm -f "$out" "$out.tmp"
    echo "build_stamps: compile failed, $out removed" >&2
    exit 1
fi
mv "$out.tmp" "$out"
echo "$PWD/$out"
