#!/bin/bash
# Diagnostic build of the product sources with random delays at every kernel entry and hand-off (-DVRG_CHAOS) for the interleaving campaign of tools/gpu.sh <tag> chaos:
#   arterynetwork_amd/csrc/libvrg_hip_chaos.so   (never loaded by the package; VRG_HIP_LIB selects it)
# A failed compile removes the stale library instead of leaving one with an older VrgCtx layout behind.
set -euo pipefail
cd "$(dirname "$0")/../arterynetwork_amd/csrc"
out=libvrg_hip_chaos.so
if ! ${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DVRG_CHAOS -o "$out.tmp" \
        vrg_device.hip vrg_engine.cpp vmask_device.hip -L/opt/rocm/lib -lrccl; then
    rm -f "$out" "$out.tmp"
    echo "build_chaos: compile failed, $out removed" >&2
    exit 1
fi
mv "$out.tmp" "$out"
echo "$PWD/$out"
