#!/bin/bash
# The FENCED TWIN of the product sources (-DVRG_FENCES: every poll and ticket an acquire, every announcing store and ticket a release, a release fence where the
# product only drains - vrg_items.h "backend shims") for the A/B of tools/gpu.sh <tag> fenced:
#   arterynetwork_amd/csrc/libvrg_hip_fenced.so   (never loaded by the package; VRG_HIP_LIB selects it)
set -euo pipefail
cd "$(dirname "$0")/../arterynetwork_amd/csrc"
out=libvrg_hip_fenced.so
if ! ${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DVRG_FENCES -o "$out.tmp" \
        vrg_device.hip vrg_engine.cpp vmask_device.hip -L/opt/rocm/lib -lrccl; then
    rm -f "$out" "$out.tmp"
    echo "build_fenced: compile failed, $out removed" >&2
    exit 1
fi
mv "$out.tmp" "$out"
echo "$PWD/$out"
