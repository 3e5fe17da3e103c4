#!/bin/bash
# A/B builds of the product sources: tools/ab_build.sh "name flags..." ...  ->  csrc/libvrg_hip_ab_<name>.so (git-ignored)
set -eu
cd "$(dirname "$0")/../arterynetwork_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
rm -f libvrg_hip_ab_*.so
for v in "$@"; do
  set -- $v; name=$1; shift
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o libvrg_hip_ab_$name.so vrg_device.hip vrg_engine.cpp vmask_device.hip -L/opt/rocm/lib -lrccl 2>&1 | grep -E "error" || true &
done
wait
ls -la libvrg_hip_ab_*.so
