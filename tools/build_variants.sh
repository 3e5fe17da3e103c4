#!/bin/bash
# Diagnostic / experimental builds of the library from the product sources (never loaded by the package itself; a tool
# or test selects one through VRG_HIP_LIB or VrgLib(path)):
#   libvrg_hip_stamps.so  -DVRG_STAMPS              in-kernel time stamps of the band chain (tools/chain_stamps.py)
#   libvrg_hip_chain.so   -DVRG_CHAIN -DVRG_STAMPS  + the persistent band kernel k_chain (option chain_kernel), band-side loads past L1
set -eu
cd "$(dirname "$0")/../arterynetwork_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
for v in "stamps -DVRG_STAMPS" "chain -DVRG_CHAIN -DVRG_STAMPS"; do
  set -- $v; name=$1; shift
  if [ ! -e libvrg_hip_$name.so ] || [ vrg_device.hip -nt libvrg_hip_$name.so ] || [ vrg_items.h -nt libvrg_hip_$name.so ] || [ vrg_types.h -nt libvrg_hip_$name.so ] || [ vrg_engine.cpp -nt libvrg_hip_$name.so ]; then
    $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o libvrg_hip_$name.so vrg_device.hip vrg_engine.cpp vmask_device.hip -L/opt/rocm/lib -lrccl 2>&1 | grep -E "error" || true
  fi
done
ls -la libvrg_hip_*.so
