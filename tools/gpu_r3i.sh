#!/bin/bash
set -u
tag=${1:?tag}; out="gpurun_out/$tag"; mkdir -p "$out"; export TMPDIR=/tmp
export VRG_HIP_LIB=$PWD/arterynetwork_amd/csrc/libvrg_hip_stamps.so
for off in 1 0; do for shp in 512x512x170 880x880x80; do python tools/chain_stamps.py $shp $off 60 2>&1 | grep -v amdgpu.ids | tee -a "$out/chain_stamps.log"; done; done
