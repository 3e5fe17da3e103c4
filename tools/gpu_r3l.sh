#!/bin/bash
set -u
tag=${1:?tag}; out="gpurun_out/$tag"; mkdir -p "$out"; export TMPDIR=/tmp
export VRG_HIP_LIB=$PWD/arterynetwork_amd/csrc/libvrg_hip_stamps.so
export VRG_CHAIN_KERNEL=1
for off in 1 0; do python tools/chain_stamps.py 512x512x170 $off 40 2>&1 | grep -v amdgpu.ids | tee -a "$out/chain_stamps_ck1.log"; done
