#!/bin/bash
set -u
tag=${1:?tag}; out="gpurun_out/$tag"; mkdir -p "$out"; export TMPDIR=/tmp
export VRG_HIP_LIB=$PWD/arterynetwork_amd/csrc/libvrg_hip_stamps.so
python tools/chain_stamps.py 512x512x170 1 60 2>&1 | grep -v amdgpu.ids | tee "$out/chain_stamps.log" | grep "k_mark\|step"
python tools/chain_stamps.py 512x512x170 0 60 2>&1 | grep -v amdgpu.ids | tee "$out/chain_stamps_beside.log" | grep "k_mark\|step"
