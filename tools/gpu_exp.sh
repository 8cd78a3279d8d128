#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== product lib"
timeout 300 python tools/time_x3s.py 2>&1 | grep -v amdgpu.ids
echo "== timing build"
RVC_HIP_LIB=$PWD/comfy-rvc_amd/csrc/variants/librvc_hip_timing.so timeout 300 python tools/time_x3s.py 2>&1 | grep -v amdgpu.ids
echo "== timing build, warm weights (1 layer)"
BENCH_NLAYERS=1 RVC_HIP_LIB=$PWD/comfy-rvc_amd/csrc/variants/librvc_hip_timing.so timeout 300 python tools/time_x3s.py 2>&1 | grep -v amdgpu.ids
echo "== timing build, RS=6"
RVC_X3S_RS=6 RVC_HIP_LIB=$PWD/comfy-rvc_amd/csrc/variants/librvc_hip_timing.so timeout 300 python tools/time_x3s.py 2>&1 | grep -v amdgpu.ids
