#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python tools/gpu_exp.py 2>&1 | grep -v amdgpu.ids
