#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== persistent pair test, product lib"
timeout 1200 python -m pytest tests/test_hip_ops.py -q -x -k "persistent_resblock_pair" 2>&1 | tail -8
echo "== persistent pair test, wait-check build"
RVC_HIP_LIB=$PWD/comfy-rvc_amd/csrc/variants/librvc_hip_check.so timeout 1200 python -m pytest tests/test_hip_ops.py -q -x -k "persistent_resblock_pair" 2>&1 | tail -8
