#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== pair tests, wait-check build"
RVC_HIP_LIB=$PWD/comfy-rvc_amd/csrc/variants/librvc_hip_check.so timeout 1200 python -m pytest tests/test_hip_ops.py -q -x -k "resblock_pair" 2>&1 | tail -4
echo "== full GPU suite, product"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
