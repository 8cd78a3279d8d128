#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== correctness (product lib, x3q on)"
timeout 900 python -m pytest tests/test_hip_ops.py -q -x -k "split_resident_resblock_pair" 2>&1 | tail -5
echo "== wait check build"
RVC_HIP_LIB=$PWD/comfy-rvc_amd/csrc/variants/librvc_hip_check.so timeout 900 python -m pytest tests/test_hip_ops.py -q -x -k "split_resident_resblock_pair" 2>&1 | tail -5
echo "== bench x3q=0"
RVC_X3Q=0 timeout 300 python tools/bench_split.py 2>&1 | grep -v "^$\|amdgpu.ids"
echo "== bench x3q=1"
RVC_X3Q=1 timeout 300 python tools/bench_split.py 2>&1 | grep -v "^$\|amdgpu.ids"
echo "== bench x3q=0 again"
RVC_X3Q=0 timeout 300 python tools/bench_split.py 128 2>&1 | grep -v "^$\|amdgpu.ids"
echo "== bench x3q=1 again"
RVC_X3Q=1 timeout 300 python tools/bench_split.py 128 2>&1 | grep -v "^$\|amdgpu.ids"
