#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== correctness + wait check build"
RVC_HIP_LIB=$PWD/comfy-rvc_amd/csrc/variants/librvc_hip_check.so timeout 900 python -m pytest tests/test_hip_ops.py -q -x -k "split_resident_resblock_pair" 2>&1 | grep -v "^$" | tail -5
timeout 900 python -m pytest tests/test_hip_ops.py -q -x -k "split_resident_resblock_pair" 2>&1 | grep -v "^$" | tail -3
for r in 3 4 0; do
echo "== bench x3q R<=$r (0 = as many as fit)"
RVC_X3Q_R=$r timeout 300 python tools/bench_split.py 128 64 2>&1 | grep -v "^$\|amdgpu.ids"
done
echo "== index tests"
timeout 600 python -m pytest tests/test_hip_pipeline.py -q -x -k "index" 2>&1 | tail -5
