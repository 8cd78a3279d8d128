#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== full gpu tests"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
echo "== uvr bench"
timeout 900 python bench.py --variant uvr_48k_v2 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/uvr_line.json; cut -c1-200 gpurun_out/uvr_line.json; grep -o '"one_clip_alone_ms": [0-9.]*' gpurun_out/uvr_line.json
echo "== main bench"
timeout 900 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
