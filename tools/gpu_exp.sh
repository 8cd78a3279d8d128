#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== mdx tests"
timeout 1200 python -m pytest tests/test_hip_mdx23c.py -x -q 2>&1 | tail -5
echo "== uvr bench"
timeout 900 python bench.py --variant uvr_48k_v2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/uvr_line.json; cut -c1-200 gpurun_out/uvr_line.json; grep -o '"one_clip_alone_ms": [0-9.]*' gpurun_out/uvr_line.json
