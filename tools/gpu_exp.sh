#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for cfg in "RVC_X3S_RS=4" "RVC_X3S_RS=0"; do
echo "== uvr bench $cfg"
env $cfg timeout 900 python bench.py --variant uvr_48k_v2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
done
echo "== mdx + ops tests"
timeout 1200 python -m pytest tests/test_hip_mdx23c.py tests/test_hip_ops.py -x -q 2>&1 | tail -4
