#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_mdx23c.py -x -q 2>&1 | tail -3
timeout 600 python3 bench.py --variant uvr_48k_v2 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-200
timeout 600 python3 bench.py --variant uvr_48k_v2 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-200
