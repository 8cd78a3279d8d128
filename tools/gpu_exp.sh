#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -5
timeout 600 python bench.py --steps 5 --warmup 2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-2500
