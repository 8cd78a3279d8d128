#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
RVC_BENCH_DEBUG=1 timeout 900 python bench.py --variant uvr_48k_v2 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | grep -v amdgpu.ids | grep "alone\|metric" | cut -c1-300
RVC_BENCH_DEBUG=1 RVC_MDX_STREAMS=3 timeout 900 python bench.py --variant uvr_48k_v2 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | grep -v amdgpu.ids | grep "alone\|metric" | cut -c1-300
