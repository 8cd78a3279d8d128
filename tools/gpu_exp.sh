#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_pipeline.py tests/test_hip_ops.py -q -x -k "pipeline or postprocess or rms" 2>&1 | tail -3
rm -rf gpurun_out/att_prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/att_prof -o att -- python3 bench.py --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
grep "rms_mix" gpurun_out/att_prof/att_kernel_stats.csv | cut -d, -f11-13
