#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for v in "" nobar "" nobar; do
  rm -rf gpurun_out/att_prof
  lib=""; [ -n "$v" ] && lib=comfy-rvc_amd/csrc/variants/librvc_hip_$v.so
  RVC_HIP_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/att_prof -o att -- python3 bench.py --variant rmvpe_60s --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "variant '$v': $(grep gru gpurun_out/att_prof/att_kernel_stats.csv | cut -d, -f11-14)"
done
RVC_HIP_LIB=comfy-rvc_amd/csrc/variants/librvc_hip_nobar.so timeout 600 python -m pytest tests/test_hip_models.py tests/test_hip_pipeline.py -q -x -k "rmvpe or gru" 2>&1 | tail -2
