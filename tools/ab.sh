#!/bin/bash
# tools/ab.sh "case filters" variant1 variant2 ... : interleaved A/B of kernel variants on ONE box
filters=$1; shift
for round in 1 2; do
  for v in "$@"; do
    echo "== $v (round $round)"
    RVC_HIP_LIB=$PWD/comfy-rvc_amd/csrc/variants/librvc_hip_$v.so python tools/bench_conv.py $filters 2>&1 | grep -v amdgpu.ids
  done
done
