#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence):  gpurun --timeout 900 -- 'bash tools/gpu_exp.sh > gpurun_out/expN.txt 2>&1'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 90 stdbuf -oL tools/micro/mfmabench | grep -v short
