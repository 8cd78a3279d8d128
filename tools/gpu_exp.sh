#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 90 stdbuf -oL tools/micro/mfmabench > gpurun_out/mfmabench_full.txt 2>&1; wc -l gpurun_out/mfmabench_full.txt
