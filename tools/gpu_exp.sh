#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
RVC_PROF_DUMP=0 timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "two_image" 2>&1 | tail -4
