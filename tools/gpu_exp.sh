#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_hip_ops.py -q -x -k "layernorm" 2>&1 | tail -5
