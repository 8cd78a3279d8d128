#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_hip_pipeline.py tests/test_hip_nodes.py tests/test_hip_models.py -q -x 2>&1 | tail -4
