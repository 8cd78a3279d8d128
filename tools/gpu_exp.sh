#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_hip_pipeline.py tests/test_hip_nodes.py -q -x 2>&1 | tail -3
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['config'].get('one_clip_alone_ms'))"
