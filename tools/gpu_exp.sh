#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence)
cd "$GRAFT_REPO_ROOT" || exit 1
for p in 0 -1 0 -1; do echo "== RVC_LANE_PRIO=$p"; RVC_LANE_PRIO=$p timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['config'].get('one_clip_alone_ms'))"; done
