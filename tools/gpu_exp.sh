#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for cfg in "RVC_X3S_BLK=440" "RVC_X3S_BLK=220" "RVC_X3S_BLK=120" "RVC_X3S_BLK=60" "RVC_X3S_BLK=440" "RVC_X3S_BLK=220" "RVC_X3S_BLK=120"; do
echo "== $cfg"
env $cfg timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['one_clip_alone_ms'])"
done
