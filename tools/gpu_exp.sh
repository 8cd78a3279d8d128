#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_models.py tests/test_hip_pipeline.py -q -x -k "rmvpe or pipeline" 2>&1 | tail -4
for v in 1 0; do
echo "== rmvpe_60s RVC_RMVPE_GI_X3S=$v"; RVC_RMVPE_GI_X3S=$v timeout 600 python bench.py --variant rmvpe_60s --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['config'].get('one_clip_alone_ms'))"
done
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['config'].get('one_clip_alone_ms'))"
