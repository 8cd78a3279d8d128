#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -rf gpurun_out/att_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/att_prof -o att -- python3 bench.py --variant rmvpe_60s --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
grep gru gpurun_out/att_prof/*/*kernel_stats.csv | cut -c1-220
for i in 1 2; do timeout 600 python bench.py --variant rmvpe_60s --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['config'].get('one_clip_alone_ms'))"; done
