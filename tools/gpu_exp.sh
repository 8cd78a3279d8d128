#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { env $@ timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], d['config']['one_clip_alone_ms'])"; }
run RVC_X3Q=1
run RVC_X3Q=1 RVC_X3PF64=2
run RVC_X3Q=0 RVC_X3PF64=2
run RVC_X3Q=1 RVC_BENCH_LANES=4
run RVC_X3Q=0 RVC_BENCH_LANES=4
run RVC_X3Q=1 RVC_X3PF64=2 RVC_BENCH_LANES=4
run RVC_X3Q=1
