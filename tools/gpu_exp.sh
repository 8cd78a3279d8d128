#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence):
#   gpurun --timeout 1800 -- 'bash tools/gpu_exp.sh > gpurun_out/expN.txt 2>&1; cat gpurun_out/expN.txt'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { env $@ timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], d['config']['one_clip_alone_ms'])"; }
run RVC_X3Q=1
run RVC_X3Q=0
