#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence):  gpurun --timeout 900 -- 'bash tools/gpu_exp.sh > gpurun_out/expN.txt 2>&1'
# (this version, round 5: XCD tile order of the split-resident kernel on the UVR chain - RVC_X3S_ROWFAST: 0 column tiles fastest, 1 by operand size, 2 row tiles fastest;
#  PMC traffic of MDX23C's products is 1.9x algorithmic with the default; profiles/r5_mdx_rowfast.txt)
cd "$GRAFT_REPO_ROOT" || exit 1
run() { env $@ timeout 400 python bench.py --variant uvr_48k_v2 --no-cpu-baseline --no-traffic --steps 6 --warmup 1 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$*', d['value'], d['config']['one_clip_alone_ms'], r['kernel'], r['kernel_ms_per_clip'], r['frac'])"; }
run RVC_X3S_ROWFAST=0
run RVC_X3S_ROWFAST=1
run RVC_X3S_ROWFAST=2
run RVC_X3S_ROWFAST=0
run RVC_X3S_ROWFAST=1
