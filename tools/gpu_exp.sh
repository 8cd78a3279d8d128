#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence):  gpurun --timeout 900 -- 'bash tools/gpu_exp.sh > gpurun_out/expN.txt 2>&1'
# (this version, round 5: the UVR chain - MDX23C is 98 % of it - with the split-resident kernel's two reduction loops; profiles/r5_mdx_x3s_mode.txt)
cd "$GRAFT_REPO_ROOT" || exit 1
run() { env $@ timeout 400 python bench.py --variant uvr_48k_v2 --no-cpu-baseline --no-roofline --no-traffic --steps 6 --warmup 1 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$*', d['value'], d['config']['one_clip_alone_ms'], d['config'].get('one_clip_alone_ms_3_chunk_streams'))"; }
run RVC_X3S_MODE=0
run RVC_X3S_MODE=2
run RVC_X3S_MODE=2 RVC_X3S_D=3
run RVC_X3S_MODE=0
