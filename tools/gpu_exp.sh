#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence):  gpurun --timeout 900 -- 'bash tools/gpu_exp.sh > gpurun_out/expN.txt 2>&1'
# (this version, round 5: clips in flight for the RMVPE-alone variant - its GRU scan holds 16 CUs for 60 % of a clip's time, so more lanes than three should pay; profiles/r5_rmvpe_lanes.txt)
cd "$GRAFT_REPO_ROOT" || exit 1
for l in 3 6 8 12 16; do
  timeout 300 python bench.py --variant rmvpe_60s --lanes $l --clips 24 --steps 8 --warmup 1 --no-cpu-baseline --no-roofline --no-traffic 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('lanes $l', d['value'], d['ms_per_step'], d['config']['one_clip_alone_ms'])"
done
for l in 3 4 5; do
  timeout 300 python bench.py --lanes $l --no-cpu-baseline --no-roofline --no-traffic 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('40k_v2 lanes $l', d['value'], d['ms_per_step'], d['config']['one_clip_alone_ms'])"
done
