#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_models.py tests/test_hip_pipeline.py -q -x -k "rmvpe or pipeline" 2>&1 | tail -3
for v in 1 0; do
  rm -rf gpurun_out/att_prof
  RVC_RMVPE_CBR2=$v timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/att_prof -o att -- python3 bench.py --variant rmvpe_60s --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python3 - $v <<'PY'
import csv,glob,sys
rows=list(csv.DictReader(open(glob.glob('gpurun_out/att_prof/**/*kernel_stats.csv',recursive=True)[0])))
n=int([r for r in rows if 'gru_scan' in r['Name']][0]['Calls'])
tot=sum(int(r['TotalDurationNs']) for r in rows)/1e6/n
gru=sum(int(r['TotalDurationNs']) for r in rows if 'gru_scan' in r['Name'])/1e6/n
print('CBR2', sys.argv[1], 'rmvpe60 ms/clip', round(tot,3), 'minus gru', round(tot-gru,3), 'launches', sum(int(r['Calls']) for r in rows)/n)
for r in rows:
    if 'conv3_small' in r['Name'] or 'cbr2' in r['Name'] or 'conv_x3_kernel<' in r['Name'] or 'conv_mfma' in r['Name']: print('   ', r['Name'][:78], int(r['Calls'])/n, round(float(r['AverageNs'])/1e3,1))
PY
done
for i in 1 2; do timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['config'].get('one_clip_alone_ms'))"; done
