#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_models.py -q -x 2>&1 | tail -3
rm -rf gpurun_out/att_prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/att_prof -o att -- python3 bench.py --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os,sys
f=glob.glob('gpurun_out/att_prof/**/*kernel_stats.csv',recursive=True)
rows=list(csv.DictReader(open(f[0])))
tot=sum(int(r['TotalDurationNs']) for r in rows)
nclip=int([r for r in rows if 'gru_scan' in r['Name']][0]['Calls'])
print('ms/clip', round(tot/1e6/nclip,2), 'launches', sum(int(r['Calls']) for r in rows)/nclip)
for r in rows:
    if 'layernorm' in r['Name']: print(r['Name'][:80], int(r['Calls'])/nclip, r['AverageNs'])
PY
