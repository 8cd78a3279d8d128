#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_models.py tests/test_hip_pipeline.py -q -x -k "synth or pipeline" 2>&1 | tail -4
for v in 1 0; do
echo "== RVC_NOISE_STREAM=$v"; RVC_NOISE_STREAM=$v timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['config'].get('one_clip_alone_ms'))"
done
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
    if 'noise_add' in r['Name'] or 'frames_kernel' in r['Name'] or 'conv_mfma' in r['Name']: print(r['Name'][:80], int(r['Calls'])/nclip, r['AverageNs'])
PY
